/*
 * qecmc_oracle_surf.c -- CPU oracle, part 2 (TEST INFRASTRUCTURE ONLY): the
 * XZZX code (src/xzzx_model.py) and the rotated surface code
 * (src/rotated_surface_model.py) stencils.  Both live on an L x L qubit matrix
 * (uint8, C order) with (L-1)^2 four-qubit plaquettes and 2(L-1) two-qubit
 * boundary half-plaquettes; they differ in the Pauli each site receives, in the
 * logical operators and in the class function.  Also the planar (unrotated) surface code of src/planar_model.py:
 * uint8[2][L][L] with layer 1 living on its first L-1 rows / columns, L(L-1) X-type and L(L-1) Z-type generators of
 * three (boundary) or four sites -- the model Chain.update_chain_fast is hard-wired to.
 *
 * Parity status: PINNED by tests/test_oracle_golden.py against f1_surf.npz /
 * f2_surf.npz (captured from the reference).
 */
#include "qecmc_oracle.h"

#include <string.h>

/* The generator (row, col, operator) as (flat site, Pauli) pairs; returns the number of sites.
 * operator 1 = full plaquette at (row, col), row, col in [0, L-2];
 * operator 3 = half plaquette number `row` on side `col` (0 top, 1 right, 2 bottom, 3 left).
 * xzzx_model.py:369-434, rotated_surface_model.py:357-381. */
int orc_surf_generator(int code, int L, int row, int col, int op, int sites[4], int paulis[4])
{
    if (code == ORC_PLANAR) {
        /* planar_model.py:297-326.  site = layer*L*L + r*L + c.  op 1 at (row in [0,L-1), col in [0,L)):
         * (0,row,col), (0,row+1,col) and the layer-1 qubits (1,row,col), (1,row,col-1) that exist;
         * op 3 at (row in [0,L), col in [0,L-1)): (0,row,col), (0,row,col+1) and (1,row,col), (1,row-1,col) */
        const int LL = L * L;
        int n = 0;
        if (op == 1) {
            sites[n++] = row * L + col;
            sites[n++] = (row + 1) * L + col;
            if (col == 0) sites[n++] = LL + row * L;
            else if (col == L - 1) sites[n++] = LL + row * L + col - 1;
            else { sites[n++] = LL + row * L + col; sites[n++] = LL + row * L + col - 1; }
        } else {
            sites[n++] = row * L + col;
            sites[n++] = row * L + col + 1;
            if (row == 0) sites[n++] = LL + col;
            else if (row == L - 1) sites[n++] = LL + (row - 1) * L + col;
            else { sites[n++] = LL + row * L + col; sites[n++] = LL + (row - 1) * L + col; }
        }
        for (int i = 0; i < n; ++i) paulis[i] = op;
        return n;
    }
    if (op == 1) {
        if (code == ORC_XZZX) {
            sites[0] = row * L + col;       paulis[0] = 1;
            sites[1] = (row + 1) * L + col; paulis[1] = 3;
            sites[2] = row * L + col + 1;   paulis[2] = 3;
            sites[3] = (row + 1) * L + col + 1; paulis[3] = 1;
        } else {
            const int p = (row % 2 == col % 2) ? 1 : 3;
            sites[0] = row * L + col;       sites[1] = row * L + col + 1;
            sites[2] = (row + 1) * L + col; sites[3] = (row + 1) * L + col + 1;
            paulis[0] = paulis[1] = paulis[2] = paulis[3] = p;
        }
        return 4;
    }
    /* half plaquettes: the same coordinates in both models */
    switch (col) {
    case 0: sites[0] = 2 * row + 1;                 sites[1] = 2 * row + 2; break;                    /* top row    */
    case 1: sites[0] = (2 * row + 1) * L + L - 1;   sites[1] = (2 * row + 2) * L + L - 1; break;      /* right col  */
    case 2: sites[0] = (L - 1) * L + 2 * row;       sites[1] = (L - 1) * L + 2 * row + 1; break;      /* bottom row */
    default: sites[0] = (2 * row) * L;              sites[1] = (2 * row + 1) * L; break;              /* left col   */
    }
    if (code == ORC_XZZX) {
        static const int pa[4][2] = {{3, 1}, {1, 3}, {1, 3}, {3, 1}};
        paulis[0] = pa[col][0]; paulis[1] = pa[col][1];
    } else {
        paulis[0] = paulis[1] = (col == 0 || col == 2) ? 1 : 3;
    }
    return 2;
}

static inline int flip_site(uint8_t *q, int op)
{
    uint8_t old = *q, neu = (uint8_t)(old ^ op);
    *q = neu;
    if (old && !neu) return -1;
    if (neu && !old) return 1;
    return 0;
}

/* _apply_stabilizer, xzzx_model.py:360-436 / rotated_surface_model.py:349-392 */
int orc_surf_apply_stabilizer(int code, int L, const uint8_t *in, uint8_t *out, int row, int col, int op)
{
    int sites[4], paulis[4];
    if (out != in) memcpy(out, in, (size_t)orc_nq(code, L));
    const int n = orc_surf_generator(code, L, row, col, op, sites, paulis);
    int dE = 0;
    for (int i = 0; i < n; ++i) dE += flip_site(&out[sites[i]], paulis[i]);
    return dE;
}

/* _find_syndrome: parity of the sites that anticommute with the generator */
static int surf_defect(int code, int L, const uint8_t *m, int row, int col, int op)
{
    int sites[4], paulis[4], d = 0;
    const int n = orc_surf_generator(code, L, row, col, op, sites, paulis);
    for (int i = 0; i < n; ++i) {
        const uint8_t q = m[sites[i]];
        if (q != 0 && q != paulis[i]) d ^= 1;
    }
    return d;
}

/* xzzx_code.syndrome, xzzx_model.py:60-83 (= RotSurCode.syndrome :108-130); defects uint8[L+1][L+1] */
void orc_surf_syndrome(int code, int L, const uint8_t *in, uint8_t *defects)
{
    const int S = L + 1;
    memset(defects, 0, (size_t)S * S);
    for (int i = 0; i < L - 1; ++i)
        for (int j = 0; j < L - 1; ++j) defects[(i + 1) * S + j + 1] = (uint8_t)surf_defect(code, L, in, i, j, 1);
    for (int i = 0; i < (L - 1) / 2; ++i)
        for (int j = 0; j < 4; ++j) {
            int r, c;
            if (j == 0) { r = 0; c = 2 * i + 2; }
            else if (j == 1) { r = 2 * i + 2; c = L; }
            else if (j == 2) { r = L; c = 2 * i + 1; }
            else { r = 2 * i + 1; c = 0; }
            defects[r * S + c] = (uint8_t)surf_defect(code, L, in, i, j, 3);
        }
}

/* _apply_logical.  XZZX (xzzx_model.py:279-313): X on the anti-diagonal iff op in {1,2}, Z on the
 * diagonal iff op in {3,2}; positions are ignored.  Rotated (rotated_surface_model.py:251-282): X on
 * column X_pos iff op in {1,3}, Z on row Z_pos iff op in {2,3}. */
int orc_surf_apply_logical(int code, int L, const uint8_t *in, uint8_t *out, int op, int xpos, int zpos)
{
    if (out != in) memcpy(out, in, (size_t)orc_nq(code, L));
    if (op == 0) return 0;
    int dE = 0;
    if (code == ORC_PLANAR) {
        /* planar_model.py:235-268: X along row X_pos of layer 0 iff op in {1,3}, Z along column Z_pos iff op in {2,3} */
        const int do_x = (op == 1 || op == 3), do_z = (op == 2 || op == 3);
        for (int i = 0; i < L; ++i) {
            if (do_x) dE += flip_site(&out[xpos * L + i], 1);
            if (do_z) dE += flip_site(&out[i * L + zpos], 3);
        }
        return dE;
    }
    if (code == ORC_XZZX) {
        const int do_x = (op == 1 || op == 2), do_z = (op == 3 || op == 2);
        if (do_x) for (int i = 0; i < L; ++i) dE += flip_site(&out[i * L + (L - 1 - i)], 1);
        if (do_z) for (int i = 0; i < L; ++i) dE += flip_site(&out[i * L + i], 3);
    } else {
        const int do_x = (op == 1 || op == 3), do_z = (op == 2 || op == 3);
        if (do_x) for (int i = 0; i < L; ++i) dE += flip_site(&out[i * L + xpos], 1);
        if (do_z) for (int i = 0; i < L; ++i) dE += flip_site(&out[zpos * L + i], 3);
    }
    return dE;
}

/* _define_equivalence_class, xzzx_model.py:455-486 / rotated_surface_model.py:411-420 */
int orc_surf_eq_class(int code, int L, const uint8_t *m)
{
    int x = 0, z = 0;
    if (code == ORC_PLANAR) {                               /* planar_model.py:379-390 */
        for (int i = 0; i < L; ++i) {
            x += (m[i * L] == 1) || (m[i * L] == 2);          /* first column of layer 0 */
            z += (m[i] == 3) || (m[i] == 2);                  /* first row of layer 0 */
        }
        return (x % 2) + 2 * (z % 2);
    }
    if (code == ORC_XZZX) {
        for (int i = 0; i < L; ++i) {
            x += m[i] == 2;                                   /* row 0 */
            z += m[i * L] == 2;                               /* column 0 */
            x += (i % 2 == 0) ? (m[i] == 1) : (m[i] == 3);
            z += (i % 2 == 0) ? (m[i * L] == 3) : (m[i * L] == 1);
        }
        if (x % 2 == 0) return (z % 2 == 0) ? 0 : 3;
        return (z % 2 == 0) ? 1 : 2;
    }
    for (int i = 0; i < L; ++i) {
        x += (m[i] == 1) || (m[i] == 2);
        z += (m[i * L] == 3) || (m[i * L] == 2);
    }
    return (x % 2) + 2 * (z % 2);
}

/* generators in table order: xzzx / rotated -- full plaquettes row-major, then half plaquette h/4 on side h%4;
 * planar -- the L(L-1) X-type generators row-major over (row in [0,L-1), col in [0,L)), then the Z-type ones
 * row-major over (row in [0,L), col in [0,L-1)) */
int orc_surf_ngen(int code, int L) { return code == ORC_PLANAR ? 2 * L * (L - 1) : L * L - 1; }

void orc_surf_gen_rco(int code, int L, int g, int *row, int *col, int *op)
{
    if (code == ORC_PLANAR) {
        const int nx = L * (L - 1);
        if (g < nx) { *op = 1; *row = g / L; *col = g % L; }
        else { *op = 3; *row = (g - nx) / (L - 1); *col = (g - nx) % (L - 1); }
        return;
    }
    const int nf = (L - 1) * (L - 1);
    if (g < nf) { *op = 1; *row = g / (L - 1); *col = g % (L - 1); }
    else { *op = 3; *row = (g - nf) / 4; *col = (g - nf) % 4; }
}

/* Planar_code.syndrom, planar_model.py:134-153 */
void orc_planar_syndrome(int L, const uint8_t *in, uint8_t *vertex, uint8_t *plaquette)
{
    const int LL = L * L;
#define YZ(l, r, c) ((in[(l) * LL + (r) * L + (c)] == 2) || (in[(l) * LL + (r) * L + (c)] == 3))
#define XY(l, r, c) ((in[(l) * LL + (r) * L + (c)] == 1) || (in[(l) * LL + (r) * L + (c)] == 2))
    for (int r = 0; r < L - 1; ++r)                                         /* vertex_defects [L-1][L] */
        for (int c = 0; c < L; ++c)
            vertex[r * L + c] = (uint8_t)((YZ(0, r + 1, c) ^ YZ(0, r, c)) ^ (YZ(1, r, c) ^ YZ(1, r, (c + L - 1) % L)));
    for (int r = 0; r < L; ++r)                                             /* plaquette_defects [L][L-1] */
        for (int c = 0; c < L - 1; ++c)
            plaquette[r * (L - 1) + c] = (uint8_t)((XY(0, r, c + 1) ^ XY(0, r, c)) ^ (XY(1, r, c) ^ XY(1, (r + L - 1) % L, c)));
#undef YZ
#undef XY
}
