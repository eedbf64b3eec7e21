// Byte-per-qubit __device__ stencils of the toric code: the GPU side of the
// Toric_code API surface (src/toric_model.py:33-56).  Used by the batched
// primitive kernels and by the single-chain update kernel; the ladder kernel
// (pteq_rs.hip) uses the 2-bit packed equivalents.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qecmc {

__device__ __forceinline__ int flip_b(uint8_t *q, int op)
{
    const uint8_t old = *q, neu = (uint8_t)(old ^ op);
    *q = neu;
    return (old == 0) - (neu == 0);   // +1 new error, -1 removed error (toric_model.py:279-282)
}

// _apply_stabilizer, toric_model.py:256-284 (in place on m; the four qubits are distinct for L >= 2)
__device__ inline int toric_apply_stabilizer_b(int L, uint8_t *m, int row, int col, int op)
{
    const int LL = L * L;
    const int rm = row == 0 ? L - 1 : row - 1, rp = row == L - 1 ? 0 : row + 1;
    const int cm = col == 0 ? L - 1 : col - 1, cp = col == L - 1 ? 0 : col + 1;
    int dE = flip_b(&m[LL + row * L + col], op) + flip_b(&m[row * L + col], op);
    if (op == 1) {
        dE += flip_b(&m[LL + row * L + cm], op);    // (1, r, c-1)
        dE += flip_b(&m[rm * L + col], op);         // (0, r-1, c)
    } else {
        dE += flip_b(&m[row * L + cp], op);         // (0, r, c+1)
        dE += flip_b(&m[LL + rp * L + col], op);    // (1, r+1, c)
    }
    return dE;
}

// _apply_logical, toric_model.py:179-225
__device__ inline int toric_apply_logical_b(int L, uint8_t *m, int op, int layer, int xpos, int zpos)
{
    if (op == 0) return 0;
    const bool do_x = (op == 1 || op == 2), do_z = (op == 3 || op == 2);
    uint8_t *ml = m + layer * L * L;
    int dE = 0;
    for (int i = 0; i < L; ++i) {
        if (do_x) dE += flip_b(layer == 0 ? &ml[xpos * L + i] : &ml[i * L + xpos], 1);
        if (do_z) dE += flip_b(layer == 0 ? &ml[i * L + zpos] : &ml[zpos * L + i], 3);
    }
    return dE;
}

// _count_errors, toric_model.py:174-176
__device__ inline int count_errors_b(int nq, const uint8_t *m)
{
    int n = 0;
    for (int i = 0; i < nq; ++i) n += m[i] != 0;
    return n;
}

// _define_equivalence_class, toric_model.py:317-351
__device__ inline int toric_eq_class_b(int L, const uint8_t *m)
{
    const int LL = L * L;
    int par[2][2] = {{0, 0}, {0, 0}};
    for (int l = 0; l < 2; ++l)
        for (int i = 0; i < LL; ++i) {
            const int v = m[l * LL + i];
            par[l][0] ^= (v == 1) | (v == 2);   // X component
            par[l][1] ^= (v == 3) | (v == 2);   // Z component
        }
    return par[0][0] + 2 * par[0][1] + 4 * par[1][0] + 8 * par[1][1];
}

// _to_class, toric_model.py:354-377
__device__ inline void toric_to_class_b(int L, uint8_t *m, int eq)
{
    const int diff = eq ^ toric_eq_class_b(L, m);
    const int ops = diff ^ ((diff & 0xA) >> 1);
    toric_apply_logical_b(L, m, ops & 3, 0, 0, 0);
    toric_apply_logical_b(L, m, ops >> 2, 1, 0, 0);
}

// Toric_code.syndrom, toric_model.py:58-101; d = uint8[2][L][L]
__device__ inline void toric_syndrome_b(int L, const uint8_t *m, uint8_t *d)
{
    const int LL = L * L;
    for (int r = 0; r < L; ++r)
        for (int c = 0; c < L; ++c) {
            const int rm = r == 0 ? L - 1 : r - 1, rp = r == L - 1 ? 0 : r + 1;
            const int cm = c == 0 ? L - 1 : c - 1, cp = c == L - 1 ? 0 : c + 1;
            auto yz = [&](int l, int rr, int cc) { const int v = m[l * LL + rr * L + cc]; return (int)(v >= 2); };
            auto xy = [&](int l, int rr, int cc) { const int v = m[l * LL + rr * L + cc]; return (int)(v == 1 || v == 2); };
            d[r * L + c] = (uint8_t)(yz(0, r, c) ^ yz(0, rm, c) ^ yz(1, r, c) ^ yz(1, r, cm));
            d[LL + r * L + c] = (uint8_t)(xy(0, r, c) ^ xy(0, r, cp) ^ xy(1, r, c) ^ xy(1, rp, c));
        }
}


}  // namespace qecmc

// ---------------------------------------------------------------------------------------------
// XZZX code (src/xzzx_model.py) and rotated surface code (src/rotated_surface_model.py): L x L
// qubits, (L-1)^2 four-qubit plaquettes + 2(L-1) two-qubit boundary half-plaquettes.
// ---------------------------------------------------------------------------------------------
namespace qecmc {

constexpr int kCodeToric = 0, kCodeXzzx = 1, kCodeRotated = 2, kCodePlanar = 3;

// qubits of a state, stabilizer generators of a code
__host__ __device__ inline int code_nq_of(int code, int L) { return (code == kCodeToric || code == kCodePlanar) ? 2 * L * L : L * L; }
__host__ __device__ inline int surf_ngen(int code, int L) { return code == kCodePlanar ? 2 * L * (L - 1) : L * L - 1; }

// generator g in table order -> (row, col, operator): xzzx / rotated -- full plaquettes row-major, then half plaquette h/4 on
// side h%4; planar -- the L(L-1) X-type generators over (row in [0,L-1), col in [0,L)), then the Z-type ones over
// (row in [0,L), col in [0,L-1)) (planar_model.py:343-352)
__host__ __device__ inline void surf_gen_rco(int code, int L, int g, int &row, int &col, int &op)
{
    if (code == kCodePlanar) {
        const int nx = L * (L - 1);
        if (g < nx) { op = 1; row = g / L; col = g % L; }
        else { op = 3; row = (g - nx) / (L - 1); col = (g - nx) % (L - 1); }
        return;
    }
    const int nf = (L - 1) * (L - 1);
    if (g < nf) { op = 1; row = g / (L - 1); col = g % (L - 1); }
    else { op = 3; row = (g - nf) >> 2; col = (g - nf) & 3; }
}

// generator (row, col, operator) -> up to 4 (flat site, Pauli) pairs; operator 1 = full plaquette,
// 3 = half plaquette `row` on side `col` (xzzx_model.py:369-434, rotated_surface_model.py:357-381)
__host__ __device__ inline int surf_generator(int code, int L, int row, int col, int op, int sites[4], int paulis[4])
{
    if (code == kCodePlanar) {
        // planar_model.py:297-326: site = layer*L*L + r*L + c; boundary generators have three sites
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
        if (code == kCodeXzzx) {
            sites[0] = row * L + col;           paulis[0] = 1;
            sites[1] = (row + 1) * L + col;     paulis[1] = 3;
            sites[2] = row * L + col + 1;       paulis[2] = 3;
            sites[3] = (row + 1) * L + col + 1; paulis[3] = 1;
        } else {
            const int p = (row % 2 == col % 2) ? 1 : 3;
            sites[0] = row * L + col;       sites[1] = row * L + col + 1;
            sites[2] = (row + 1) * L + col; sites[3] = (row + 1) * L + col + 1;
            paulis[0] = paulis[1] = paulis[2] = paulis[3] = p;
        }
        return 4;
    }
    switch (col) {
        case 0: sites[0] = 2 * row + 1;               sites[1] = 2 * row + 2; break;
        case 1: sites[0] = (2 * row + 1) * L + L - 1; sites[1] = (2 * row + 2) * L + L - 1; break;
        case 2: sites[0] = (L - 1) * L + 2 * row;     sites[1] = (L - 1) * L + 2 * row + 1; break;
        default: sites[0] = (2 * row) * L;            sites[1] = (2 * row + 1) * L; break;
    }
    if (code == kCodeXzzx) {
        paulis[0] = (col == 0 || col == 3) ? 3 : 1;
        paulis[1] = (col == 0 || col == 3) ? 1 : 3;
    } else {
        paulis[0] = paulis[1] = (col == 0 || col == 2) ? 1 : 3;
    }
    return 2;
}

__device__ inline int surf_apply_stabilizer_b(int code, int L, uint8_t *m, int row, int col, int op)
{
    int sites[4], paulis[4], dE = 0;
    const int n = surf_generator(code, L, row, col, op, sites, paulis);
    for (int i = 0; i < n; ++i) dE += flip_b(&m[sites[i]], paulis[i]);
    return dE;
}

// _apply_logical: xzzx_model.py:279-313 (anti-diagonal X, diagonal Z, positions ignored);
// rotated_surface_model.py:251-282 (X on column X_pos iff op in {1,3}, Z on row Z_pos iff op in {2,3})
__device__ inline int surf_apply_logical_b(int code, int L, uint8_t *m, int op, int xpos, int zpos)
{
    if (op == 0) return 0;
    int dE = 0;
    if (code == kCodePlanar) {      // planar_model.py:235-268: X along row X_pos of layer 0 iff op in {1,3}, Z along column Z_pos iff op in {2,3}
        for (int i = 0; i < L; ++i) {
            if (op == 1 || op == 3) dE += flip_b(&m[xpos * L + i], 1);
            if (op == 2 || op == 3) dE += flip_b(&m[i * L + zpos], 3);
        }
        return dE;
    }
    if (code == kCodeXzzx) {
        if (op == 1 || op == 2) for (int i = 0; i < L; ++i) dE += flip_b(&m[i * L + (L - 1 - i)], 1);
        if (op == 3 || op == 2) for (int i = 0; i < L; ++i) dE += flip_b(&m[i * L + i], 3);
    } else {
        if (op == 1 || op == 3) for (int i = 0; i < L; ++i) dE += flip_b(&m[i * L + xpos], 1);
        if (op == 2 || op == 3) for (int i = 0; i < L; ++i) dE += flip_b(&m[zpos * L + i], 3);
    }
    return dE;
}

// _define_equivalence_class: xzzx_model.py:455-486, rotated_surface_model.py:411-420
__device__ inline int surf_eq_class_b(int code, int L, const uint8_t *m)
{
    int x = 0, z = 0;
    if (code == kCodePlanar) {      // planar_model.py:379-390: X/Y parity of layer 0's first column, Z/Y parity of its first row
        for (int i = 0; i < L; ++i) {
            x ^= (m[i * L] == 1) | (m[i * L] == 2);
            z ^= (m[i] == 3) | (m[i] == 2);
        }
        return x + 2 * z;
    }
    if (code == kCodeXzzx) {
        for (int i = 0; i < L; ++i) {
            const int a = m[i], b = m[i * L];
            x ^= (a == 2) ^ ((i & 1) ? (a == 3) : (a == 1));
            z ^= (b == 2) ^ ((i & 1) ? (b == 1) : (b == 3));
        }
        return x ? (z ? 2 : 1) : (z ? 3 : 0);
    }
    for (int i = 0; i < L; ++i) {
        x ^= (m[i] == 1) | (m[i] == 2);
        z ^= (m[i * L] == 3) | (m[i * L] == 2);
    }
    return x + 2 * z;
}

// xzzx_code.syndrome / RotSurCode.syndrome: d = uint8[L+1][L+1]
__device__ inline void surf_syndrome_b(int code, int L, const uint8_t *m, uint8_t *d)
{
    const int S = L + 1;
    for (int i = 0; i < S * S; ++i) d[i] = 0;
    auto defect = [&](int row, int col, int op) {
        int sites[4], paulis[4], v = 0;
        const int n = surf_generator(code, L, row, col, op, sites, paulis);
        for (int i = 0; i < n; ++i) v ^= (m[sites[i]] != 0 && m[sites[i]] != paulis[i]);
        return (uint8_t)v;
    };
    for (int i = 0; i < L - 1; ++i)
        for (int j = 0; j < L - 1; ++j) d[(i + 1) * S + j + 1] = defect(i, j, 1);
    for (int i = 0; i < (L - 1) / 2; ++i) {
        d[2 * i + 2] = defect(i, 0, 3);
        d[(2 * i + 2) * S + L] = defect(i, 1, 3);
        d[L * S + 2 * i + 1] = defect(i, 2, 3);
        d[(2 * i + 1) * S] = defect(i, 3, 3);
    }
}


// Planar_code.syndrom, planar_model.py:134-153: d = vertex_defects uint8[L-1][L] followed by plaquette_defects uint8[L][L-1]
__device__ inline void planar_syndrome_b(int L, const uint8_t *m, uint8_t *d)
{
    const int LL = L * L;
    auto yz = [&](int l, int r, int c) { const uint8_t q = m[l * LL + r * L + c]; return (int)(q == 2 || q == 3); };
    auto xy = [&](int l, int r, int c) { const uint8_t q = m[l * LL + r * L + c]; return (int)(q == 1 || q == 2); };
    for (int r = 0; r < L - 1; ++r)
        for (int c = 0; c < L; ++c)
            d[r * L + c] = (uint8_t)(yz(0, r + 1, c) ^ yz(0, r, c) ^ yz(1, r, c) ^ yz(1, r, (c + L - 1) % L));
    uint8_t *q = d + (L - 1) * L;
    for (int r = 0; r < L; ++r)
        for (int c = 0; c < L - 1; ++c)
            q[r * (L - 1) + c] = (uint8_t)(xy(0, r, c + 1) ^ xy(0, r, c) ^ xy(1, r, c) ^ xy(1, (r + L - 1) % L, c));
}

}  // namespace qecmc
