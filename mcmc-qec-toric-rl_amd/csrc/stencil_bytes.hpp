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
