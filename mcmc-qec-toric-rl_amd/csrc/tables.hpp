// Host-side table builders of libqecmc: everything the plan precomputes for the kernels -- acceptance and swap thresholds,
// ladder temperatures, power tables of the biased / alpha rules, stabilizer-generator tables, logical-operator masks, the
// count-change table, the colour phases of scan = 2.  Pure host C++ (no HIP call): capi.hip includes it, and
// tables_test_api.cpp builds it alone with -fsanitize=address,undefined so that tests/test_host_tables.py can check every
// table against values the CPU oracle computes ("-fsanitize=address on host lib").
#pragma once
#include "../../include/qecmc.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "stencil_bytes.hpp"   // surf_ngen / surf_gen_rco / surf_generator: the plaquette codes' generator geometry

namespace qecmc {
namespace tables {

// px^n, py^n, pz^n, pI^n for n = 0..nq (mcmc_biased.py:25-31): the same libm pow() the reference calls
inline std::vector<double> bias_tables(double p, double eta, size_t nq)
{
    const double pz = p * eta / (eta + 1), px = p / (2 * (eta + 1)), py = px, pi = 1 - px - py - pz;
    std::vector<double> t(4 * (nq + 1));
    for (size_t n = 0; n <= nq; ++n) {
        t[n] = std::pow(px, (double)n);
        t[(nq + 1) + n] = std::pow(py, (double)n);
        t[2 * (nq + 1) + n] = std::pow(pz, (double)n);
        t[3 * (nq + 1) + n] = std::pow(pi, (double)n);
    }
    return t;
}

// The same table for the "alpha" noise model: (p_x, p_y, p_z) from (pz_tilde, alpha) exactly as Chain_alpha.update_chain
// forms them (mcmc_alpha.py:31-36)
inline std::vector<double> alpha_tables(double pz_tilde, double alpha, size_t nq)
{
    const double p_tilde = pz_tilde + 2 * std::pow(pz_tilde, alpha);
    const double p = p_tilde / (1 + p_tilde);
    const double pz = pz_tilde * (1 - p), px = std::pow(pz_tilde, alpha) * (1 - p), py = px, pi = 1 - px - py - pz;
    std::vector<double> t(4 * (nq + 1));
    for (size_t n = 0; n <= nq; ++n) {
        t[n] = std::pow(px, (double)n);
        t[(nq + 1) + n] = std::pow(py, (double)n);
        t[2 * (nq + 1) + n] = std::pow(pz, (double)n);
        t[3 * (nq + 1) + n] = std::pow(pi, (double)n);
    }
    return t;
}

// ceil(v * 2^32) as used by every integer acceptance test: u < v  <=>  x < ceil(v*2^32) for u = x*2^-32
inline uint64_t thr64(double v)
{
    if (!(v < 1.0)) return 1ull << 32;
    if (!(v > 0.0)) return 0;
    return (uint64_t)std::ceil(v * 4294967296.0);
}
// ceil(v * 2^44): the same test on the 44-bit acceptance uniform of the non-top proposals
inline uint64_t thr44(double v)
{
    if (!(v < 1.0)) return 1ull << 44;
    if (!(v > 0.0)) return 0;
    return (uint64_t)std::ceil(v * 17592186044416.0);
}
inline uint32_t thr32(double v)
{
    const uint64_t t = thr64(v);
    return t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t;
}

// np.linspace(p_bottom, p_top, Nc) and Ladder.p_diff (src/mcmc.py:65,69)
inline void ladder_probabilities(double p_bottom, double p_top, int Nc, std::vector<double> &pl, std::vector<double> &pd)
{
    pl.assign(Nc, p_bottom);
    pd.assign(Nc > 1 ? Nc - 1 : 0, 1.0);
    if (Nc > 1) {
        const double step = (p_top - p_bottom) / (double)(Nc - 1);
        for (int i = 0; i < Nc; ++i) pl[i] = (double)i * step + p_bottom;
        pl[Nc - 1] = p_top;
    }
    for (int i = 0; i + 1 < Nc; ++i) pd[i] = (pl[i] * (1 - pl[i + 1])) / (pl[i + 1] * (1 - pl[i]));
}

inline double chain_factor(double p) { return (p / 3.0) / (1.0 - p); }   // src/mcmc.py:16

// XOR masks of the toric logical operators on the 2-bit packed state
// (toric_model.py:192-223): kind 0 = X on layer-0 row, 1 = Z on layer-0 column,
// 2 = X on layer-1 column, 3 = Z on layer-1 row; position L = identity.
inline std::vector<uint32_t> toric_logical_masks(int L, int W)
{
    const int LL = L * L;
    std::vector<uint32_t> m((size_t)4 * (L + 1) * W, 0u);
    auto set = [&](int kind, int pos, int q, uint32_t op) { m[((size_t)kind * (L + 1) + pos) * W + (q >> 4)] ^= op << ((q & 15) * 2); };
    for (int pos = 0; pos < L; ++pos)
        for (int i = 0; i < L; ++i) {
            set(0, pos, pos * L + i, 1);
            set(1, pos, i * L + pos, 3);
            set(2, pos, LL + i * L + pos, 1);
            set(3, pos, LL + pos * L + i, 3);
        }
    return m;
}

// XOR masks of the xzzx / rotated logical operators, same [4][L+1][W] layout as the toric table
// (kinds 2, 3 unused): kind 0 = X (xzzx: anti-diagonal for every position; rotated: column `pos`),
// kind 1 = Z (xzzx: diagonal; rotated: row `pos`); xzzx_model.py:291-311, rotated_surface_model.py:260-280
inline std::vector<uint32_t> surf_logical_masks(int code, int L, int W)
{
    std::vector<uint32_t> m((size_t)4 * (L + 1) * W, 0u);
    auto set = [&](int kind, int pos, int q, uint32_t op) { m[((size_t)kind * (L + 1) + pos) * W + (q >> 4)] ^= op << ((q & 15) * 2); };
    for (int pos = 0; pos < L; ++pos)
        for (int i = 0; i < L; ++i) {
            if (code == QECMC_XZZX) { set(0, pos, i * L + (L - 1 - i), 1); set(1, pos, i * L + i, 3); }
            else if (code == QECMC_PLANAR) { set(0, pos, pos * L + i, 1); set(1, pos, i * L + pos, 3); }   // row X_pos / column Z_pos of layer 0
            else { set(0, pos, i * L + pos, 1); set(1, pos, pos * L + i, 3); }
        }
    return m;
}

// generator table of the toric code in sweep order: all X-type (row-major), then all Z-type; the four sites of
// toric_model.py:261-269, each as u16 (flat site << 2 | pauli)
inline std::vector<uint32_t> toric_generator_table(int L)
{
    const int LL = L * L;
    std::vector<uint32_t> t((size_t)4 * LL, 0u);
    for (int op = 0; op < 2; ++op)
        for (int r = 0; r < L; ++r)
            for (int c = 0; c < L; ++c) {
                const int rm = (r + L - 1) % L, rp = (r + 1) % L, cm = (c + L - 1) % L, cp = (c + 1) % L;
                const uint32_t pauli = op == 0 ? 1u : 3u;
                uint32_t q[4];
                q[0] = LL + r * L + c; q[1] = r * L + c;
                if (op == 0) { q[2] = LL + r * L + cm; q[3] = rm * L + c; }
                else { q[2] = r * L + cp; q[3] = LL + rp * L + c; }
                const int g = op * LL + r * L + c;
                t[2 * g] = ((q[0] << 2) | pauli) | (((q[1] << 2) | pauli) << 16);
                t[2 * g + 1] = ((q[2] << 2) | pauli) | (((q[3] << 2) | pauli) << 16);
            }
    return t;
}

// generator table of the plaquette codes: entry g = 4 x u16 (site << 2 | pauli), two u32 per generator
inline std::vector<uint32_t> surf_generator_table(int code, int L)
{
    const int n_gen = surf_ngen(code, L);
    std::vector<uint32_t> t((size_t)2 * n_gen, 0u);
    for (int g = 0; g < n_gen; ++g) {
        int row, col, op, sites[4], paulis[4];
        surf_gen_rco(code, L, g, row, col, op);
        const int n = surf_generator(code, L, row, col, op, sites, paulis);
        uint32_t e[4] = {0, 0, 0, 0};
        for (int i = 0; i < n; ++i) e[i] = ((uint32_t)sites[i] << 2) | (uint32_t)paulis[i];
        t[2 * g] = e[0] | (e[1] << 16);
        t[2 * g + 1] = e[2] | (e[3] << 16);
    }
    return t;
}


// The distinct Pauli patterns among the generators (four 2-bit Paulis, site 0 in bits 1:0; 0 = no site) and every generator's
// pattern id: the rows of the biased rules' count-change table and of the plaquette codes' dE table.
inline void generator_patterns(const std::vector<uint32_t> &gt, std::vector<uint8_t> &gen_type, std::vector<uint32_t> &patterns)
{
    gen_type.assign(gt.size() / 2, 0);
    patterns.clear();
    for (size_t g = 0; g < gen_type.size(); ++g) {
        uint32_t ops = 0;
        for (int u = 0; u < 4; ++u) ops |= ((u < 2 ? gt[2 * g] >> (16 * u) : gt[2 * g + 1] >> (16 * (u - 2))) & 3u) << (2 * u);
        size_t t = 0;
        while (t < patterns.size() && patterns[t] != ops) ++t;
        if (t == patterns.size()) patterns.push_back(ops);
        gen_type[g] = (uint8_t)t;
    }
}

// The biased / alpha rules' table of count changes: one row of 256 (the four old 2-bit fields) per Pauli pattern;
// entry = dx + (dz << 10) + ((dx + dy) << 20), wrapping (it is added to packed counts n_x | n_z << 10 | (n_x + n_y) << 20)
inline std::vector<uint32_t> count_change_table(const std::vector<uint32_t> &patterns)
{
    std::vector<uint32_t> lut(256 * patterns.size(), 0u);
    for (size_t t = 0; t < patterns.size(); ++t)
        for (uint32_t F = 0; F < 256; ++F) {
            int d[4] = {0, 0, 0, 0};                                     // change of the counts of I, X, Y, Z
            for (int u = 0; u < 4; ++u) {
                const uint32_t old = (F >> (2 * u)) & 3u, neu = old ^ ((patterns[t] >> (2 * u)) & 3u);
                d[old]--; d[neu]++;
            }
            lut[256 * t + F] = (uint32_t)d[1] + ((uint32_t)d[3] << 10) + ((uint32_t)(d[1] + d[2]) << 20);
        }
    return lut;
}

// ceil(p_diff[i]^d 2^32) for d = 0 .. nq (mcmc.py:149): the swap thresholds [Nc-1][nq+1]
inline std::vector<uint64_t> swap_thresholds(const std::vector<double> &pdiff, int nq)
{
    std::vector<uint64_t> sw((pdiff.empty() ? 1 : pdiff.size()) * (size_t)(nq + 1), 0);
    for (size_t i = 0; i < pdiff.size(); ++i)
        for (int d = 0; d <= nq; ++d) sw[i * (size_t)(nq + 1) + d] = thr64(std::pow(pdiff[i], (double)d));
    return sw;
}

// scan = 3 (QECMC_SCAN_WAVE, ladder_wu.hpp): one 64-byte descriptor per generator (one cache line; 12 dwords used), read with scalar loads by a wavefront whose
// 64 ladders test that generator together.  dwords 0-3: site i as (state word) | (bit shift) << 8; 4-7: the value to xor into
// that word (Pauli << shift; a site that does not exist: 0, on word 0); 8, 9: the error-count change of a site as a 4-entry byte
// table indexed by its old 2-bit field, 4 (1 + change) -- 8 for an identity, 0 for the generator's own Pauli, else 4 -- for the
// generator's first Pauli (dword 8) and its second one, if any (dword 9); 10, 11: per site the byte or-ed into / and-ed with the old
// field to form the table index (4: second table; a missing site: index of a "no change" entry, field masked away).
// 12-15: the same for the alpha rule (below).  Empty if some generator carries three different Paulis (none of the four code models does).
inline std::vector<uint32_t> wave_descriptors(const std::vector<uint32_t> &gt)
{
    const size_t G = gt.size() / 2;
    std::vector<uint32_t> d(16 * G, 0u);
    auto table = [](uint32_t P) { uint32_t t = 0; for (uint32_t f = 0; f < 4; ++f) t |= (f == 0 ? 8u : f == P ? 0u : 4u) << (8 * f); return t; };
    for (size_t g = 0; g < G; ++g) {
        uint32_t pa = 0, pb = 0;
        uint32_t e[4];
        for (int u = 0; u < 4; ++u) {
            e[u] = (u < 2 ? gt[2 * g] >> (16 * u) : gt[2 * g + 1] >> (16 * (u - 2))) & 0xFFFFu;
            const uint32_t P = e[u] & 3u;
            if (!P) continue;
            if (!pa || pa == P) pa = P;
            else if (!pb || pb == P) pb = P;
            else return {};
        }
        if (!pa) return {};
        const uint32_t neutral = pa == 1u ? 2u : 1u;            // a field value that is neither the identity nor Pauli pa: table entry 4
        uint32_t omask = 0, amask = 0;
        for (int u = 0; u < 4; ++u) {
            const uint32_t P = e[u] & 3u, q = e[u] >> 2;
            if (P) {
                d[16 * g + u] = (q >> 4) | (((q & 15u) * 2u) << 8);
                d[16 * g + 4 + u] = P << ((q & 15u) * 2u);
                amask |= 3u << (8 * u);
                if (P != pa) omask |= 4u << (8 * u);
            } else {
                omask |= neutral << (8 * u);
            }
        }
        d[16 * g + 8] = table(pa);
        d[16 * g + 9] = pb ? table(pb) : table(pa);
        d[16 * g + 10] = omask;
        d[16 * g + 11] = amask;
        // the alpha rule (mcmc_alpha.py:31-36 weighs n_z and n_x + n_y): per Pauli the byte 4 ((dz + 1) + 9 (dxy + 1)) of applying it to old field f;
        // the four bytes add up to the byte offset of (D_xy, D_z) in the kernel's 9 x 9 table, a missing site (selector 0x0C: the constant 0)
        // is made up for by the offset in dword 15
        auto atable = [](uint32_t P) {
            uint32_t t = 0;
            for (uint32_t f = 0; f < 4; ++f) {
                const uint32_t fn = f ^ P;
                const int dz = (int)(fn == 3u) - (int)(f == 3u), dxy = (int)(fn == 1u || fn == 2u) - (int)(f == 1u || f == 2u);
                t |= (uint32_t)(4 * ((dz + 1) + 9 * (dxy + 1))) << (8 * f);
            }
            return t;
        };
        uint32_t omask_a = 0, n_real = 0;
        for (int u = 0; u < 4; ++u) {
            const uint32_t P = e[u] & 3u;
            if (P) { ++n_real; if (P != pa) omask_a |= 4u << (8 * u); }
            else omask_a |= 0x0Cu << (8 * u);
        }
        d[16 * g + 12] = atable(pa);
        d[16 * g + 13] = pb ? atable(pb) : atable(pa);
        d[16 * g + 14] = omask_a;
        d[16 * g + 15] = (4u - n_real) * 40u;
    }
    return d;
}

// scan = 2 (QECMC_SCAN_COLOUR): the generators cut into PHASES of mutually disjoint generators (no shared qubit), which one
// wavefront proposes at once.  Greedy colouring in table order -- colour(g) = the smallest colour no earlier generator sharing
// a qubit with g has -- then every colour class, in increasing g, in consecutive chunks of at most 64 (one lane each).
// Returns [n_phases][64] generator indices, 0xFFFF = idle lane.  (The CPU oracle states the same rule on its own.)
inline std::vector<uint16_t> colour_phases(const std::vector<uint32_t> &gt, int &n_phases)
{
    const size_t G = gt.size() / 2;
    auto sites = [&](size_t g, int (&q)[4]) {
        int n = 0;
        for (int u = 0; u < 4; ++u) {
            const uint32_t e = (u < 2 ? gt[2 * g] >> (16 * u) : gt[2 * g + 1] >> (16 * (u - 2))) & 0xFFFFu;
            if (e & 3u) q[n++] = (int)(e >> 2);
        }
        return n;
    };
    std::vector<int> colour(G, 0);
    int n_colours = 0;
    for (size_t g = 0; g < G; ++g) {
        int qg[4];
        const int ng = sites(g, qg);
        std::vector<char> used(n_colours + 1, 0);
        for (size_t h = 0; h < g; ++h) {
            int qh[4];
            const int nh = sites(h, qh);
            bool share = false;
            for (int i = 0; i < ng && !share; ++i)
                for (int j = 0; j < nh; ++j) share |= qg[i] == qh[j];
            if (share) used[colour[h]] = 1;
        }
        int c = 0;
        while (used[c]) ++c;
        colour[g] = c;
        n_colours = std::max(n_colours, c + 1);
    }
    std::vector<uint16_t> ph;
    n_phases = 0;
    for (int c = 0; c < n_colours; ++c) {
        int fill = 64;                                     // lanes used in the current phase (64: none open)
        for (size_t g = 0; g < G; ++g) {
            if (colour[g] != c) continue;
            if (fill == 64) { ph.insert(ph.end(), 64, (uint16_t)0xFFFFu); ++n_phases; fill = 0; }
            ph[(size_t)(n_phases - 1) * 64 + fill++] = (uint16_t)g;
        }
    }
    return ph;
}

}  // namespace tables
}  // namespace qecmc
