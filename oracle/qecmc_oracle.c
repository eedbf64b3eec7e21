/*
 * qecmc_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see qecmc_oracle.h).
 *
 * A plain-C restatement of the reference hot path, written from the reference's
 * behaviour (file:line citations are relative to /root/reference).  It keeps the
 * reference's random-scan semantics, draw order and floating-point formulas so
 * that, fed the same uniform stream, it reproduces the reference bit for bit
 * (fixtures F1/F2/F4 under tests/golden/).
 *
 * Parity status: PINNED by tests/test_oracle_golden.py.
 */
#include "qecmc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1,2,3", */
/* SC'11; Random123 v1.x constants).  Published algorithm, restated here.     */
/* ------------------------------------------------------------------------ */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_rng_init_stream(orc_rng *r, const double *stream, uint64_t len)
{
    memset(r, 0, sizeof *r);
    r->mode = 0; r->stream = stream; r->len = len;
}

void orc_rng_init_philox(orc_rng *r, uint64_t seed, uint32_t syndrome)
{
    memset(r, 0, sizeof *r);
    r->mode = 1; r->seed = seed; r->syndrome = syndrome;
}

/* One uniform in [0,1).  Stream mode ignores the address and just takes the
 * next injected double (the reference has a single global `random` stream).
 * Philox mode: word `w` of block (k, sub) of `stream_id`; the uniform is the
 * `nbits`-bit field that starts `shl` bits below the top of the word, scaled by
 * 2^-nbits (shl = 0, nbits = 32: the whole word). */
#define ORC_SUB_REFINE 4u   /* acceptance-refinement blocks of the non-top proposals */
static double orc_draw_field(orc_rng *r, uint32_t stream_id, uint64_t k, uint32_t sub, int w, int shl, int nbits)
{
    r->consumed++;
    if (r->mode == 0) {
        if (r->pos >= r->len) abort();   /* fixture stream exhausted: a test bug */
        return r->stream[r->pos++];
    }
    const uint32_t *cw;
    if (sub == ORC_SUB_REFINE) {
        if (!(r->c2_valid && r->c2_stream == stream_id && r->c2_k == k && r->c2_sub == sub)) {
            uint32_t ctr[4] = {(uint32_t)k, (uint32_t)((k >> 32) & 0xFFFFu) | (sub << 16), r->syndrome, stream_id};
            uint32_t key[2] = {(uint32_t)r->seed, (uint32_t)(r->seed >> 32)};
            orc_philox4x32_10(ctr, key, r->c2_w);
            r->c2_stream = stream_id; r->c2_k = k; r->c2_sub = sub; r->c2_valid = 1;
        }
        cw = r->c2_w;
    } else {
        if (!(r->c_valid && r->c_stream == stream_id && r->c_k == k && r->c_sub == sub)) {
            uint32_t ctr[4], key[2];
            ctr[0] = (uint32_t)k;
            ctr[1] = (uint32_t)((k >> 32) & 0xFFFFu) | (sub << 16);
            ctr[2] = r->syndrome;
            ctr[3] = stream_id;
            key[0] = (uint32_t)r->seed;
            key[1] = (uint32_t)(r->seed >> 32);
            orc_philox4x32_10(ctr, key, r->c_w);
            r->c_stream = stream_id; r->c_k = k; r->c_sub = sub; r->c_valid = 1;
        }
        cw = r->c_w;
    }
    uint32_t field = (uint32_t)(cw[w] << shl) >> (32 - nbits);
    return (double)field / (double)(1ull << nbits);
}

static double orc_draw(orc_rng *r, uint32_t stream_id, uint64_t k, uint32_t sub, int w)
{
    return orc_draw_field(r, stream_id, k, sub, w, 0, 32);
}

#define ORC_SWAP_STREAM 0x100u
/* Chains updated by the non-top rule (mcmc.py:37-43) draw from the "diagonal" streams: slot c at ladder step T uses stream
 * ORC_DIAG_STREAM + (c + T) mod Nc; the top rule keeps stream Nc - 1.  (Any one-to-one assignment of (slot, step) to
 * (stream, index) is as good as another; this one lets a GPU wave, whose slot moves down by one every step, stay on one stream.) */
#define ORC_DIAG_STREAM 0x400u

/* ------------------------------------------------------------------------ */
/* toric stencils                                                            */
/* ------------------------------------------------------------------------ */
static inline int wrap(int a, int L) { a %= L; return a < 0 ? a + L : a; }

/* one qubit ^= op with the reference's error-count bookkeeping
 * (toric_model.py:275-282 / :208-217) */
static inline int flip_qubit(uint8_t *q, int op)
{
    uint8_t old = *q, neu = (uint8_t)(old ^ op);
    *q = neu;
    if (old && !neu) return -1;
    if (neu && !old) return 1;
    return 0;
}

/* _apply_stabilizer, src/toric_model.py:256-284.  op=1 touches
 * (1,r,c),(1,r,c-1),(0,r,c),(0,r-1,c); op=3 touches (1,r,c),(0,r,c),(0,r,c+1),(1,r+1,c).
 * The reference reads old values from the INPUT matrix and writes into a copy. */
int orc_toric_apply_stabilizer(int L, const uint8_t *in, uint8_t *out, int row, int col, int op)
{
    const int LL = L * L;
    int lay[4], rr[4], cc[4];
    if (op == 1) {
        lay[0] = 1; lay[1] = 1; lay[2] = 0; lay[3] = 0;
        rr[0] = row; rr[1] = row; rr[2] = row; rr[3] = wrap(row - 1, L);
        cc[0] = col; cc[1] = wrap(col - 1, L); cc[2] = col; cc[3] = col;
    } else {
        lay[0] = 1; lay[1] = 0; lay[2] = 0; lay[3] = 1;
        rr[0] = row; rr[1] = row; rr[2] = row; rr[3] = wrap(row + 1, L);
        cc[0] = col; cc[1] = col; cc[2] = wrap(col + 1, L); cc[3] = col;
    }
    if (out != in) memcpy(out, in, (size_t)2 * LL);
    int dE = 0;
    for (int i = 0; i < 4; ++i) {
        int idx = lay[i] * LL + rr[i] * L + cc[i];
        uint8_t old = in == out ? out[idx] : in[idx];
        uint8_t neu = (uint8_t)(old ^ op);
        out[idx] = neu;
        if (old && !neu) dE -= 1;
        else if (neu && !old) dE += 1;
    }
    return dE;
}

/* _apply_logical, src/toric_model.py:179-225.  Layer 0: X on row X_pos, Z on
 * column Z_pos; layer 1 is the transpose (:201-202).  X and Z updates are
 * interleaved per index exactly as in the reference loop (:219-223). */
int orc_toric_apply_logical(int L, const uint8_t *in, uint8_t *out, int op, int layer, int xpos, int zpos)
{
    const int LL = L * L;
    if (out != in) memcpy(out, in, (size_t)2 * LL);
    if (op == 0) return 0;
    int do_x = (op == 1 || op == 2);
    int do_z = (op == 3 || op == 2);
    uint8_t *m = out + layer * LL;
    int dE = 0;
    for (int i = 0; i < L; ++i) {
        if (do_x) {
            int r = xpos, c = i;
            dE += flip_qubit(layer == 0 ? &m[r * L + c] : &m[c * L + r], 1);
        }
        if (do_z) {
            int r = i, c = zpos;
            dE += flip_qubit(layer == 0 ? &m[r * L + c] : &m[c * L + r], 3);
        }
    }
    return dE;
}

/* _count_errors, src/toric_model.py:174-176 */
int64_t orc_count_errors(size_t nq, const uint8_t *in)
{
    int64_t n = 0;
    for (size_t i = 0; i < nq; ++i) n += in[i] != 0;
    return n;
}

/* _define_equivalence_class, src/toric_model.py:317-351 */
int orc_toric_eq_class(int L, const uint8_t *in)
{
    const int LL = L * L;
    int cnt[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int l = 0; l < 2; ++l)
        for (int i = 0; i < LL; ++i) cnt[l][in[l * LL + i] & 3]++;
    int x1 = (cnt[0][1] + cnt[0][2]) % 2, z1 = (cnt[0][3] + cnt[0][2]) % 2;
    int x2 = (cnt[1][1] + cnt[1][2]) % 2, z2 = (cnt[1][3] + cnt[1][2]) % 2;
    return x1 + z1 * 2 + x2 * 4 + z2 * 8;
}

/* _to_class, src/toric_model.py:354-377 */
void orc_toric_to_class(int L, const uint8_t *in, uint8_t *out, int eq)
{
    int diff = eq ^ orc_toric_eq_class(L, in);
    int x = (0xA & diff) >> 1;
    int ops = diff ^ x;
    int ops2 = ops >> 2, ops1 = ops & 3;
    if (out != in) memcpy(out, in, (size_t)2 * L * L);
    orc_toric_apply_logical(L, out, out, ops1, 0, 0, 0);
    orc_toric_apply_logical(L, out, out, ops2, 1, 0, 0);
}

/* Toric_code.syndrom, src/toric_model.py:58-101.  defects_out uint8[2][L][L]:
 * [0] vertex defects (Y/Z parity), [1] plaquette defects (X/Y parity). */
void orc_toric_syndrome(int L, const uint8_t *in, uint8_t *defects_out)
{
    const int LL = L * L;
    const uint8_t *q0 = in, *q1 = in + LL;
    for (int r = 0; r < L; ++r)
        for (int c = 0; c < L; ++c) {
#define YZ(m, rr, cc) (((m)[wrap(rr, L) * L + wrap(cc, L)] == 2) || ((m)[wrap(rr, L) * L + wrap(cc, L)] == 3))
#define XY(m, rr, cc) (((m)[wrap(rr, L) * L + wrap(cc, L)] == 1) || ((m)[wrap(rr, L) * L + wrap(cc, L)] == 2))
            int charge0 = YZ(q0, r, c) ^ YZ(q0, r - 1, c);  /* roll(+1, axis 0)  :66  */
            int charge1 = YZ(q1, r, c) ^ YZ(q1, r, c - 1);  /* roll(+1, axis 1)  :73  */
            int flux0 = XY(q0, r, c) ^ XY(q0, r, c + 1);    /* roll(-1, axis 1)  :86  */
            int flux1 = XY(q1, r, c) ^ XY(q1, r + 1, c);    /* roll(-1, axis 0)  :94  */
#undef YZ
#undef XY
            defects_out[r * L + c] = (uint8_t)(charge0 ^ charge1);
            defects_out[LL + r * L + c] = (uint8_t)(flux0 ^ flux1);
        }
}

/* ------------------------------------------------------------------------ */
/* Chain.update_chain, src/mcmc.py:19-43 (and Chain_biased, mcmc_biased.py:20-59) */
/* ------------------------------------------------------------------------ */
int orc_nq(int code, int L) { return (code == ORC_TORIC || code == ORC_PLANAR) ? 2 * L * L : L * L; }
int orc_ncls(int code) { return code == ORC_TORIC ? 16 : 4; }

int orc_eq_class(int code, int L, const uint8_t *m)
{
    return code == ORC_TORIC ? orc_toric_eq_class(L, m) : orc_surf_eq_class(code, L, m);
}

/* Philox address of a proposal's draws (every code model).  Non-top chains (mcmc.py:38-43) need a generator and an
 * acceptance uniform: proposal k owns ONE word, word k&3 of block (k>>2, sub 1), so a block feeds four proposals.  Its
 * top 20 bits pick the generator; its low 12 bits are the leading bits of the acceptance uniform, which continues with
 * word k&3 of the refinement block (k>>2, sub 4): u = (a12 * 2^32 + w) * 2^-44.  (The refinement word matters only when
 * the 12 leading bits do not decide the comparison, once in 4096 proposals; the GPU computes it on demand.)
 * Top chains (mcmc.py:21-35) in sweep mode keep block (k, 0): word 0 selects logical / stabilizer, word 1 picks the generator,
 * words 1-3 carry a logical operator; the acceptance uniform is word 0 of block (k, 2).  In random scan a top-chain
 * proposal needs no more than two words, so two proposals share a block: words A, B = 2 (k & 1), 2 (k & 1) + 1 of block
 * (k >> 1, sub 5); A[31:16] selects, B picks the generator, and a logical operator's fields are cut from A[15:0] and B
 * (model_random_logical_ex; the biased / alpha rules: top_accept44 below). */
static double nontop_accept(const orc_model *m, orc_rng *rng, uint32_t slot, uint64_t k)
{
    (void)m;
    if (rng->mode == 0) return orc_draw(rng, slot, k, 0, 0);              /* injected stream: the next draw */
    const double hi = orc_draw_field(rng, slot, k >> 2, 1, (int)(k & 3), 20, 12);
    const double lo = orc_draw(rng, slot, k >> 2, ORC_SUB_REFINE, (int)(k & 3));
    rng->consumed--;                                                      /* one uniform */
    return hi + lo * (1.0 / 4096.0);                                      /* exact: 44 bits */
}

/* The top chain under the biased / alpha rules (mcmc_biased.py:32-46, mcmc_alpha.py:42-58) always draws an acceptance uniform.
 * Philox mode: like the depolarizing top chain, proposal k owns words A, B = 2 (k & 1), 2 (k & 1) + 1 of block (k >> 1, sub 5):
 * A[31:16] selects logical / stabilizer, a logical operator's fields are cut from A[15:0] and B[31:16]
 * (model_random_logical_ex, packed), and B is used like a non-top proposal's word: its top 20 bits pick the generator, its low
 * 12 bits lead the 44-bit acceptance uniform that word 2 (k & 1) + 1 of the refinement block (k >> 1, sub 4) completes. */
static double top_accept44(orc_rng *rng, uint32_t slot, uint64_t k)
{
    if (rng->mode == 0) return orc_draw(rng, slot, k, 0, 0);              /* injected stream: the next draw */
    const int wb = 2 * (int)(k & 1) + 1;
    const double hi = orc_draw_field(rng, slot, k >> 1, 5, wb, 20, 12);
    const double lo = orc_draw(rng, slot, k >> 1, ORC_SUB_REFINE, wb);
    rng->consumed--;                                                      /* one uniform */
    return hi + lo * (1.0 / 4096.0);                                      /* exact: 44 bits */
}
static int top_select_logical(orc_rng *rng, uint32_t slot, uint64_t k, double p_logical)
{
    return orc_draw_field(rng, slot, k >> 1, 5, 2 * (int)(k & 1), 0, 16) < p_logical;   /* mcmc.py:23 (a 16-bit uniform) */
}

/* _apply_random_stabilizer: a uniform choice among the stabilizer generators.
 * toric (toric_model.py:287-296): three draws row, col, op -> uniform over the 2L^2 generators.
 * xzzx / rotated (xzzx_model.py:439-452, rotated_surface_model.py:395-408): FIVE draws, always: rows, cols in [0,L-1),
 * rows2 in [0,(L-1)/2), cols2 in [0,4), then `u > phalf` picks the full plaquette (rows, cols), else the half plaquette
 * (rows2, cols2); phalf = 2/(L+1) makes that uniform over the (L-1)^2 + 2(L-1) = L^2 - 1 generators.
 * The injected-stream mode consumes the draws as the reference does.  Philox mode spends ONE word on the same uniform
 * choice, g = floor(x * G / 2^32): toric op = 1 if g < L^2 else 3, (row, col) = divmod(g mod L^2, L); xzzx / rotated
 * g < (L-1)^2 is the full plaquette divmod(g, L-1), otherwise h = g - (L-1)^2 is half plaquette h/4 on side h%4. */
static int model_random_stabilizer(const orc_model *m, const uint8_t *in, uint8_t *out, orc_rng *rng,
                                   uint32_t slot, uint64_t k, int w0)
{
    const int L = m->L;
    const int G = m->code == ORC_TORIC ? 2 * L * L : orc_surf_ngen(m->code, L);
    int g = -1;
    if (rng->mode != 0) {
        /* non-top: the top 20 bits of the proposal's word, g = floor(x20 * G / 2^20); top chain: word 1 of its block */
        /* (w0 == 2: the top chain's packed layout -- two proposals per block (k >> 1, 5), the generator from the second word) */
        /* (w0 == 3: the biased / alpha top chain -- the same packed block, the generator from the top 20 bits of the second word) */
        const double u = w0 == 0 ? orc_draw_field(rng, slot, k >> 2, 1, (int)(k & 3), 0, 20)
                       : w0 == 2 ? orc_draw(rng, slot, k >> 1, 5, 2 * (int)(k & 1) + 1)
                       : w0 == 3 ? orc_draw_field(rng, slot, k >> 1, 5, 2 * (int)(k & 1) + 1, 0, 20) : orc_draw(rng, slot, k, 0, 1);
        g = (int)(u * G);
        rng->consumed += (m->code == ORC_TORIC || m->code == ORC_PLANAR) ? 2 : 4;   /* counted like the reference's three / five draws */
    }
    if (m->code == ORC_TORIC) {
        int row, col, op;
        if (g < 0) {
            row = (int)(orc_draw(rng, slot, k, 0, w0) * L);           /* toric_model.py:291 */
            col = (int)(orc_draw(rng, slot, k, 0, w0 + 1) * L);       /* :292 */
            op = (int)(orc_draw(rng, slot, k, 0, w0 + 2) * 2);        /* :293 */
            if (op == 0) op = 3;
        } else {
            op = g < L * L ? 1 : 3;
            row = (g % (L * L)) / L;
            col = g % L;
        }
        return orc_toric_apply_stabilizer(L, in, out, row, col, op);
    }
    if (g >= 0) {
        int row, col, op;
        orc_surf_gen_rco(m->code, L, g, &row, &col, &op);
        return orc_surf_apply_stabilizer(m->code, L, in, out, row, col, op);
    }
    if (m->code == ORC_PLANAR) {                                                /* planar_model.py:343-352 */
        const int short_side = (int)((L - 1) * orc_draw(rng, slot, k, 0, w0));
        const int long_side = (int)(L * orc_draw(rng, slot, k, 0, w0));
        if (orc_draw(rng, slot, k, 0, w0) < 0.5) return orc_surf_apply_stabilizer(m->code, L, in, out, short_side, long_side, 1);
        return orc_surf_apply_stabilizer(m->code, L, in, out, long_side, short_side, 3);
    }
    int rows = (int)((L - 1) * orc_draw(rng, slot, k, 0, w0));                  /* xzzx_model.py:442-445 */
    int cols = (int)((L - 1) * orc_draw(rng, slot, k, 0, w0));
    int rows2 = (int)(((L - 1) / 2.0) * orc_draw(rng, slot, k, 0, w0));
    int cols2 = (int)(4 * orc_draw(rng, slot, k, 0, w0));
    double phalf = (double)(L * L - (L - 1) * (L - 1) - 1) / (double)(L * L - 1);
    if (orc_draw(rng, slot, k, 0, w0) > phalf)                                  /* :446 */
        return orc_surf_apply_stabilizer(m->code, L, in, out, rows, cols, 1);
    return orc_surf_apply_stabilizer(m->code, L, in, out, rows2, cols2, 3);
}

/* _apply_random_logical.  toric (toric_model.py:228-253): op0, op1 first, then per layer X_pos iff
 * op in {1,2}, Z_pos iff op in {3,2}; Philox addressing keeps the proposal in block (k,0): op0 / op1 =
 * top two bits of words 1 / 2, X_pos of layer 0 / 1 = the low 30 bits of words 1 / 2, Z_pos of layer
 * 0 / 1 = the high / low half of word 3.  xzzx / rotated (xzzx_model.py:340-357,
 * rotated_surface_model.py:331-346): one operator (top two bits of word 1), X_pos iff op in {1,2} (low 30
 * bits of word 1), Z_pos iff op in {3,2} (high half of word 3). */
static int model_random_logical_ex(const orc_model *m, const uint8_t *in, uint8_t *out, orc_rng *rng,
                                   uint32_t slot, uint64_t k, int packed)
{
    const int L = m->L;
    if (m->code == ORC_TORIC && packed) {
        /* the random-scan top chain: words A, B = words 2 (k & 1), 2 (k & 1) + 1 of block (k >> 1, 5).
         * A = select[31:16] | op0[15:14] | op1[13:12] | X_pos0[11:0];  B = Z_pos0[31:21] | X_pos1[20:10] | Z_pos1[9:0] */
        const uint64_t kb = k >> 1;
        const int wa = 2 * (int)(k & 1), wb = wa + 1;
        int ops[2];
        ops[0] = (int)(orc_draw_field(rng, slot, kb, 5, wa, 16, 2) * 4);
        ops[1] = (int)(orc_draw_field(rng, slot, kb, 5, wa, 18, 2) * 4);
        int dE = 0;
        if (out != in) memcpy(out, in, (size_t)2 * L * L);
        for (int layer = 0; layer < 2; ++layer) {
            int op = ops[layer], xpos = 0, zpos = 0;
            if (op == 1 || op == 2) xpos = (int)((layer == 0 ? orc_draw_field(rng, slot, kb, 5, wa, 20, 12) : orc_draw_field(rng, slot, kb, 5, wb, 11, 11)) * L);
            if (op == 3 || op == 2) zpos = (int)((layer == 0 ? orc_draw_field(rng, slot, kb, 5, wb, 0, 11) : orc_draw_field(rng, slot, kb, 5, wb, 22, 10)) * L);
            dE += orc_toric_apply_logical(L, out, out, op, layer, xpos, zpos);
        }
        return dE;
    }
    if (m->code == ORC_TORIC) {
        int ops[2];
        ops[0] = (int)(orc_draw(rng, slot, k, 0, 1) * 4);
        ops[1] = (int)(orc_draw(rng, slot, k, 0, 2) * 4);
        int dE = 0;
        if (out != in) memcpy(out, in, (size_t)2 * L * L);
        for (int layer = 0; layer < 2; ++layer) {
            int op = ops[layer], xpos = 0, zpos = 0;
            if (op == 1 || op == 2) xpos = (int)(orc_draw_field(rng, slot, k, 0, 1 + layer, 2, 30) * L);
            if (op == 3 || op == 2) zpos = (int)(orc_draw_field(rng, slot, k, 0, 3, 16 * layer, 16) * L);
            dE += orc_toric_apply_logical(L, out, out, op, layer, xpos, zpos);
        }
        return dE;
    }
    if (packed) {
        /* plaquette codes, random-scan depolarizing top chain: A = select[31:16] | op[15:14] | X_pos[13:0], Z_pos = B[31:16] */
        const uint64_t kb = k >> 1;
        const int wa = 2 * (int)(k & 1), wb = wa + 1;
        int op = (int)(orc_draw_field(rng, slot, kb, 5, wa, 16, 2) * 4), xpos = 0, zpos = 0;
        if (op == 1 || op == 2) xpos = (int)(orc_draw_field(rng, slot, kb, 5, wa, 18, 14) * L);
        if (op == 3 || op == 2) zpos = (int)(orc_draw_field(rng, slot, kb, 5, wb, 0, 16) * L);
        return orc_surf_apply_logical(m->code, L, in, out, op, xpos, zpos);
    }
    int op = (int)(orc_draw(rng, slot, k, 0, 1) * 4), xpos = 0, zpos = 0;
    if (op == 1 || op == 2) xpos = (int)(orc_draw_field(rng, slot, k, 0, 1, 2, 30) * L);
    if (op == 3 || op == 2) zpos = (int)(orc_draw_field(rng, slot, k, 0, 3, 0, 16) * L);
    return orc_surf_apply_logical(m->code, L, in, out, op, xpos, zpos);
}

static int model_random_logical(const orc_model *m, const uint8_t *in, uint8_t *out, orc_rng *rng, uint32_t slot, uint64_t k)
{
    return model_random_logical_ex(m, in, out, rng, slot, k, 0);
}

/* scan = 1: proposal k of a chain tests generator k mod G, in the fixed order "all X-type (row-major), then all
 * Z-type" (toric) / "all plaquettes (row-major), then the half plaquettes" (xzzx, rotated) */
static int model_sweep_stabilizer(const orc_model *m, const uint8_t *in, uint8_t *out, uint64_t k)
{
    const int L = m->L;
    if (m->code == ORC_TORIC) {
        const int LL = L * L;
        int g = (int)(k % (uint64_t)(2 * LL)), op = 1;
        if (g >= LL) { op = 3; g -= LL; }
        return orc_toric_apply_stabilizer(L, in, out, g / L, g % L, op);
    }
    int row, col, op;
    orc_surf_gen_rco(m->code, L, (int)(k % (uint64_t)orc_surf_ngen(m->code, L)), &row, &col, &op);
    return orc_surf_apply_stabilizer(m->code, L, in, out, row, col, op);
}

/* Systematic-sweep Metropolis (scan = 1).  Non-top chains: generator k mod G, accepted iff u < f^dE with u = word
 * k&3 of Philox block (k>>2, sub 3).  Top chain at p >= 0.75 (every move is accepted): generator k mod G is applied
 * with probability 1/2 (coin = bit k&31 of word (k>>5)&3 of block (k>>7, sub 3); without the coin a full sweep at
 * f = 1 would compose to the identity), and every 8th proposal (k mod 8 == 0) also applies one uniformly random
 * logical operator drawn as in random scan from block (k, 0) -- a single one already randomises the class.  A top
 * chain below p = 0.75 (1-chain ladder) keeps the per-proposal rule: word 0 of block (k,0) selects a logical
 * proposal, otherwise generator k mod G, Metropolis test with block (k,2). */
static void chain_update_sweep(const orc_model *m, uint8_t *state, double p, double p_logical, uint64_t iters,
                               orc_rng *rng, uint32_t slot, uint64_t k0, uint8_t *scratch)
{
    const size_t nq = (size_t)orc_nq(m->code, m->L);
    const double factor = (p / 3.0) / (1.0 - p);
    for (uint64_t j = 0; j < iters; ++j) {
        const uint64_t k = k0 + j;
        if (p_logical != 0 && p >= 0.75) {
            if ((k & 7) == 0) {
                model_random_logical(m, state, scratch, rng, slot, k);
                memcpy(state, scratch, nq);
            }
            model_sweep_stabilizer(m, state, scratch, k);
            if (orc_draw_field(rng, slot, k >> 7, 3, (int)((k >> 5) & 3), 31 - (int)(k & 31), 1) >= 0.5) memcpy(state, scratch, nq);
        } else if (p_logical != 0) {
            int dE;
            if (orc_draw(rng, slot, k, 0, 0) < p_logical) dE = model_random_logical(m, state, scratch, rng, slot, k);
            else dE = model_sweep_stabilizer(m, state, scratch, k);
            if (dE <= 0 || orc_draw(rng, slot, k, 2, 0) < pow(factor, (double)dE)) memcpy(state, scratch, nq);
        } else {
            const int dE = model_sweep_stabilizer(m, state, scratch, k);
            if (orc_draw(rng, slot, k >> 2, 3, (int)(k & 3)) < pow(factor, (double)dE)) memcpy(state, scratch, nq);
        }
    }
}

/* scan = 2 ("colour"): the generators are cut into PHASES of mutually disjoint generators (no shared qubit), and a chain
 * advances one whole phase at a time -- on the GPU one wavefront pass, here the members one after the other, which is the same
 * thing because they commute and their tests do not see each other.  The phases, stated here independently of the library:
 * greedy colouring in sweep order (colour(g) = the smallest colour that no earlier generator sharing a qubit with g has), then
 * every colour class, in increasing g, in consecutive chunks of at most 64.  tab[phase][i] = generator, -1 = none. */
typedef struct { int code, L, n_phases; int *tab; } colour_phases_t;
static colour_phases_t g_phases[4 * 65];            /* one entry per (code, L): four codes, L <= 64 */
static int g_n_phases_cached = 0;

static const colour_phases_t *colour_phases(const orc_model *m)
{
    const colour_phases_t *hit = NULL;
#pragma omp critical(orc_colour_phases)
    {
        for (int i = 0; i < g_n_phases_cached; ++i)
            if (g_phases[i].code == m->code && g_phases[i].L == m->L) hit = &g_phases[i];
        if (!hit && g_n_phases_cached < (int)(sizeof g_phases / sizeof g_phases[0])) {
            const int L = m->L, nq = orc_nq(m->code, L);
            const int G = m->code == ORC_TORIC ? 2 * L * L : orc_surf_ngen(m->code, L);
            uint8_t *zero = calloc((size_t)nq, 1), *pat = malloc((size_t)G * nq);
            int *colour = malloc(sizeof(int) * (size_t)G), n_colours = 0;
            for (int g = 0; g < G; ++g) model_sweep_stabilizer(m, zero, pat + (size_t)g * nq, (uint64_t)g);   /* the generator's own Paulis */
            for (int g = 0; g < G; ++g) {
                int c = 0, again = 1;
                while (again) {
                    again = 0;
                    for (int h = 0; h < g && !again; ++h) {
                        if (colour[h] != c) continue;
                        for (int q = 0; q < nq; ++q)
                            if (pat[(size_t)g * nq + q] && pat[(size_t)h * nq + q]) { again = 1; break; }
                    }
                    if (again) ++c;
                }
                colour[g] = c;
                if (c + 1 > n_colours) n_colours = c + 1;
            }
            int n_ph = 0;
            for (int c = 0; c < n_colours; ++c) {
                int members = 0;
                for (int g = 0; g < G; ++g) members += colour[g] == c;
                n_ph += (members + 63) / 64;
            }
            colour_phases_t *e = &g_phases[g_n_phases_cached];
            e->code = m->code; e->L = L; e->n_phases = n_ph; e->tab = malloc(sizeof(int) * (size_t)n_ph * 64);
            for (int i = 0; i < n_ph * 64; ++i) e->tab[i] = -1;
            int ph = -1;
            for (int c = 0; c < n_colours; ++c) {
                int fill = 64;
                for (int g = 0; g < G; ++g) {
                    if (colour[g] != c) continue;
                    if (fill == 64) { ++ph; fill = 0; }
                    e->tab[ph * 64 + fill++] = g;
                }
            }
            free(zero); free(pat); free(colour);
            ++g_n_phases_cached;
            hit = e;
        }
    }
    return hit;
}

int orc_colour_phases(int code, int L, int *tab_out, int cap)
{
    orc_model m = {code, L, ORC_NOISE_DEPOLARIZING, 0.0, 0.0, 2, 0, {0.0, 0.0, 0.0}};
    const colour_phases_t *ph = colour_phases(&m);
    if (!ph) return -1;
    for (int i = 0; i < ph->n_phases * 64 && i < cap; ++i) tab_out[i] = ph->tab[i];
    return ph->n_phases;
}

/* Colour-parallel Metropolis (scan = 2; `iters` counts phases, k = phase index of the chain).  Top chain with logical moves (it
 * must accept every move, p >= 0.75): before the phase, with probability p_logical (word 0 of block (k, 0)), one uniformly random
 * logical operator drawn from words 1-3 of that block as in the other scans.  Then every generator of phase k mod P, member i
 * drawing u = word k & 3 of block (k >> 2, 8 + i) -- a member's block serves four consecutive phases --: a chain with f < 1 accepts iff
 * u < f^dE (mcmc.py:42); a chain with f >= 1 --
 * where a coin-less sweep would compose to the identity -- applies the generator iff u >= 1/2. */
static void chain_update_colour(const orc_model *m, uint8_t *state, double p, double p_logical, uint64_t iters,
                                orc_rng *rng, uint32_t slot, uint64_t k0, uint8_t *scratch)
{
    const size_t nq = (size_t)orc_nq(m->code, m->L);
    const double factor = (p / 3.0) / (1.0 - p);
    const colour_phases_t *ph = colour_phases(m);
    if (!ph) abort();
    for (uint64_t j = 0; j < iters; ++j) {
        const uint64_t k = k0 + j;
        if (p_logical != 0 && orc_draw(rng, slot, k, 0, 0) < p_logical) {
            model_random_logical(m, state, scratch, rng, slot, k);
            memcpy(state, scratch, nq);
        }
        const int *members = ph->tab + (size_t)(k % (uint64_t)ph->n_phases) * 64;
        for (int i = 0; i < 64; ++i) {
            if (members[i] < 0) continue;
            const int dE = model_sweep_stabilizer(m, state, scratch, (uint64_t)members[i]);
            const double u = orc_draw(rng, slot, k >> 2, 8u + (uint32_t)i, (int)(k & 3));
            const int acc = factor >= 1.0 ? u >= 0.5 : (dE <= 0 || u < pow(factor, (double)dE));
            if (acc) memcpy(state, scratch, nq);
        }
    }
}

/* scan = 3 ("wave"): the reference's random-scan chain (mcmc.py:19-43) with a generator pick SHARED by the 64 ladders whose global
 * indices agree above bit 6 (a GPU wavefront): the pick does not depend on the state (toric_model.py:287-296) and ladders of different
 * syndromes never interact, so every ladder keeps exactly the reference's law; only the noise of different syndromes is correlated.
 * Philox addressing by ladder step T = k0 / iters and proposal j of the step (1 <= iters <= 128; the ladder's streams are its
 * slots': no diagonal streams here), slot c:
 *   pick    S = 128 / iters steps share a window: w = T / S, P = (T % S) iters + j; words A, B = 2 (P & 1), 2 (P & 1) + 1 of block
 *           (64 w + (P >> 1), sub 9) with ctr[2] = syndrome >> 6, stream 0x800 + c:  g = floor(B G / 2^32).  Top chain (p_logical != 0;
 *           it must accept every move, p >= 0.75): logical iff A[31:16] < p_logical 2^16, the operator's fields cut from A[15:0]
 *           and B as in the packed layout of scan = 0;
 *   accept  ten proposals share block (T ceil(iters / 10) + j / 10, sub 10) of the ladder's own index, stream c: field f = j % 10 is
 *           a12 = bits 12 (f & 1) .. + 11 of word f >> 1 (f < 8), byte 3 of word 0 | low nibble of byte 3 of word 1 << 8 (f = 8), the
 *           same of words 2, 3 (f = 9); w32 = word j & 3 of block (T ceil(iters / 4) + (j >> 2), sub 11);
 *           u = (a12 2^32 + w32) 2^-44 < f^dE (mcmc.py:42). */
static void wave_block(const orc_rng *rng, uint32_t synd_word, uint32_t stream, uint64_t k, uint32_t sub, uint32_t out[4])
{
    const uint32_t ctr[4] = {(uint32_t)k, (uint32_t)((k >> 32) & 0xFFFFu) | (sub << 16), synd_word, stream};
    const uint32_t key[2] = {(uint32_t)rng->seed, (uint32_t)(rng->seed >> 32)};
    orc_philox4x32_10(ctr, key, out);
}

/* _apply_random_logical (toric_model.py:228-253, xzzx_model.py:340-357) from the two words of a packed top-chain proposal */
static int logical_from_words(const orc_model *m, const uint8_t *in, uint8_t *out, uint32_t A, uint32_t B)
{
    const int L = m->L;
#define FLD(word, shl, nbits) ((uint32_t)((word) << (shl)) >> (32 - (nbits)))
    if (m->code == ORC_TORIC) {
        /* A = select[31:16] | op0[15:14] | op1[13:12] | X_pos0[11:0];  B = Z_pos0[31:21] | X_pos1[20:10] | Z_pos1[9:0] */
        const int ops[2] = {(int)FLD(A, 16, 2), (int)FLD(A, 18, 2)};
        int dE = 0;
        if (out != in) memcpy(out, in, (size_t)2 * L * L);
        for (int layer = 0; layer < 2; ++layer) {
            const int op = ops[layer];
            int xpos = 0, zpos = 0;
            if (op == 1 || op == 2) xpos = layer == 0 ? (int)((FLD(A, 20, 12) * (uint32_t)L) >> 12) : (int)((FLD(B, 11, 11) * (uint32_t)L) >> 11);
            if (op == 3 || op == 2) zpos = layer == 0 ? (int)((FLD(B, 0, 11) * (uint32_t)L) >> 11) : (int)((FLD(B, 22, 10) * (uint32_t)L) >> 10);
            dE += orc_toric_apply_logical(L, out, out, op, layer, xpos, zpos);
        }
        return dE;
    }
    /* plaquette codes: A = select[31:16] | op[15:14] | X_pos[13:0], Z_pos = B[31:16] */
    const int op = (int)FLD(A, 16, 2);
    int xpos = 0, zpos = 0;
    if (op == 1 || op == 2) xpos = (int)((FLD(A, 18, 14) * (uint32_t)L) >> 14);
    if (op == 3 || op == 2) zpos = (int)((FLD(B, 0, 16) * (uint32_t)L) >> 16);
#undef FLD
    return orc_surf_apply_logical(m->code, L, in, out, op, xpos, zpos);
}

/* (work-queue runs, orc_pteq_wave_queue below: the pick belongs to the lane's POSITION -- group rng->wave_group, the workgroup's own
 *  step counter = the ladder's step + rng->wave_t0 -- while the acceptance uniforms stay the ladder's own) */
static void chain_update_wave(const orc_model *m, uint8_t *state, double p, double p_logical, uint64_t iters,
                              orc_rng *rng, uint32_t slot, uint64_t k0, uint8_t *scratch)
{
    const size_t nq = (size_t)orc_nq(m->code, m->L);
    const double factor = (p / 3.0) / (1.0 - p);
    const int G = m->code == ORC_TORIC ? 2 * m->L * m->L : orc_surf_ngen(m->code, m->L);
    if (rng->mode == 0) abort();                       /* a Philox-mode rule */
    if (p_logical != 0 && !(p >= 0.75)) abort();       /* the top chain of scan = 3 accepts every move */
    if (iters < 1 || iters > 128) abort();
    const uint64_t T = k0 / iters, S = 128 / iters, nch = (iters + 9) / 10, nc4 = (iters + 3) / 4;
    const uint64_t Tp = T + (rng->wave_override ? rng->wave_t0 : 0);                 /* the step that addresses the pick */
    const uint32_t group = rng->wave_override ? rng->wave_group : rng->syndrome >> 6;
    for (uint64_t j = 0; j < iters; ++j) {
        const uint64_t P = (Tp % S) * iters + j;
        uint32_t pw[4];
        wave_block(rng, group, 0x800u + slot, (Tp / S) * 64 + (P >> 1), 9u, pw);
        const uint32_t A = pw[2 * (P & 1)], B = pw[2 * (P & 1) + 1];
        const int g = (int)(((uint64_t)B * (uint32_t)G) >> 32);
        rng->consumed += 2;
        if (p_logical != 0) {                                      /* mcmc.py:20-31 at p >= 0.75 */
            if ((double)(A >> 16) / 65536.0 < p_logical) logical_from_words(m, state, scratch, A, B);
            else model_sweep_stabilizer(m, state, scratch, (uint64_t)g);
            memcpy(state, scratch, nq);
            continue;
        }
        const int dE = model_sweep_stabilizer(m, state, scratch, (uint64_t)g);   /* :38-40 */
        uint32_t aw[4], rw[4];
        wave_block(rng, rng->syndrome, slot, T * nch + j / 10, 10u, aw);
        wave_block(rng, rng->syndrome, slot, T * nc4 + (j >> 2), 11u, rw);
        const int f = (int)(j % 10);
        const uint32_t a12 = f < 8 ? (aw[f >> 1] >> (12 * (f & 1))) & 0xFFFu
                                   : ((aw[f == 8 ? 0 : 2] >> 24) | (((aw[f == 8 ? 1 : 3] >> 24) & 0xFu) << 8));
        const double u = ((double)a12 * 4294967296.0 + (double)rw[j & 3]) / 17592186044416.0;   /* 44 bits: exact */
        if (u < pow(factor, (double)dE)) memcpy(state, scratch, nq);                              /* :42 */
    }
}

/* p_x^nx p_y^ny p_z^nz p_I^(num-nx-ny-nz), mcmc_biased.py:31,43 (left-to-right products of pow()) */
static double biased_weight(const uint8_t *s, int nq, double px, double py, double pz)
{
    int nx = 0, ny = 0, nz = 0;
    for (int i = 0; i < nq; ++i) { nx += s[i] == 1; ny += s[i] == 2; nz += s[i] == 3; }
    return pow(px, (double)nx) * pow(py, (double)ny) * pow(pz, (double)nz) * pow(1 - px - py - pz, (double)(nq - nx - ny - nz));
}

/* scan = 2 (colour phases) under the biased / alpha rules.  The members of a phase are tested "at once", so the rule cannot carry Q3's
 * frozen p_b (a member's ratio would depend on what the members before it did): every generator is a Metropolis move for the noise
 * model's own weight, u < w(s ^ g) / w(s) = (px / pI)^dxy (pz / pI)^dz with (dxy, dz) the change of n_x + n_y and n_z of that generator
 * alone (px = py in both models; mcmc_biased.py:25-31, mcmc_alpha.py:31-36) -- the law the reference's rule has at iters = 1, where Q3 is
 * vacuous.  `coin`: a rung whose ratios are all 1 (Ladder_alpha's top, pz_tilde = 1) applies every generator with probability 1/2 and
 * its logical operators unseen, like the depolarizing top rung; any other top rung tests a logical operator like every move,
 * u = word 0 of block (k, 1) < w(new) / w(old) on the power tables' products (biased_weight).  Returns whether a move was accepted. */
static int chain_update_colour_rule(const orc_model *m, uint8_t *state, double px, double py, double pz, int coin, double p_logical,
                                    uint64_t iters, orc_rng *rng, uint32_t slot, uint64_t k0, uint8_t *scratch)
{
    const size_t nq = (size_t)orc_nq(m->code, m->L);
    const double pi = 1 - px - py - pz;
    const double fxy = px / pi, fz = pz / pi;
    const colour_phases_t *ph = colour_phases(m);
    int accepted = 0;
    if (!ph) abort();
    for (uint64_t j = 0; j < iters; ++j) {
        const uint64_t k = k0 + j;
        if (p_logical != 0 && orc_draw(rng, slot, k, 0, 0) < p_logical) {
            model_random_logical(m, state, scratch, rng, slot, k);
            if (coin || orc_draw(rng, slot, k, 1, 0) < biased_weight(scratch, (int)nq, px, py, pz) / biased_weight(state, (int)nq, px, py, pz)) {
                memcpy(state, scratch, nq);
                accepted = 1;
            }
        }
        const int *members = ph->tab + (size_t)(k % (uint64_t)ph->n_phases) * 64;
        for (int i = 0; i < 64; ++i) {
            if (members[i] < 0) continue;
            model_sweep_stabilizer(m, state, scratch, (uint64_t)members[i]);
            const double u = orc_draw(rng, slot, k >> 2, 8u + (uint32_t)i, (int)(k & 3));
            int acc;
            if (coin) acc = u >= 0.5;
            else {
                int dz = 0, dxy = 0;
                for (size_t q = 0; q < nq; ++q) {
                    dz += (scratch[q] == 3) - (state[q] == 3);
                    dxy += (scratch[q] == 1 || scratch[q] == 2) - (state[q] == 1 || state[q] == 2);
                }
                acc = u < pow(fxy, (double)dxy) * pow(fz, (double)dz);
            }
            if (acc) { memcpy(state, scratch, nq); accepted = 1; }
        }
    }
    return accepted;
}

/* exp(y) for y <= 0 from IEEE add / multiply / fma only, so the GPU (same operation sequence) returns the same bits.
 * y >= 0 returns 1 (the callers only need to know the value is >= 1); results below 2^-1022 flush to 0. */
double orc_det_exp(double y)
{
    if (!(y < 0.0)) return 1.0;
    if (y < -745.0) return 0.0;
    const double t = y * 1.4426950408889634;                 /* 1/ln 2 */
    const long long k = (long long)(t - 0.5);                /* round to nearest (t < 0), truncating cast */
    double r = fma(-(double)k, 6.93147180369123816490e-01, y);      /* ln2 high part */
    r = fma(-(double)k, 1.90821492927058770002e-10, r);              /* ln2 low part */
    double q = 1.0 / 6227020800.0;                           /* Horner, 1/13! ... 1/0! */
    q = fma(q, r, 1.0 / 479001600.0);
    q = fma(q, r, 1.0 / 39916800.0);
    q = fma(q, r, 1.0 / 3628800.0);
    q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);
    q = fma(q, r, 1.0 / 720.0);
    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    if (k < -1022) return 0.0;
    union { uint64_t u; double d; } sc;
    sc.u = (uint64_t)(k + 1023) << 52;                       /* 2^k */
    return q * sc.d;
}

/* scan = 3 under the alpha rule (mcmc_alpha.py:27-70): the picks of chain_update_wave -- shared by the 64 ladders of a wavefront --, the
 * ladder's own 44-bit acceptance uniform against p_n / p_b with p_b frozen at loop entry (Q3); the top rung (pz_tilde = 1: every weight
 * is 0.25^num, the ratio exactly 1) accepts every move.  Returns whether a move was accepted; *n_eff follows (:58,:70). */
static int chain_update_wave_alpha(const orc_model *m, uint8_t *state, double pz_tilde, double p_logical, uint64_t iters,
                                   orc_rng *rng, uint32_t slot, uint64_t k0, uint8_t *scratch, double *n_eff)
{
    const size_t nq = (size_t)orc_nq(m->code, m->L);
    const double alpha = m->alpha;
    const double p_tilde = pz_tilde + 2 * pow(pz_tilde, alpha);
    const double p = p_tilde / (1 + p_tilde);
    const double pz = pz_tilde * (1 - p), px = pow(pz_tilde, alpha) * (1 - p), py = px;
    const double pb = biased_weight(state, (int)nq, px, py, pz);
    const int G = orc_surf_ngen(m->code, m->L);
    if (rng->mode == 0 || m->code == ORC_TORIC) abort();
    if (p_logical != 0 && !(pz_tilde >= 1.0)) abort();
    if (iters < 1 || iters > 128) abort();
    const uint64_t T = k0 / iters, S = 128 / iters, nch = (iters + 9) / 10, nc4 = (iters + 3) / 4;
    const uint64_t Tp = T + (rng->wave_override ? rng->wave_t0 : 0);
    const uint32_t group = rng->wave_override ? rng->wave_group : rng->syndrome >> 6;
    int accepted = 0;
    for (uint64_t j = 0; j < iters; ++j) {
        const uint64_t P = (Tp % S) * iters + j;
        uint32_t pw[4];
        wave_block(rng, group, 0x800u + slot, (Tp / S) * 64 + (P >> 1), 9u, pw);
        const uint32_t A = pw[2 * (P & 1)], B = pw[2 * (P & 1) + 1];
        const int g = (int)(((uint64_t)B * (uint32_t)G) >> 32);
        rng->consumed += 2;
        double u = 0.0;                                                       /* (the top rung: any u < 1) */
        if (p_logical != 0 && (double)(A >> 16) / 65536.0 < p_logical) logical_from_words(m, state, scratch, A, B);
        else model_sweep_stabilizer(m, state, scratch, (uint64_t)g);
        if (p_logical == 0) {
            uint32_t aw[4], rw[4];
            wave_block(rng, rng->syndrome, slot, T * nch + j / 10, 10u, aw);
            wave_block(rng, rng->syndrome, slot, T * nc4 + (j >> 2), 11u, rw);
            const int f = (int)(j % 10);
            const uint32_t a12 = f < 8 ? (aw[f >> 1] >> (12 * (f & 1))) & 0xFFFu
                                       : ((aw[f == 8 ? 0 : 2] >> 24) | (((aw[f == 8 ? 1 : 3] >> 24) & 0xFu) << 8));
            u = ((double)a12 * 4294967296.0 + (double)rw[j & 3]) / 17592186044416.0;
        }
        if (u < biased_weight(scratch, (int)nq, px, py, pz) / pb) {
            memcpy(state, scratch, nq);
            int nx = 0, ny = 0, nz = 0;
            for (size_t i = 0; i < nq; ++i) { nx += state[i] == 1; ny += state[i] == 2; nz += state[i] == 3; }
            *n_eff = nz + alpha * (nx + ny);
            accepted = 1;
        }
    }
    return accepted;
}

/* Chain_alpha.update_chain, mcmc_alpha.py:27-70: the biased rule with (p_x, p_y, p_z) derived from (pz_tilde, alpha),
 * p_b frozen at loop entry (Q3), and n_eff = n_z + alpha (n_x + n_y) refreshed on every accepted move (:58,:70). */
int orc_chain_update_alpha(const orc_model *m, uint8_t *state, double pz_tilde, double p_logical, uint64_t iters,
                               orc_rng *rng, uint32_t slot, uint64_t k0, uint8_t *scratch, double *n_eff)
{
    const size_t nq = (size_t)orc_nq(m->code, m->L);
    const double alpha = m->alpha;
    const double p_tilde = pz_tilde + 2 * pow(pz_tilde, alpha);                         /* :32 */
    const double p = p_tilde / (1 + p_tilde);                                           /* :34 */
    const double pz = pz_tilde * (1 - p), px = pow(pz_tilde, alpha) * (1 - p), py = px; /* :35-36 */
    if (m->scan == 3) return chain_update_wave_alpha(m, state, pz_tilde, p_logical, iters, rng, slot, k0, scratch, n_eff);
    if (m->scan == 2) {
        /* (the attribute follows an accepted move, :58,:70: after the call it is the final configuration's if any move was accepted) */
        const int acc = chain_update_colour_rule(m, state, px, py, pz, pz_tilde >= 1.0, p_logical, iters, rng, slot, k0, scratch);
        if (acc) {
            int nx = 0, ny = 0, nz = 0;
            for (size_t i = 0; i < nq; ++i) { nx += state[i] == 1; ny += state[i] == 2; nz += state[i] == 3; }
            *n_eff = nz + alpha * (nx + ny);
        }
        return acc;
    }
    const double pb = biased_weight(state, (int)nq, px, py, pz);                        /* :38-41 */
    int accepted = 0;
    for (uint64_t j = 0; j < iters; ++j) {
        const uint64_t k = k0 + j;
        double u;
        if (p_logical != 0) {
            if (top_select_logical(rng, slot, k, p_logical)) model_random_logical_ex(m, state, scratch, rng, slot, k, 1);
            else model_random_stabilizer(m, state, scratch, rng, slot, k, 3);
            u = top_accept44(rng, slot, k);
        } else {
            model_random_stabilizer(m, state, scratch, rng, slot, k, 0);
            u = nontop_accept(m, rng, slot, k);
        }
        if (u < biased_weight(scratch, (int)nq, px, py, pz) / pb) {
            memcpy(state, scratch, nq);
            int nx = 0, ny = 0, nz = 0;
            for (size_t i = 0; i < nq; ++i) { nx += state[i] == 1; ny += state[i] == 2; nz += state[i] == 3; }
            *n_eff = nz + alpha * (nx + ny);
            accepted = 1;
        }
    }
    return accepted;
}

void orc_chain_update(const orc_model *m, uint8_t *state, double p, double p_logical, uint64_t iters,
                      orc_rng *rng, uint32_t slot, uint64_t k0, uint8_t *scratch)
{
    const size_t nq = (size_t)orc_nq(m->code, m->L);
    if (m->scan == 1 && m->noise == ORC_NOISE_DEPOLARIZING) {
        chain_update_sweep(m, state, p, p_logical, iters, rng, slot, k0, scratch);
        return;
    }
    if (m->scan == 2 && m->noise == ORC_NOISE_DEPOLARIZING) {
        chain_update_colour(m, state, p, p_logical, iters, rng, slot, k0, scratch);
        return;
    }
    if (m->scan == 3 && m->noise == ORC_NOISE_DEPOLARIZING) {
        chain_update_wave(m, state, p, p_logical, iters, rng, slot, k0, scratch);
        return;
    }
    if (m->noise == ORC_NOISE_BIASED) {
        /* Chain_biased.update_chain, mcmc_biased.py:20-59.  pb is computed ONCE before the loop and never
         * refreshed after an accept (reference quirk Q3, reproduced). */
        const double eta = m->eta;
        const double pz = p * eta / (eta + 1), px = p / (2 * (eta + 1)), py = px;      /* :25-27 */
        if (m->scan == 2) { chain_update_colour_rule(m, state, px, py, pz, 0, p_logical, iters, rng, slot, k0, scratch); return; }
        const double pb = biased_weight(state, (int)nq, px, py, pz);                    /* :28-31 */
        for (uint64_t j = 0; j < iters; ++j) {
            uint64_t k = k0 + j;
            double u;
            if (p_logical != 0) {                                                       /* :32-46 */
                if (top_select_logical(rng, slot, k, p_logical)) model_random_logical_ex(m, state, scratch, rng, slot, k, 1);
                else model_random_stabilizer(m, state, scratch, rng, slot, k, 3);
                const double pn = biased_weight(scratch, (int)nq, px, py, pz);
                u = top_accept44(rng, slot, k);
                if (u < pn / pb) memcpy(state, scratch, nq);
            } else {                                                                    /* :49-59 */
                model_random_stabilizer(m, state, scratch, rng, slot, k, 0);
                const double pn = biased_weight(scratch, (int)nq, px, py, pz);
                u = nontop_accept(m, rng, slot, k);
                if (u < pn / pb) memcpy(state, scratch, nq);
            }
        }
        return;
    }
    if (m->noise == ORC_NOISE_XYZ) {
        /* Chain_xyz.update_chain_fast -> _update_chain_fast_xyz, mcmc.py:112-114,162-173 (no logical branch) */
        const double tot = (m->pxyz[0] + m->pxyz[1]) + m->pxyz[2];                      /* p_xyz.sum() */
        const double f[3] = {m->pxyz[0] / (1.0 - tot), m->pxyz[1] / (1.0 - tot), m->pxyz[2] / (1.0 - tot)};   /* :110 */
        for (uint64_t j = 0; j < iters; ++j) {
            uint64_t k = k0 + j;
            model_random_stabilizer(m, state, scratch, rng, slot, k, 0);
            int c0[4] = {0, 0, 0, 0}, c1[4] = {0, 0, 0, 0};
            for (size_t q = 0; q < nq; ++q) { c0[state[q]]++; c1[scratch[q]]++; }      /* _count_errors_xyz, planar_model.py:225-229 */
            /* (factors ** qubit_errors_change).prod(), :170 */
            const double w = (pow(f[0], (double)(c1[1] - c0[1])) * pow(f[1], (double)(c1[2] - c0[2]))) * pow(f[2], (double)(c1[3] - c0[3]));
            if (nontop_accept(m, rng, slot, k) < w) memcpy(state, scratch, nq);
        }
        return;
    }
    const double factor = (p / 3.0) / (1.0 - p);                  /* mcmc.py:16 */
    if (p_logical != 0) {                                          /* mcmc.py:20 */
        for (uint64_t j = 0; j < iters; ++j) {
            uint64_t k = k0 + j;
            int dE;
            /* the packed layout (two proposals per block): the select is a 16-bit uniform, A[31:16] */
            const int packed = 1;
            const double usel = packed ? orc_draw_field(rng, slot, k >> 1, 5, 2 * (int)(k & 1), 0, 16) : orc_draw(rng, slot, k, 0, 0);
            if (usel < p_logical)                                  /* mcmc.py:23 */
                dE = model_random_logical_ex(m, state, scratch, rng, slot, k, packed);
            else
                dE = model_random_stabilizer(m, state, scratch, rng, slot, k, packed ? 2 : 1);
            if (p >= 0.75 || dE <= 0) {                            /* mcmc.py:30 */
                memcpy(state, scratch, nq);
                continue;
            }
            if (orc_draw(rng, slot, k, 2, 0) < pow(factor, (double)dE))   /* mcmc.py:34 */
                memcpy(state, scratch, nq);
        }
    } else {
        for (uint64_t j = 0; j < iters; ++j) {                     /* mcmc.py:38 */
            uint64_t k = k0 + j;
            int dE = model_random_stabilizer(m, state, scratch, rng, slot, k, 0);
            if (nontop_accept(m, rng, slot, k) < pow(factor, (double)dE))   /* mcmc.py:42 */
                memcpy(state, scratch, nq);
        }
    }
}

void orc_toric_chain_update(int L, uint8_t *state, double p, double p_logical, uint64_t iters,
                            orc_rng *rng, uint32_t slot, uint64_t k0, uint8_t *scratch)
{
    orc_model m = {ORC_TORIC, L, ORC_NOISE_DEPOLARIZING, 0.0, 0.0, 0, 0, {0.0, 0.0, 0.0}};
    orc_chain_update(&m, state, p, p_logical, iters, rng, slot, k0, scratch);
}

/* ------------------------------------------------------------------------ */
/* Ladder, src/mcmc.py:49-103 (Ladder_biased, mcmc_biased.py:66-124)          */
/* ------------------------------------------------------------------------ */
static void fill_ladder_p(double p_bottom, double p_top, int Nc, double *p_ladder, double *p_diff)
{
    /* np.linspace(p_bottom, p_top, Nc): y[i] = i*step + start, y[-1] = stop */
    if (Nc == 1) {
        p_ladder[0] = p_bottom;
        return;
    }
    double step = (p_top - p_bottom) / (double)(Nc - 1);
    for (int i = 0; i < Nc; ++i) p_ladder[i] = (double)i * step + p_bottom;
    p_ladder[Nc - 1] = p_top;
    for (int i = 0; i < Nc - 1; ++i)                               /* mcmc.py:69 */
        p_diff[i] = (p_ladder[i] * (1 - p_ladder[i + 1])) / (p_ladder[i + 1] * (1 - p_ladder[i]));
}

orc_ladder *orc_ladder_new(const orc_model *m, const uint8_t *init, double p_bottom, int Nc, double p_logical)
{
    orc_ladder *ld = (orc_ladder *)calloc(1, sizeof *ld);
    ld->model = *m;
    ld->L = m->L; ld->Nc = Nc; ld->nq = orc_nq(m->code, m->L); ld->p_logical = p_logical;
    ld->p_ladder = (double *)calloc((size_t)Nc, sizeof(double));
    ld->p_diff = (double *)calloc((size_t)(Nc > 1 ? Nc - 1 : 1), sizeof(double));
    ld->states = (uint8_t *)malloc((size_t)Nc * ld->nq);
    ld->flags = (uint8_t *)calloc((size_t)Nc, 1);
    ld->scratch = (uint8_t *)malloc((size_t)ld->nq);
    ld->swap_acc = (uint64_t *)calloc((size_t)(Nc > 1 ? Nc - 1 : 1), sizeof(uint64_t));
    ld->nerr_sum = (uint64_t *)calloc((size_t)Nc, sizeof(uint64_t));
    ld->n_eff = (double *)calloc((size_t)Nc, sizeof(double));
    {   /* Chain_alpha.__init__, mcmc_alpha.py:18-22 */
        int nx = 0, ny = 0, nz = 0;
        for (int i = 0; i < ld->nq; ++i) { nx += init[i] == 1; ny += init[i] == 2; nz += init[i] == 3; }
        ld->n_eff_cnt = (uint32_t *)calloc((size_t)Nc * 2, sizeof(uint32_t));
        for (int c = 0; c < Nc; ++c) {
            ld->n_eff[c] = nz + m->alpha * (nx + ny);
            ld->n_eff_cnt[2 * c] = (uint32_t)nz; ld->n_eff_cnt[2 * c + 1] = (uint32_t)(nx + ny);
        }
    }
    /* p_top = 0.75 (mcmc.py:62), (eta+1)/(2 eta+1) (mcmc_biased.py:81) or pz_tilde_top = 1 (mcmc_alpha.py:94) */
    const double p_top = m->noise == ORC_NOISE_BIASED ? (m->eta + 1) / (2 * m->eta + 1) : m->noise == ORC_NOISE_ALPHA ? 1.0 : 0.75;
    fill_ladder_p(p_bottom, p_top, Nc, ld->p_ladder, ld->p_diff);    /* mcmc.py:62-69 */
    for (int c = 0; c < Nc; ++c) memcpy(ld->states + (size_t)c * ld->nq, init, (size_t)ld->nq); /* :72 */
    ld->flags[Nc - 1] = 1;                                          /* mcmc.py:75 */
    return ld;
}

orc_ladder *orc_toric_ladder_new(int L, const uint8_t *init, double p_bottom, int Nc, double p_logical)
{
    orc_model m = {ORC_TORIC, L, ORC_NOISE_DEPOLARIZING, 0.0, 0.0, 0, 0, {0.0, 0.0, 0.0}};
    return orc_ladder_new(&m, init, p_bottom, Nc, p_logical);
}

void orc_ladder_free(orc_ladder *ld)
{
    if (!ld) return;
    free(ld->p_ladder); free(ld->p_diff); free(ld->states); free(ld->flags); free(ld->scratch); free(ld->n_eff); free(ld->n_eff_cnt);
    free(ld->swap_acc); free(ld->nerr_sum);
    free(ld);
}

/* Ladder.step(iters), src/mcmc.py:94-103 (mcmc_biased.py:115-124) */
void orc_ladder_step(orc_ladder *ld, uint64_t iters, orc_rng *rng)
{
    const int Nc = ld->Nc, nq = ld->nq;
    const uint64_t k0 = ld->step_index * iters;
    const int is_alpha = ld->model.noise == ORC_NOISE_ALPHA;
    for (int c = 0; c < Nc; ++c) {                                  /* update_ladder :81-83 */
        const double pl = c == Nc - 1 ? ld->p_logical : 0.0;
        const uint32_t strm = (pl != 0 || ld->model.scan == 3) ? (uint32_t)c : ORC_DIAG_STREAM + (uint32_t)(((uint64_t)c + ld->step_index) % (uint64_t)Nc);
        if (is_alpha) {
            if (orc_chain_update_alpha(&ld->model, ld->states + (size_t)c * nq, ld->p_ladder[c], pl,
                                       iters, rng, strm, k0, ld->scratch, &ld->n_eff[c])) {
                const uint8_t *st = ld->states + (size_t)c * nq;         /* the counts n_eff was formed from */
                uint32_t nz = 0, nxy = 0;
                for (int q = 0; q < (int)nq; ++q) { nz += st[q] == 3; nxy += st[q] == 1 || st[q] == 2; }
                ld->n_eff_cnt[2 * c] = nz; ld->n_eff_cnt[2 * c + 1] = nxy;
            }
        } else
            orc_chain_update(&ld->model, ld->states + (size_t)c * nq, ld->p_ladder[c], pl, iters, rng, strm, k0, ld->scratch);
    }
    for (int i = Nc - 2; i >= 0; --i) {                             /* :96 */
        int64_t ne_lo = orc_count_errors((size_t)nq, ld->states + (size_t)i * nq);
        int64_t ne_hi = orc_count_errors((size_t)nq, ld->states + (size_t)(i + 1) * nq);
        int flip;
        /* _r_flip: mcmc.py:146 skips the draw when ne_hi < ne_lo; mcmc_biased.py:154-156 always draws
         * (rel_p ** negative > 1, so the outcome is the same; only the stream position differs) */
        if (is_alpha) {
            /* Ladder_alpha.r_flip, mcmc_alpha.py:118-123: the chains' n_eff attributes -- which do NOT travel with
             * the codes when they are swapped (Q4) -- and the ratio of the two pz_tilde's; always draws */
            const double u = orc_draw(rng, ORC_SWAP_STREAM, ld->step_index, (uint32_t)i >> 2, i & 3);
            const double base = ld->p_ladder[i] / ld->p_ladder[i + 1], e = ld->n_eff[i + 1] - ld->n_eff[i];
            flip = ld->model.det_pow ? u < orc_det_exp(e * log(base)) : u < pow(base, e);
        } else
        if (ld->model.noise != ORC_NOISE_BIASED && ne_hi < ne_lo) flip = 1;
        else flip = orc_draw(rng, ORC_SWAP_STREAM, ld->step_index, (uint32_t)i >> 2, i & 3)
                    < pow(ld->p_diff[i], (double)(ne_hi - ne_lo));  /* :149 */
        ld->swap_acc[i] += (uint64_t)(flip != 0);
        if (flip) {                                                 /* :98-99 */
            memcpy(ld->scratch, ld->states + (size_t)i * nq, (size_t)nq);
            memcpy(ld->states + (size_t)i * nq, ld->states + (size_t)(i + 1) * nq, (size_t)nq);
            memcpy(ld->states + (size_t)(i + 1) * nq, ld->scratch, (size_t)nq);
            uint8_t f = ld->flags[i]; ld->flags[i] = ld->flags[i + 1]; ld->flags[i + 1] = f;
        }
    }
    ld->flags[Nc - 1] = 1;                                          /* :100 */
    if (ld->flags[0] == 1) { ld->tops0++; ld->flags[0] = 0; }       /* :101-103 */
    for (int c = 0; c < Nc; ++c) ld->nerr_sum[c] += (uint64_t)orc_count_errors((size_t)nq, ld->states + (size_t)c * nq);
    ld->step_index++;
}

void orc_toric_ladder_step(orc_ladder *ld, uint64_t iters, orc_rng *rng) { orc_ladder_step(ld, iters, rng); }

/* ------------------------------------------------------------------------ */
/* decoders.PTEQ, decoders.py:25-89, and conv_crit_error_based_PT :93-105     */
/* (PTEQ_biased, decoders_biasednoise.py:28-90, is the same loop around       */
/*  Ladder_biased)                                                            */
/* ------------------------------------------------------------------------ */
static double mean_range(const double *a, uint64_t lo, uint64_t hi)
{
    /* np.average of integer-valued doubles: the sum is exact, order irrelevant.
     * Empty slice -> NaN (numpy warns and returns nan; comparisons are False). */
    if (hi <= lo) return NAN;
    double s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += a[i];
    return s / (double)(hi - lo);
}

static double sum_range(const double *a, uint64_t lo, uint64_t hi)
{
    double s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += a[i];      /* integer-valued: exact */
    return s;
}

/* decoders.PTEQ's loop (decoders.py:55-82) one ladder step at a time: what orc_pteq runs to the end, and what the work-queue
 * restatement (orc_pteq_wave_queue) interleaves over the 64 lanes of a workgroup */
typedef struct pteq_run {
    orc_ladder *ld;
    const orc_model *m;
    int ncls, SEQ, TOPS, tops_burn, conv_mode, det_series, converged;
    double eps;
    uint64_t steps, iters, step, since_burn, resulting_burn_in, recorded, conv_start, conv_streak;
    uint32_t eq[16];
    double *series, *series_z, *series_xy;
} pteq_run;

static void pteq_run_init(pteq_run *r, const orc_model *m, const uint8_t *init, double p, int Nc, int SEQ, int TOPS, int tops_burn,
                          double eps, uint64_t steps, uint64_t iters, int conv_mode)
{
    memset(r, 0, sizeof *r);
    r->ld = orc_ladder_new(m, init, p, Nc, 0.5);                    /* decoders.py:52 */
    r->m = m; r->ncls = orc_ncls(m->code); r->SEQ = SEQ; r->TOPS = TOPS; r->tops_burn = tops_burn; r->conv_mode = conv_mode;
    r->eps = eps; r->steps = steps; r->iters = iters;
    r->series = conv_mode ? (double *)calloc((size_t)steps, sizeof(double)) : NULL;
    /* det_pow: the GPU's form of the alpha series means, (sum n_z + alpha sum n_xy) / len from exact integer sums */
    r->det_series = conv_mode && m->noise == ORC_NOISE_ALPHA && m->det_pow;
    r->series_z = r->det_series ? (double *)calloc((size_t)steps, sizeof(double)) : NULL;
    r->series_xy = r->det_series ? (double *)calloc((size_t)steps, sizeof(double)) : NULL;
}

/* one iteration of the loop of decoders.py:55; returns 1 when the run has ended (criterion or `steps`) */
static int pteq_run_step(pteq_run *r, orc_rng *rng)
{
    orc_ladder *ld = r->ld;
    const orc_model *m = r->m;
    const uint64_t step = r->step;
    orc_ladder_step(ld, r->iters, rng);                             /* :57 */
    int cur = orc_eq_class(m->code, m->L, ld->states);              /* :60 */
    if (ld->tops0 >= (uint64_t)r->tops_burn) {                      /* :63 */
        r->since_burn = step - r->resulting_burn_in;
        r->eq[cur] += 1;                                            /* :66-67 (running row) */
        r->recorded = r->since_burn + 1;
        if (r->series) r->series[r->since_burn] = m->noise == ORC_NOISE_ALPHA ? ld->n_eff[0]     /* decoders_biasednoise.py:204 */
                                                                           : (double)orc_count_errors((size_t)ld->nq, ld->states);
        if (r->det_series) { r->series_z[r->since_burn] = ld->n_eff_cnt[0]; r->series_xy[r->since_burn] = ld->n_eff_cnt[1]; }
    } else {
        r->resulting_burn_in += 1;                                  /* :71 */
    }
    r->step = step + 1;
    if (r->conv_mode == 1 && ld->tops0 >= (uint64_t)r->TOPS) {      /* :74 */
        uint64_t l = r->since_burn + 1;
        double q2 = mean_range(r->series, l / 4, l / 2);
        double q4 = mean_range(r->series, 3 * l / 4, l);
        if (r->det_series) {
            const double n2 = (double)(l / 2 - l / 4), n4 = (double)(l - 3 * l / 4);
            q2 = n2 > 0 ? (sum_range(r->series_z, l / 4, l / 2) + m->alpha * sum_range(r->series_xy, l / 4, l / 2)) / n2 : NAN;
            q4 = n4 > 0 ? (sum_range(r->series_z, 3 * l / 4, l) + m->alpha * sum_range(r->series_xy, 3 * l / 4, l)) / n4 : NAN;
        }
        double err = fabs(q2 - q4);
        if (err < r->eps) {                                         /* :102 */
            if (r->conv_streak >= (uint64_t)r->SEQ) { r->converged = 1; return 1; }
            r->conv_streak = ld->tops0 - r->conv_start;             /* :79 */
        } else {
            r->conv_streak = 0;                                     /* :81-82 */
            r->conv_start = ld->tops0;
        }
    }
    return r->step >= r->steps;
}

static void pteq_run_finish(pteq_run *r, orc_pteq_result *res, uint8_t *final_states)
{
    memcpy(res->counts, r->eq, sizeof r->eq);
    res->samples = r->recorded;
    res->tops0 = r->ld->tops0;
    res->steps_done = r->step;
    res->converged = r->converged;
    memset(res->percent, 0, sizeof res->percent);
    for (int i = 0; i < r->ncls; ++i)                               /* :89 */
        res->percent[i] = (uint8_t)((double)r->eq[i] / (double)(r->since_burn + 1) * 100.0);
    if (final_states) memcpy(final_states, r->ld->states, (size_t)r->ld->Nc * r->ld->nq);
    free(r->series); free(r->series_z); free(r->series_xy);
    orc_ladder_free(r->ld);
    r->ld = NULL;
}

void orc_pteq(const orc_model *m, const uint8_t *init, double p, int Nc, int SEQ, int TOPS, int tops_burn,
              double eps, uint64_t steps, uint64_t iters, int conv_mode, orc_rng *rng,
              orc_pteq_result *res, uint8_t *final_states)
{
    pteq_run r;
    pteq_run_init(&r, m, init, p, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, conv_mode);
    while (r.step < steps)
        if (pteq_run_step(&r, rng)) break;
    pteq_run_finish(&r, res, final_states);
}

/* The criterion-stopped runs of scan = 3 on a persistent grid (the GPU's work queue, restated): workgroup g of `grid` owns the
 * ladders [g C, min(N, (g+1) C)), C = ceil(N / grid) rounded up to a multiple of 64; its 64 lanes start on the first 64 of them and run in lockstep, one ladder step
 * per workgroup step t.  A lane whose ladder ends at workgroup step t (criterion, or `steps` of its own steps) idles during steps
 * t + 1 and t + 2 (the GPU books a step behind the next one's barrier and restages behind the one after) and starts the workgroup's
 * next unassigned ladder at step t + 3 -- lanes that end together take them in lane order.  A
 * ladder's acceptance and swap uniforms are addressed by its own index and its own step; the generator picks belong to the lane's
 * position: group (first_syndrome >> 6) + g, workgroup step t.  Deterministic for a given (N, grid, first_syndrome); every ladder
 * has the reference's law. */
void orc_pteq_wave_queue(const orc_model *m, const uint8_t *init, uint64_t N, uint32_t first_syndrome, double p, int Nc, int SEQ,
                         int TOPS, int tops_burn, double eps, uint64_t steps, uint64_t iters, uint64_t seed, uint32_t grid, int n_threads,
                         uint32_t *counts_out, uint64_t *samples_out, uint64_t *tops0_out, uint64_t *steps_done_out, uint8_t *converged_out)
{
    const size_t nq = (size_t)orc_nq(m->code, m->L);
    const uint64_t C = ((N + grid - 1) / grid + 63) / 64 * 64;        /* whole groups of 64 per workgroup */
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t g = 0; g < (int64_t)grid; ++g) {
        const uint64_t lo = (uint64_t)g * C, hi = lo + C < N ? lo + C : N;
        if (lo >= N) continue;
        pteq_run run[64];
        orc_rng rng[64];
        int64_t lad[64];                      /* the lane's ladder, -1: none */
        uint64_t start[64];                   /* workgroup step a waiting lane starts its next ladder at */
        uint64_t next = lo + 64;              /* the workgroup's queue */
        int alive = 0;
        for (int l = 0; l < 64; ++l) {
            lad[l] = lo + (uint64_t)l < hi ? (int64_t)(lo + (uint64_t)l) : -1;
            start[l] = 0;
            if (lad[l] >= 0) ++alive;
        }
        int running[64];
        memset(running, 0, sizeof running);
        for (uint64_t t = 0; alive > 0; ++t) {
            for (int l = 0; l < 64; ++l) {
                if (lad[l] < 0) continue;
                if (!running[l]) {
                    if (t < start[l]) continue;
                    pteq_run_init(&run[l], m, init + (size_t)lad[l] * nq, p, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, 1);
                    orc_rng_init_philox(&rng[l], seed, first_syndrome + (uint32_t)lad[l]);
                    rng[l].wave_override = 1;
                    rng[l].wave_group = (first_syndrome >> 6) + (uint32_t)g;
                    rng[l].wave_t0 = t;
                    running[l] = 1;
                }
                if (pteq_run_step(&run[l], &rng[l])) {
                    orc_pteq_result res;
                    pteq_run_finish(&run[l], &res, NULL);
                    const size_t s = (size_t)lad[l];
                    memcpy(counts_out + s * 16, res.counts, sizeof res.counts);
                    samples_out[s] = res.samples; tops0_out[s] = res.tops0;
                    if (steps_done_out) steps_done_out[s] = res.steps_done;
                    if (converged_out) converged_out[s] = (uint8_t)res.converged;
                    running[l] = 0;
                    lad[l] = -2;                                      /* ended at this step: takes a new ladder below */
                }
            }
            for (int l = 0; l < 64; ++l) {                            /* lanes that ended at step t, in lane order */
                if (lad[l] != -2) continue;
                if (next < hi) { lad[l] = (int64_t)next++; start[l] = t + 3; }
                else { lad[l] = -1; --alive; }
            }
        }
    }
}

void orc_toric_pteq(int L, const uint8_t *init, double p, int Nc, int SEQ, int TOPS, int tops_burn,
                    double eps, uint64_t steps, uint64_t iters, int conv_mode, orc_rng *rng,
                    orc_pteq_result *res, uint8_t *final_states)
{
    orc_model m = {ORC_TORIC, L, ORC_NOISE_DEPOLARIZING, 0.0, 0.0, 0, 0, {0.0, 0.0, 0.0}};
    orc_pteq(&m, init, p, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, conv_mode, rng, res, final_states);
}

void orc_toric_pteq_batch(int L, const uint8_t *init, uint64_t N, uint32_t first_syndrome, double p,
                          int Nc, int tops_burn, uint64_t steps, uint64_t iters, uint64_t seed,
                          int n_threads, uint32_t *counts_out, uint64_t *samples_out,
                          uint64_t *tops0_out, uint8_t *final_states)
{
    orc_model m = {ORC_TORIC, L, ORC_NOISE_DEPOLARIZING, 0.0, 0.0, 0, 0, {0.0, 0.0, 0.0}};
    orc_pteq_batch(&m, init, N, first_syndrome, p, Nc, 2, 10, tops_burn, 0.1, steps, iters, 0, seed,
                   n_threads, counts_out, samples_out, tops0_out, NULL, NULL, final_states);
}

void orc_toric_pteq_batch_conv(int L, const uint8_t *init, uint64_t N, uint32_t first_syndrome, double p,
                               int Nc, int SEQ, int TOPS, int tops_burn, double eps, uint64_t steps,
                               uint64_t iters, int conv_mode, uint64_t seed, int n_threads,
                               uint32_t *counts_out, uint64_t *samples_out, uint64_t *tops0_out,
                               uint64_t *steps_done_out, uint8_t *converged_out, uint8_t *final_states)
{
    orc_model m = {ORC_TORIC, L, ORC_NOISE_DEPOLARIZING, 0.0, 0.0, 0, 0, {0.0, 0.0, 0.0}};
    orc_pteq_batch(&m, init, N, first_syndrome, p, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, conv_mode, seed,
                   n_threads, counts_out, samples_out, tops0_out, steps_done_out, converged_out, final_states);
}

/* counts_out is [N][16] for every code (4-class codes use the first four entries) */
void orc_pteq_batch(const orc_model *m, const uint8_t *init, uint64_t N, uint32_t first_syndrome, double p,
                    int Nc, int SEQ, int TOPS, int tops_burn, double eps, uint64_t steps,
                    uint64_t iters, int conv_mode, uint64_t seed, int n_threads,
                    uint32_t *counts_out, uint64_t *samples_out, uint64_t *tops0_out,
                    uint64_t *steps_done_out, uint8_t *converged_out, uint8_t *final_states)
{
    const size_t nq = (size_t)orc_nq(m->code, m->L);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t s = 0; s < (int64_t)N; ++s) {
        orc_rng rng;
        orc_rng_init_philox(&rng, seed, first_syndrome + (uint32_t)s);
        orc_pteq_result res;
        orc_pteq(m, init + (size_t)s * nq, p, Nc, SEQ, TOPS, tops_burn, eps, steps, iters, conv_mode, &rng,
                 &res, final_states ? final_states + (size_t)s * Nc * nq : NULL);
        memcpy(counts_out + (size_t)s * 16, res.counts, sizeof res.counts);
        samples_out[s] = res.samples;
        tops0_out[s] = res.tops0;
        if (steps_done_out) steps_done_out[s] = res.steps_done;
        if (converged_out) converged_out[s] = (uint8_t)res.converged;
    }
}


/* ------------------------------------------------------------------------ */
/*  unique-chain estimators (decoders.py:138-233)                            */
/* ------------------------------------------------------------------------ */
uint64_t orc_state_key(const uint8_t *state, size_t nq)
{
    uint64_t h = 0xCBF29CE484222325ull;                 /* FNV-1a, 64 bit */
    for (size_t i = 0; i < nq; ++i) { h ^= state[i]; h *= 0x100000001B3ull; }
    h ^= h >> 32;                                      /* the low bits index the table: fold the high half in */
    return h ? h : 1;
}

int orc_uset_insert(uint64_t *tab, uint64_t cap, uint64_t key)
{
    uint64_t i = (key * 0x9E3779B97F4A7C15ull) >> 20 & (cap - 1);
    for (;;) {
        if (tab[i] == 0) { tab[i] = key; return 1; }
        if (tab[i] == key) return 0;
        i = (i + 1) & (cap - 1);
    }
}

void orc_ptdc_droplet(const orc_model *m, const uint8_t *init, double p_sampling, int Nc, uint64_t steps, uint64_t iters,
                      orc_rng *rng, uint64_t *tab, uint64_t cap, uint32_t *hist, int per_rung, uint32_t *mhist,
                      double conv_mult, uint64_t *steps_done, orc_xyz_sink *xyz)
{
    orc_ladder *ld = orc_ladder_new(m, init, p_sampling, Nc, 0.0);      /* decoders.py:182,196: no p_logical */
    const size_t nq = (size_t)ld->nq;
    /* the early stop looks at the droplet's OWN dictionary (each droplet is a separate process in the reference), while `tab`
     * may be shared by the droplets of a class: keep a private set for it */
    uint64_t *own = (conv_mult != 0 && !per_rung) ? (uint64_t *)calloc(cap, sizeof(uint64_t)) : NULL;
    uint64_t shortest = 2 * (uint64_t)m->L * (uint64_t)m->L;           /* :140, :242 */
    double stop = (double)steps;                                        /* :241 (PTDC_droplet sets it at step 0, :156) */
    uint64_t step;
    for (step = 0; step < steps; ++step) {                              /* :142 */
        orc_ladder_step(ld, iters, rng);                                /* :144 */
        for (int c = 0; c < Nc; ++c) {                                  /* :146-152 */
            const uint8_t *st = ld->states + (size_t)c * nq;
            const size_t set = per_rung ? (size_t)c : 0;
            const uint64_t n = orc_count_errors(nq, st), key = orc_state_key(st, nq);
            if (orc_uset_insert(tab + set * cap, cap, key)) {
                hist[set * (nq + 1) + n]++;
                if (xyz) {
                    uint32_t c[4] = {0, 0, 0, 0};
                    for (size_t q = 0; q < nq; ++q) c[st[q]]++;
                    xyz->out[xyz->n++] = c[1] | c[2] << 10 | c[3] << 20;
                }
            }
            if (mhist) mhist[set * (nq + 1) + n]++;
            if (own && orc_uset_insert(own, cap, key) && n <= shortest) { shortest = n; stop = (double)step * conv_mult; }   /* :153-156 */
        }
        if (own && (double)step >= stop && step * 100 >= steps) { ++step; break; }   /* :159-162 */
    }
    if (steps_done) *steps_done = step;
    free(own);
    orc_ladder_free(ld);
}

void orc_ptdc_batch(const orc_model *m, const uint8_t *init, uint64_t N, int ncls, int D, int init_per_droplet,
                    uint32_t first_syndrome, double p_sampling, int Nc, uint64_t steps, uint64_t iters, uint64_t seed,
                    int n_threads, uint32_t *hist_out, int per_rung, uint32_t *mhist_out, double conv_mult, uint32_t *xyz_out)
{
    const size_t nq = (size_t)orc_nq(m->code, m->L);
    uint64_t cap = 16;
    while (cap < 2 * steps * (per_rung ? 1u : (uint64_t)Nc * (uint64_t)D)) cap <<= 1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t sc = 0; sc < (int64_t)(N * (uint64_t)ncls); ++sc) {
        uint64_t *tab = (uint64_t *)calloc(cap * (per_rung ? (size_t)Nc : 1), sizeof(uint64_t));
        const uint64_t maxu = steps * (uint64_t)Nc * (uint64_t)D;
        orc_xyz_sink sink = {xyz_out ? xyz_out + (size_t)sc * maxu : NULL, 0};
        if (xyz_out) memset(sink.out, 0xFF, maxu * sizeof(uint32_t));
        for (int d = 0; d < D; ++d) {
            const size_t out = per_rung ? ((size_t)sc * D + d) * Nc * (nq + 1) : (size_t)sc * (nq + 1);
            if (per_rung || d == 0) {
                memset(hist_out + out, 0, (per_rung ? Nc : 1) * (nq + 1) * sizeof(uint32_t));
                if (mhist_out) memset(mhist_out + out, 0, (per_rung ? Nc : 1) * (nq + 1) * sizeof(uint32_t));
            }
            if (per_rung) memset(tab, 0, cap * (size_t)Nc * sizeof(uint64_t));
            orc_rng rng;
            orc_rng_init_philox(&rng, seed, first_syndrome + (uint32_t)(sc * D + d));
            orc_ptdc_droplet(m, init + (size_t)(init_per_droplet ? sc * D + d : sc) * nq, p_sampling, Nc, steps, iters, &rng, tab, cap,
                             hist_out + out, per_rung, mhist_out ? mhist_out + out : NULL, conv_mult, NULL,
                             xyz_out && !per_rung ? &sink : NULL);
        }
        free(tab);
    }
}

/* ------------------------------------------------------------------------ */
/* Syndrome generation (generate_data.py:57-60,110-131): N error chains from    */
/* generate_random_error -- toric_model.py:15-23 (error w.p. p, Pauli uniform),  */
/* xzzx_model.py:16-30 / rotated_surface_model.py:25-38 / planar_model.py:18-40  */
/* (one uniform against p_z, p_z + p_x, p_z + p_x + p_y) -- then, if `hide`, one  */
/* apply_random_logical (toric_model.py:228-253, xzzx_model.py:340-357).  Philox  */
/* mode only: qubit q draws words (2 (q & 1), 2 (q & 1) + 1) of block (q >> 1, 0)  */
/* of stream 0x200, the logical operator's fields come from block (0, 1).          */
/* ------------------------------------------------------------------------ */
static uint64_t gen_thr64(double v)          /* u < v  <=>  x < ceil(v 2^32) for u = x 2^-32 */
{
    if (!(v < 1.0)) return 1ull << 32;
    if (!(v > 0.0)) return 0;
    return (uint64_t)ceil(v * 4294967296.0);
}

void orc_generate_syndromes(int code, int L, uint64_t N, double p_x, double p_y, double p_z, int hide, uint64_t seed,
                            uint32_t first_syndrome, uint8_t *init_out, uint8_t *raw_out, int32_t *eq_true_out)
{
    const int nq = orc_nq(code, L);
    const uint64_t tz = code == ORC_TORIC ? gen_thr64((p_x + p_y) + p_z) : gen_thr64(p_z);
    const uint64_t tzx = gen_thr64(p_z + p_x), tzxy = gen_thr64((p_z + p_x) + p_y);
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint8_t *tmp = (uint8_t *)malloc((size_t)nq);
    for (uint64_t s = 0; s < N; ++s) {
        uint8_t *m = init_out + s * (size_t)nq;
        const uint32_t syn = first_syndrome + (uint32_t)s;
        for (int q = 0; q < nq; ++q) {
            uint32_t ctr[4] = {(uint32_t)(q >> 1), 0u, syn, 0x200u}, w[4];
            orc_philox4x32_10(ctr, key, w);
            const uint32_t u1 = w[2 * (q & 1)], u2 = w[2 * (q & 1) + 1];
            uint8_t v = 0;
            if (code == ORC_TORIC) {
                if ((uint64_t)u1 < tz) v = (uint8_t)(1u + (uint32_t)(((uint64_t)u2 * 3u) >> 32));      /* randint(3) + 1 */
            } else {
                v = (uint64_t)u1 < tz ? 3 : (uint64_t)u1 < tzx ? 1 : (uint64_t)u1 < tzxy ? 2 : 0;
                if (code == ORC_PLANAR && q >= L * L) {                 /* planar_model.py:38-39 */
                    const int rc = q - L * L, row = rc / L, col = rc % L;
                    if (row == L - 1 || col == L - 1) v = 0;
                }
            }
            m[q] = v;
        }
        if (raw_out) memcpy(raw_out + s * (size_t)nq, m, (size_t)nq);
        if (eq_true_out) eq_true_out[s] = orc_eq_class(code, L, m);
        if (!hide) continue;
        uint32_t ctr[4] = {0u, 1u << 16, syn, 0x200u}, x[4];
        orc_philox4x32_10(ctr, key, x);
        if (code == ORC_TORIC) {
            const int op0 = (int)(x[1] >> 30), op1 = (int)(x[2] >> 30);
            const int x0 = (op0 == 1 || op0 == 2) ? (int)(((uint64_t)(uint32_t)(x[1] << 2) * (uint32_t)L) >> 32) : 0;
            const int z0 = (op0 == 3 || op0 == 2) ? (int)(((x[3] >> 16) * (uint32_t)L) >> 16) : 0;
            const int x1 = (op1 == 1 || op1 == 2) ? (int)(((uint64_t)(uint32_t)(x[2] << 2) * (uint32_t)L) >> 32) : 0;
            const int z1 = (op1 == 3 || op1 == 2) ? (int)(((x[3] & 0xFFFFu) * (uint32_t)L) >> 16) : 0;
            orc_toric_apply_logical(L, m, tmp, op0, 0, x0, z0);
            orc_toric_apply_logical(L, tmp, m, op1, 1, x1, z1);
        } else {
            const int op = (int)(x[1] >> 30);
            const int xp = (op == 1 || op == 2) ? (int)(((uint64_t)(uint32_t)(x[1] << 2) * (uint32_t)L) >> 32) : 0;
            const int zp = (op == 3 || op == 2) ? (int)(((x[3] >> 16) * (uint32_t)L) >> 16) : 0;
            orc_surf_apply_logical(code, L, m, tmp, op, xp, zp);
            memcpy(m, tmp, (size_t)nq);
        }
    }
    free(tmp);
}
