// C-ABI of libqecmc (include/qecmc.h): argument checking, host-side threshold
// tables, device buffers and kernel launches.  No CPU compute fallback: every
// entry point that computes needs a HIP device.
#include "../../include/qecmc.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "stencil_bytes.hpp"
#include "tables.hpp"

using namespace qecmc;
using namespace qecmc::tables;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(QECMC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int use_device(int dev)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(QECMC_ERR_NO_DEVICE, "no HIP device visible: libqecmc has no CPU fallback");
    if (dev < 0 || dev >= n) return fail(QECMC_ERR_NO_DEVICE, "device %d out of range (0..%d)", dev, n - 1);
    HIP_TRY(hipSetDevice(dev));
    return 0;
}

// Small device blocks are recycled: the drop-in entry points (one chain or one ladder per call, mcmc.py:81-103 driven from a Python loop)
// would otherwise spend most of a call in hipMalloc / hipFree -- hipFree also waits for the device.  Blocks up to kPoolBlock bytes are
// rounded up to a power of two and kept, per device, until kPoolBytes are cached; larger ones (the batched calls) go to the runtime as before.
// Nothing here is zero-filled, exactly like hipMalloc: every user writes its buffer before a kernel reads it.
class DevPool {
public:
    static constexpr size_t kPoolBlock = size_t(1) << 20, kPoolBytes = size_t(32) << 20;
    static DevPool &get() { static DevPool *pool = new DevPool; return *pool; }   // (never destroyed: the HIP runtime may be gone before static destructors run)
    hipError_t take(size_t bytes, void **p, size_t *cap, int *dev)
    {
        *cap = 0;
        if (bytes > kPoolBlock) return hipMalloc(p, bytes);
        size_t want = 256;
        while (want < bytes) want <<= 1;
        if (hipError_t e = hipGetDevice(dev)) return e;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].cap == want && free_[i].dev == *dev) {
                    *p = free_[i].p; *cap = want;
                    cached_ -= want;
                    free_[i] = free_.back(); free_.pop_back();
                    return hipSuccess;
                }
        }
        *cap = want;
        return hipMalloc(p, want);
    }
    void give(void *p, size_t cap, int dev)
    {
        if (cap) {
            std::lock_guard<std::mutex> g(mu_);
            if (cached_ + cap <= kPoolBytes) { free_.push_back({p, cap, dev}); cached_ += cap; return; }
        }
        (void)hipFree(p);
    }
private:
    struct Blk { void *p; size_t cap; int dev; };
    std::mutex mu_;
    std::vector<Blk> free_;
    size_t cached_ = 0;
};

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;          // pooled block size (0: straight from hipMalloc)
    int dev = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) DevPool::get().give(p, cap, dev); }
    hipError_t alloc(size_t bytes) { return DevPool::get().take(bytes ? bytes : 1, &p, &cap, &dev); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

int check_code_L(int code, int L)
{
    if (code != QECMC_TORIC && code != QECMC_XZZX && code != QECMC_ROTATED && code != QECMC_PLANAR)
        return fail(QECMC_ERR_INVALID, "code %d unknown (0 toric, 1 xzzx, 2 rotated, 3 planar)", code);
    if (L < 2 || L > 64) return fail(QECMC_ERR_INVALID, "L=%d out of range [2,64]", L);
    if ((code == QECMC_XZZX || code == QECMC_ROTATED) && (L < 3 || L % 2 == 0))
        return fail(QECMC_ERR_INVALID, "L=%d: the xzzx / rotated models need odd L >= 3 (their half-plaquette indexing, xzzx_model.py:444)", L);
    return 0;
}

inline size_t code_nq(int code, int L) { return (size_t)code_nq_of(code, L); }

}  // namespace

struct qecmc_plan {
    qecmc_params prm;
    LadderArgs args;
    DevBuf swap_thr, lmask, acc_top, gen, bias, lnb, xyz_lut, gen_type, queue, phases, wu_desc, col_thr;
    uint32_t queue_grid = 0;                                   // persistent grid of the work-queue kernels (0: not a queue plan)
    size_t lds_bytes;
    uint32_t *d_swap_acc = nullptr, *d_nerr_sum = nullptr;   // qecmc_plan_set_stats (caller-owned)
};

namespace {

int validate_params(const qecmc_params *p)
{
    if (!p) return fail(QECMC_ERR_INVALID, "params is NULL");
    if (p->abi_size != sizeof(qecmc_params))
        return fail(QECMC_ERR_INVALID, "params->abi_size=%u, this library expects %zu", p->abi_size, sizeof(qecmc_params));
    if (int rc = check_code_L(p->code, p->L)) return rc;
    if (p->Nc < 1 || p->Nc > kMaxNc) return fail(QECMC_ERR_INVALID, "Nc=%d out of range [1,%d]", p->Nc, kMaxNc);
    if (p->noise != QECMC_NOISE_DEPOLARIZING && p->noise != QECMC_NOISE_BIASED && p->noise != QECMC_NOISE_ALPHA) return fail(QECMC_ERR_INVALID, "noise model %d unknown", p->noise);
    if (p->noise == QECMC_NOISE_ALPHA) {
        if (!(p->alpha > 0.0)) return fail(QECMC_ERR_INVALID, "alpha=%g must be positive", p->alpha);
        if (!(p->p > 0.0) || !(p->p <= 1.0)) return fail(QECMC_ERR_INVALID, "pz_tilde=%g must be in (0, 1]", p->p);
        if (p->code == QECMC_TORIC || p->code == QECMC_PLANAR) return fail(QECMC_ERR_UNSUPPORTED, "alpha noise is built for the xzzx and rotated codes (the reference sizes its weights for L^2 qubits, mcmc_alpha.py:27)");
    } else
    if (p->noise == QECMC_NOISE_BIASED) {
        if (!(p->eta > 0.0)) return fail(QECMC_ERR_INVALID, "eta=%g must be positive", p->eta);
        if (!(p->p > 0.0) || !(p->p < (p->eta + 1) / (2 * p->eta + 1))) return fail(QECMC_ERR_INVALID, "p=%g must be in (0, (eta+1)/(2 eta+1))", p->p);
        if (p->code == QECMC_TORIC || p->code == QECMC_PLANAR) return fail(QECMC_ERR_UNSUPPORTED, "biased noise is built for the xzzx and rotated codes (BASELINE config 4)");
    } else if (!(p->p > 0.0) || !(p->p <= 0.75)) return fail(QECMC_ERR_INVALID, "p=%g must be in (0, 0.75]", p->p);
    if (!(p->p_logical >= 0.0) || !(p->p_logical <= 1.0)) return fail(QECMC_ERR_INVALID, "p_logical=%g must be in [0,1]", p->p_logical);
    if (p->scan != QECMC_SCAN_RANDOM && p->scan != QECMC_SCAN_SWEEP && p->scan != QECMC_SCAN_COLOUR && p->scan != QECMC_SCAN_WAVE) return fail(QECMC_ERR_INVALID, "scan mode %d unknown", p->scan);
    if (p->scan != QECMC_SCAN_RANDOM && p->noise != QECMC_NOISE_DEPOLARIZING && !(p->scan == QECMC_SCAN_WAVE && p->noise == QECMC_NOISE_ALPHA) && p->scan != QECMC_SCAN_COLOUR)
        return fail(QECMC_ERR_UNSUPPORTED, "the sweep scan is built for the depolarizing rule only, the wave scan for the depolarizing and alpha rules");
    if (p->scan == QECMC_SCAN_WAVE) {
        if (p->Nc < 2) return fail(QECMC_ERR_UNSUPPORTED, "scan = wave needs a ladder whose top rung sits at p = 0.75 (Nc >= 2)");
        if (p->first_syndrome & 63u) return fail(QECMC_ERR_INVALID, "scan = wave shares a generator pick among the 64 ladders of a wavefront: first_syndrome=%u must be a multiple of 64", p->first_syndrome);
    }
    if (p->scan == QECMC_SCAN_COLOUR) {
        if (p->p_logical > 0.0 && p->Nc < 2 && p->noise != QECMC_NOISE_BIASED) return fail(QECMC_ERR_UNSUPPORTED, "scan = colour needs the top rung at p = 0.75 (Nc >= 2) when logical moves are on");
    }
    if (p->conv_mode != QECMC_CONV_NONE && p->conv_mode != QECMC_CONV_ERROR_BASED) return fail(QECMC_ERR_INVALID, "conv_mode %d unknown", p->conv_mode);
    if (p->conv_mode == QECMC_CONV_ERROR_BASED && (p->TOPS < 0 || p->SEQ < 0 || !(p->eps >= 0))) return fail(QECMC_ERR_INVALID, "TOPS, SEQ and eps must be non-negative");
    if (p->iters == 0 || p->iters > 0xFFFFFFFFull) return fail(QECMC_ERR_INVALID, "iters out of range");
    if (p->tops_burn < 0) return fail(QECMC_ERR_INVALID, "tops_burn must be >= 0");
    if (p->replicas < 0 || p->replicas > 65536) return fail(QECMC_ERR_INVALID, "replicas=%d out of range [0, 65536]", p->replicas);
    // the R ladders of a syndrome add their class counts / samples / tops0 into uint32 outputs: at most `steps` each
    if (p->replicas > 1 && (uint64_t)p->replicas * p->steps > 0xFFFFFFFFull)
        return fail(QECMC_ERR_INVALID, "replicas * steps = %llu overflows the summed 32-bit class counts: lower one of them",
                    (unsigned long long)((uint64_t)p->replicas * p->steps));
    return 0;
}

int build_plan(const qecmc_params *p, qecmc_plan *pl)
{
    pl->prm = *p;
    LadderArgs &a = pl->args;
    std::memset(&a, 0, sizeof a);
    const int L = p->L, Nc = p->Nc, nq = (int)code_nq(p->code, L), W = (nq + 15) / 16, ncls = p->code == QECMC_TORIC ? 16 : 4;
    const bool alpha = p->noise == QECMC_NOISE_ALPHA;
    const bool biased = p->noise == QECMC_NOISE_BIASED || alpha;     // table-driven acceptance pn / pb
    a.code = p->code; a.noise = p->noise; a.alpha = p->alpha;
    a.replicas = p->replicas > 1 ? (uint32_t)p->replicas : 1u;
    a.tune = p->flags & 0xFFFFu;                                // developer switches (qecmc_flag): which variant runs, never what it computes
    a.L = L; a.Nc = Nc; a.W = W; a.nq = nq; a.ncls = ncls;
    a.iters = (uint32_t)p->iters;
    a.seed_lo = (uint32_t)p->seed; a.seed_hi = (uint32_t)(p->seed >> 32);
    a.tops_burn = (uint32_t)p->tops_burn;
    a.conv_mode = p->conv_mode; a.TOPS = (uint32_t)p->TOPS; a.SEQ = (uint32_t)p->SEQ; a.eps = p->eps;
    a.thr_logical = p->p_logical > 0 ? thr64(p->p_logical) : 0;
    const uint32_t n_gen = p->code == QECMC_TORIC ? 2u * L * L : (uint32_t)surf_ngen(p->code, L);
    if (n_gen > kMaxGenLds)   // every kernel path stages the generator table in LDS
        return fail(QECMC_ERR_UNSUPPORTED, "L=%d: %u generators exceed the LDS table of %u (needed by scan=1 and by the xzzx / rotated codes)", L, n_gen, kMaxGenLds);
    const std::vector<uint32_t> gt = p->code == QECMC_TORIC ? toric_generator_table(L) : surf_generator_table(p->code, L);
    std::vector<uint8_t> gen_type(gt.size() / 2, 0);
    std::vector<uint32_t> xyz_lut;
    const bool typed = biased || (p->code != QECMC_TORIC && !p->scan);   // the generators' Pauli patterns: the biased rules' count-change table,
    std::vector<uint32_t> patterns;                                      // the plaquette codes' dE table (ladder_kernel.hpp, DELUT)
    if (typed) {
        generator_patterns(gt, gen_type, patterns);
        if (patterns.size() > 16) return fail(QECMC_ERR_UNSUPPORTED, "%zu distinct generator Pauli patterns (> 16)", patterns.size());
        a.n_types = (int)patterns.size();
        for (size_t t = 0; t < patterns.size(); ++t) a.type_ops[t] = (uint8_t)patterns[t];
    }
    if (biased) {
        // the biased / alpha rules' table of count changes (tables.hpp)
        if (nq > 511) return fail(QECMC_ERR_UNSUPPORTED, "biased / alpha noise packs the error counts in 10-bit fields: nq=%d", nq);
        xyz_lut = count_change_table(patterns);
    }
    pl->lds_bytes = p->scan == QECMC_SCAN_COLOUR ? sizeof(uint32_t) * ((size_t)Nc * W + 4 * (size_t)Nc + (size_t)ncls)   // (ladder_colour.hip: one ladder per workgroup)
                  : p->scan == QECMC_SCAN_WAVE ? wu_lds_bytes(Nc, W, ncls, L, p->conv_mode != 0, alpha)
                                                 : ladder_lds_bytes(L, Nc, W, ncls, ladder_gen_dwords(p->code, p->noise, p->scan, n_gen, Nc, nq, a.n_types));
    if (pl->lds_bytes > 160 * 1024)
        return fail(QECMC_ERR_UNSUPPORTED, "L=%d Nc=%d needs %zu B of LDS per workgroup (> 160 KiB)", L, Nc, pl->lds_bytes);
    const bool wave_queue = p->scan == QECMC_SCAN_WAVE && p->conv_mode != 0;      // (ladder_wu.hpp: the workgroups' own work queues)
    if (ladder_uses_queue(p->code, p->noise, p->scan, p->conv_mode, L, Nc, p->p_logical) || wave_queue) {
        // runs that stop by the convergence criterion: a persistent grid (what one launch keeps resident) fed from a counter
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, p->device));
        // (waves per CU: 8 per SIMD for the 512-thread depolarizing kernels, 4 for the 1024-thread ones and for the biased / alpha
        // queue kernels, which run at 128 VGPRs: ladder_biased.hip)
        const size_t per_cu_lds = (160 * 1024) / pl->lds_bytes,
                     per_cu_waves = (size_t)(wave_queue ? (alpha ? 4 * kWuAlphaQueueWaves : W > 12 ? 24 : 32)      // (ladder_wu.hpp wu_pick_it: 8 waves per SIMD, 6 at 16 words)
                                                        : (Nc * 64 <= 512 && !p->noise) ? 32 : 16) / (size_t)Nc;
        size_t per_cu = per_cu_lds < per_cu_waves ? per_cu_lds : per_cu_waves;
        if (per_cu < 1) per_cu = 1;
        pl->queue_grid = (uint32_t)(per_cu * (size_t)prop.multiProcessorCount);
        if (p->flags >> 16) pl->queue_grid = p->flags >> 16;   // tests: force refills on small batches
        if (pl->queue_grid == 0) pl->queue_grid = 1;
        if (!wave_queue) HIP_TRY(pl->queue.alloc(sizeof(uint32_t)));
    }

    std::vector<double> pladder, pdiff;
    // p_top = 0.75 (mcmc.py:62) or (eta+1)/(2 eta+1) (mcmc_biased.py:81)
    // ... or pz_tilde_top = 1 (mcmc_alpha.py:94)
    ladder_probabilities(p->p, alpha ? 1.0 : biased ? (p->eta + 1) / (2 * p->eta + 1) : 0.75, Nc, pladder, pdiff);   // mcmc.py:62-69
    if (alpha) pdiff.assign(pdiff.size(), 0.0);      // the depolarizing tables below are unused by the table-driven rules
    for (int c = 0; c < Nc && !biased; ++c) {
        const double f = chain_factor(pladder[c]);
        if (f >= 1.0 && !biased) a.acc_all_mask |= 1u << c;
        for (int d = 1; d <= 4; ++d) {
            a.acc_thr[c][d - 1] = thr32(std::pow(f, (double)d));                     // mcmc.py:42
            a.acc_thr44[c][d - 1] = thr44(std::pow(f, (double)d));
        }
    }
    std::vector<uint32_t> top_tbl(nq + 1, 0u);                             // mcmc.py:34 for a top chain below p = 0.75
    for (int d = 1; d <= nq && !biased; ++d) top_tbl[d] = thr32(std::pow(chain_factor(pladder[Nc - 1]), (double)d));
    const std::vector<uint64_t> sw = swap_thresholds(pdiff, nq);             // mcmc.py:149
    for (int i = 0; i + 1 < Nc; ++i) {
        const double l2 = std::log2(pdiff[i]);
        a.swap_inv_log2[i] = (std::isfinite(l2) && l2 < 0) ? (float)(1.0 / l2) : 0.0f;
    }
    a.swap_fast_ok = 1;
    for (int i = 0; i + 1 < Nc; ++i)
        if (nq >= 1 && sw[(size_t)i * (nq + 1) + 1] > 0xFFFFFFFFull) a.swap_fast_ok = 0;
    const std::vector<uint32_t> lm = p->code == QECMC_TORIC ? toric_logical_masks(L, W) : surf_logical_masks(p->code, L, W);
    a.scan = p->scan;
    if (p->scan == QECMC_SCAN_COLOUR) {
        // the colour phases: groups of mutually disjoint generators, one wavefront pass each (tables.hpp)
        int n_phases = 0;
        const std::vector<uint16_t> ph = colour_phases(gt, n_phases);
        HIP_TRY(pl->phases.alloc(ph.size() * sizeof(uint16_t)));
        HIP_TRY(hipMemcpy(pl->phases.p, ph.data(), ph.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        a.phase_tab = pl->phases.as<uint16_t>();
        a.n_phases = (uint32_t)n_phases;
        // Ladder_alpha's top rung sits at pz_tilde = 1 (mcmc_alpha.py:94): every weight ratio is 1, it takes the coin like the depolarizing top rung
        if (alpha && Nc >= 2) a.acc_all_mask |= 1u << (Nc - 1);
        pl->lds_bytes = sizeof(uint32_t) * colour_lds_dwords(Nc, W, ncls, a.n_phases, (uint32_t)(gt.size() / 2), L, nq, a.swap_fast_ok != 0, p->noise);
        if (pl->lds_bytes > 160 * 1024)
            return fail(QECMC_ERR_UNSUPPORTED, "scan = colour: L=%d Nc=%d needs %zu B of LDS per workgroup (> 160 KiB)", L, Nc, pl->lds_bytes);
        if (p->p_logical > 0.0 && p->noise != QECMC_NOISE_BIASED && !((a.acc_all_mask >> (Nc - 1)) & 1u))
            return fail(QECMC_ERR_UNSUPPORTED, "scan = colour needs a top rung that accepts every move (p_top = 0.75) when logical moves are on");
    }
    if (p->scan == QECMC_SCAN_WAVE) {
        // the wave-uniform random scan (ladder_wu.hpp): one scalar-loadable descriptor per generator; states in registers
        const std::vector<uint32_t> wd = wave_descriptors(gt);
        a.n_gen = (uint32_t)(gt.size() / 2);
        pl->lds_bytes = wu_lds_bytes(Nc, W, ncls, L, p->conv_mode != 0, alpha);
        a.conv_mode = p->conv_mode;
        if (wd.empty()) return fail(QECMC_ERR_UNSUPPORTED, "scan = wave: a generator with three different Paulis");
        HIP_TRY(pl->wu_desc.alloc(wd.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(pl->wu_desc.p, wd.data(), wd.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        a.wu_desc = pl->wu_desc.as<uint32_t>();
    }
    {
        a.n_gen = (uint32_t)(gt.size() / 2);
        HIP_TRY(pl->gen.alloc(gt.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(pl->gen.p, gt.data(), gt.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        a.gen = pl->gen.as<uint2>();
    }
    if (biased) {
        std::vector<double> bt;
        for (int c = 0; c < Nc; ++c) {
            const std::vector<double> t = alpha ? alpha_tables(pladder[c], p->alpha, (size_t)nq) : bias_tables(pladder[c], p->eta, (size_t)nq);
            bt.insert(bt.end(), t.begin(), t.end());
        }
        HIP_TRY(pl->bias.alloc(bt.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(pl->bias.p, bt.data(), bt.size() * sizeof(double), hipMemcpyHostToDevice));
        a.bias_tbl = pl->bias.as<double>();
        for (int c = 0; c < Nc; ++c) {
            // log2(px / pI), log2(pz / pI) of rung c (px = py in both models: bias_tables / alpha_tables): the fast test's slopes
            const double *t = &bt[(size_t)c * 4 * (nq + 1)];
            a.bias_l2[c][0] = std::log2(t[1] / t[3 * (nq + 1) + 1]);
            a.bias_l2[c][1] = std::log2(t[2 * (nq + 1) + 1] / t[3 * (nq + 1) + 1]);
            a.bias_l2f[c][0] = (float)a.bias_l2[c][0];
            a.bias_l2f[c][1] = (float)a.bias_l2[c][1];
            // the fast test may run in single precision / fp16 count changes on this rung (ladder_kernel.hpp: a count changes by at
            // most 4 per proposal since loop entry; fp16 holds integers up to 2048; |l d| <= 2000 keeps the exponent's error below a unit)
            if (a.iters <= 512u && 4.0 * (double)a.iters * std::max(std::fabs(a.bias_l2[c][0]), std::fabs(a.bias_l2[c][1])) <= 2000.0)
                a.bias_f32ok |= 1u << c;
        }
        if (p->scan == QECMC_SCAN_COLOUR) {
            // scan = 2 under these rules (ladder_colour.hip): a generator is a Metropolis move for the model's own weight, accepted iff
            // u < (px / pI)^dxy (pz / pI)^dz -- as integers, u <= ceil(ratio 2^32) - 1 -- for its changes (dz, dxy) of n_z and n_x + n_y
            std::vector<uint32_t> ct((size_t)Nc * 81);
            for (int c = 0; c < Nc; ++c) {
                const double *t = &bt[(size_t)c * 4 * (nq + 1)];
                const double fxy = t[1] / t[3 * (nq + 1) + 1], fz = t[2 * (nq + 1) + 1] / t[3 * (nq + 1) + 1];
                for (int dz = -4; dz <= 4; ++dz)
                    for (int dxy = -4; dxy <= 4; ++dxy) {
                        const uint64_t th = thr64(std::pow(fxy, (double)dxy) * std::pow(fz, (double)dz));
                        if (th == 0) return fail(QECMC_ERR_UNSUPPORTED, "scan = colour: an acceptance ratio of rung %d underflows", c);
                        ct[(size_t)c * 81 + 9 * (dz + 4) + (dxy + 4)] = (uint32_t)(th - 1);
                    }
            }
            HIP_TRY(pl->col_thr.alloc(ct.size() * sizeof(uint32_t)));
            HIP_TRY(hipMemcpy(pl->col_thr.p, ct.data(), ct.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            a.col_thr = pl->col_thr.as<uint32_t>();
        }
        HIP_TRY(pl->xyz_lut.alloc(xyz_lut.size() * sizeof(uint32_t)));
        HIP_TRY(hipMemcpy(pl->xyz_lut.p, xyz_lut.data(), xyz_lut.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        a.xyz_lut = pl->xyz_lut.as<uint32_t>();
    }
    if (typed) {
        HIP_TRY(pl->gen_type.alloc(gen_type.size()));
        HIP_TRY(hipMemcpy(pl->gen_type.p, gen_type.data(), gen_type.size(), hipMemcpyHostToDevice));
        a.gen_type = pl->gen_type.as<uint8_t>();
    }
    if (alpha) {
        std::vector<double> lnb(Nc > 1 ? Nc - 1 : 1, 0.0);
        for (int i = 0; i + 1 < Nc; ++i) lnb[i] = std::log(pladder[i] / pladder[i + 1]);   // mcmc_alpha.py:123
        HIP_TRY(pl->lnb.alloc(lnb.size() * sizeof(double)));
        HIP_TRY(hipMemcpy(pl->lnb.p, lnb.data(), lnb.size() * sizeof(double), hipMemcpyHostToDevice));
        a.alpha_lnb = pl->lnb.as<double>();
    }
    HIP_TRY(pl->swap_thr.alloc(sw.size() * sizeof(uint64_t)));
    HIP_TRY(pl->lmask.alloc(lm.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(pl->swap_thr.p, sw.data(), sw.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(pl->lmask.p, lm.data(), lm.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(pl->acc_top.alloc(top_tbl.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(pl->acc_top.p, top_tbl.data(), top_tbl.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    a.acc_tbl_top = pl->acc_top.as<uint32_t>();
    a.swap_thr = pl->swap_thr.as<uint64_t>();
    a.lmask = pl->lmask.as<uint32_t>();
    if (p->scan == QECMC_SCAN_WAVE && (!wu_supported(a) || pl->lds_bytes > 160 * 1024))
        return fail(QECMC_ERR_UNSUPPORTED, "scan = wave: L=%d Nc=%d p=%g is outside what it is built for (depolarizing rule: a top rung that accepts every move, at most "
                    "16 packed state words per rung -- toric / planar L <= 11, xzzx / rotated L <= 16 --, fixed-length runs of up to 8 rungs 32 words -- toric L <= 16, xzzx / rotated L <= 22; alpha rule: xzzx / rotated L <= 11, "
                    "4 iters max|log2 ratio| <= 2000 --, %zu B of LDS)", L, Nc, p->p, pl->lds_bytes);
    return 0;
}

}  // namespace

extern "C" {

int qecmc_abi_version(void) { return QECMC_ABI_VERSION; }
const char *qecmc_last_error(void) { return g_err.c_str(); }
int qecmc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------- primitives
#define PRIM_PROLOGUE()                                   \
    if (int rc = check_code_L(code, L)) return rc;        \
    if (int rc = use_device(0)) return rc;                \
    const size_t nq = code_nq(code, L);                   \
    (void)nq;                                             \
    if (N == 0) return 0

int qecmc_apply_stabilizer(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *rows,
                           const int32_t *cols, const int32_t *ops, int32_t *dE)
{
    PRIM_PROLOGUE();
    if (!in || !out || !rows || !cols || !ops || !dE) return fail(QECMC_ERR_INVALID, "NULL buffer");
    for (uint64_t i = 0; i < N; ++i) {
        if (ops[i] != 1 && ops[i] != 3) return fail(QECMC_ERR_INVALID, "stabilizer %llu: operator %d is not 1 or 3", (unsigned long long)i, ops[i]);
        const int rmax = code == QECMC_TORIC ? L : code == QECMC_PLANAR ? (ops[i] == 1 ? L - 1 : L) : (ops[i] == 1 ? L - 1 : (L - 1) / 2);
        const int cmax = code == QECMC_TORIC ? L : code == QECMC_PLANAR ? (ops[i] == 1 ? L : L - 1) : (ops[i] == 1 ? L - 1 : 4);
        if (rows[i] < 0 || rows[i] >= rmax || cols[i] < 0 || cols[i] >= cmax) return fail(QECMC_ERR_INVALID, "stabilizer %llu: (row,col)=(%d,%d) outside [0,%d)x[0,%d)", (unsigned long long)i, rows[i], cols[i], rmax, cmax);
    }
    DevBuf din, dout, dr, dc, dop, dd;
    HIP_TRY(din.alloc(N * nq)); HIP_TRY(dout.alloc(N * nq));
    HIP_TRY(dr.alloc(N * 4)); HIP_TRY(dc.alloc(N * 4)); HIP_TRY(dop.alloc(N * 4)); HIP_TRY(dd.alloc(N * 4));
    HIP_TRY(hipMemcpy(din.p, in, N * nq, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dr.p, rows, N * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dc.p, cols, N * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dop.p, ops, N * 4, hipMemcpyHostToDevice));
    HIP_TRY(launch_apply_stabilizer(code, L, N, din.as<uint8_t>(), dout.as<uint8_t>(), dr.as<int32_t>(), dc.as<int32_t>(), dop.as<int32_t>(), dd.as<int32_t>(), 0));
    HIP_TRY(hipMemcpy(out, dout.p, N * nq, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dE, dd.p, N * 4, hipMemcpyDeviceToHost));
    return 0;
}

int qecmc_apply_logical(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *ops,
                        const int32_t *layers, const int32_t *xpos, const int32_t *zpos, int32_t *dE)
{
    PRIM_PROLOGUE();
    if (!in || !out || !ops || !layers || !xpos || !zpos || !dE) return fail(QECMC_ERR_INVALID, "NULL buffer");
    for (uint64_t i = 0; i < N; ++i) {
        if (ops[i] < 0 || ops[i] > 3) return fail(QECMC_ERR_INVALID, "logical %llu: operator %d outside [0,3]", (unsigned long long)i, ops[i]);
        if (layers[i] != 0 && layers[i] != 1) return fail(QECMC_ERR_INVALID, "logical %llu: layer %d is not 0 or 1", (unsigned long long)i, layers[i]);
        if (xpos[i] < 0 || xpos[i] >= L || zpos[i] < 0 || zpos[i] >= L) return fail(QECMC_ERR_INVALID, "logical %llu: position outside [0,%d)", (unsigned long long)i, L);
    }
    DevBuf din, dout, d0, d1, d2, d3, dd;
    HIP_TRY(din.alloc(N * nq)); HIP_TRY(dout.alloc(N * nq));
    HIP_TRY(d0.alloc(N * 4)); HIP_TRY(d1.alloc(N * 4)); HIP_TRY(d2.alloc(N * 4)); HIP_TRY(d3.alloc(N * 4)); HIP_TRY(dd.alloc(N * 4));
    HIP_TRY(hipMemcpy(din.p, in, N * nq, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d0.p, ops, N * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d1.p, layers, N * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d2.p, xpos, N * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d3.p, zpos, N * 4, hipMemcpyHostToDevice));
    HIP_TRY(launch_apply_logical(code, L, N, din.as<uint8_t>(), dout.as<uint8_t>(), d0.as<int32_t>(), d1.as<int32_t>(), d2.as<int32_t>(), d3.as<int32_t>(), dd.as<int32_t>(), 0));
    HIP_TRY(hipMemcpy(out, dout.p, N * nq, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dE, dd.p, N * 4, hipMemcpyDeviceToHost));
    return 0;
}

int qecmc_count_errors(int code, int L, uint64_t N, const uint8_t *in, int64_t *n)
{
    PRIM_PROLOGUE();
    if (!in || !n) return fail(QECMC_ERR_INVALID, "NULL buffer");
    DevBuf din, dn;
    HIP_TRY(din.alloc(N * nq)); HIP_TRY(dn.alloc(N * 8));
    HIP_TRY(hipMemcpy(din.p, in, N * nq, hipMemcpyHostToDevice));
    HIP_TRY(launch_count_errors((int)nq, N, din.as<uint8_t>(), dn.as<int64_t>(), 0));
    HIP_TRY(hipMemcpy(n, dn.p, N * 8, hipMemcpyDeviceToHost));
    return 0;
}

int qecmc_eq_class(int code, int L, uint64_t N, const uint8_t *in, int32_t *cls)
{
    PRIM_PROLOGUE();
    if (!in || !cls) return fail(QECMC_ERR_INVALID, "NULL buffer");
    DevBuf din, dc;
    HIP_TRY(din.alloc(N * nq)); HIP_TRY(dc.alloc(N * 4));
    HIP_TRY(hipMemcpy(din.p, in, N * nq, hipMemcpyHostToDevice));
    HIP_TRY(launch_eq_class(code, L, N, din.as<uint8_t>(), dc.as<int32_t>(), 0));
    HIP_TRY(hipMemcpy(cls, dc.p, N * 4, hipMemcpyDeviceToHost));
    return 0;
}

int qecmc_to_class(int code, int L, uint64_t N, const uint8_t *in, uint8_t *out, const int32_t *eq)
{
    PRIM_PROLOGUE();
    if (!in || !out || !eq) return fail(QECMC_ERR_INVALID, "NULL buffer");
    if (code != QECMC_TORIC) return fail(QECMC_ERR_UNSUPPORTED, "to_class exists for the toric code only (toric_model.py:354)");
    for (uint64_t i = 0; i < N; ++i)
        if (eq[i] < 0 || eq[i] > 15) return fail(QECMC_ERR_INVALID, "to_class %llu: class %d outside [0,16)", (unsigned long long)i, eq[i]);
    DevBuf din, dout, de;
    HIP_TRY(din.alloc(N * nq)); HIP_TRY(dout.alloc(N * nq)); HIP_TRY(de.alloc(N * 4));
    HIP_TRY(hipMemcpy(din.p, in, N * nq, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(de.p, eq, N * 4, hipMemcpyHostToDevice));
    HIP_TRY(launch_to_class(L, N, din.as<uint8_t>(), dout.as<uint8_t>(), de.as<int32_t>(), 0));
    HIP_TRY(hipMemcpy(out, dout.p, N * nq, hipMemcpyDeviceToHost));
    return 0;
}

int qecmc_syndrome(int code, int L, uint64_t N, const uint8_t *in, uint8_t *defects_out)
{
    PRIM_PROLOGUE();
    if (!in || !defects_out) return fail(QECMC_ERR_INVALID, "NULL buffer");
    const size_t nd = code == QECMC_TORIC ? nq : code == QECMC_PLANAR ? (size_t)2 * L * (L - 1) : (size_t)(L + 1) * (L + 1);
    DevBuf din, dout;
    HIP_TRY(din.alloc(N * nq)); HIP_TRY(dout.alloc(N * nd));
    HIP_TRY(hipMemcpy(din.p, in, N * nq, hipMemcpyHostToDevice));
    HIP_TRY(launch_syndrome(code, L, N, din.as<uint8_t>(), dout.as<uint8_t>(), 0));
    HIP_TRY(hipMemcpy(defects_out, dout.p, N * nd, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------- syndrome generation
static int generate_args(int code, int L, uint64_t N, double p_x, double p_y, double p_z, int hide_class, uint64_t seed,
                         uint32_t first_syndrome, GenArgs &a)
{
    if (int rc = check_code_L(code, L)) return rc;
    if (!(p_x >= 0.0) || !(p_y >= 0.0) || !(p_z >= 0.0) || !(p_x + p_y + p_z <= 1.0))
        return fail(QECMC_ERR_INVALID, "(p_x, p_y, p_z) = (%g, %g, %g) must be non-negative with a sum <= 1", p_x, p_y, p_z);
    if (code == QECMC_TORIC && !(p_x == p_y && p_y == p_z))
        return fail(QECMC_ERR_INVALID, "the toric model's generate_random_error(p) draws the Pauli uniformly (toric_model.py:15-23): pass p_x = p_y = p_z = p / 3");
    if (N + first_syndrome > 0xFFFFFFFFull) return fail(QECMC_ERR_INVALID, "global syndrome index exceeds 32 bits");
    std::memset(&a, 0, sizeof a);
    a.N = N; a.code = code; a.L = L; a.nq = (int)code_nq(code, L); a.hide = hide_class != 0;
    a.first_syndrome = first_syndrome; a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32);
    if (code == QECMC_TORIC) a.thr_z = thr64((p_x + p_y) + p_z);
    else { a.thr_z = thr64(p_z); a.thr_zx = thr64(p_z + p_x); a.thr_zxy = thr64((p_z + p_x) + p_y); }
    return 0;
}

int qecmc_generate_syndromes_dev(int code, int L, uint64_t N, double p_x, double p_y, double p_z, int hide_class, uint64_t seed,
                                 uint32_t first_syndrome, void *d_init_out, void *d_raw_out, void *d_eq_true_out, void *hip_stream)
{
    GenArgs a;
    if (int rc = generate_args(code, L, N, p_x, p_y, p_z, hide_class, seed, first_syndrome, a)) return rc;
    if (N == 0) return 0;
    if (!d_init_out) return fail(QECMC_ERR_INVALID, "NULL device buffer");
    a.out = static_cast<uint8_t *>(d_init_out); a.raw = static_cast<uint8_t *>(d_raw_out); a.eq_true = static_cast<int32_t *>(d_eq_true_out);
    HIP_TRY(launch_generate(a, static_cast<hipStream_t>(hip_stream)));
    return 0;
}

int qecmc_generate_syndromes(int code, int L, uint64_t N, double p_x, double p_y, double p_z, int hide_class, uint64_t seed,
                             uint32_t first_syndrome, uint8_t *init_out, uint8_t *raw_out, int32_t *eq_true_out)
{
    GenArgs a;
    if (int rc = generate_args(code, L, N, p_x, p_y, p_z, hide_class, seed, first_syndrome, a)) return rc;
    if (int rc = use_device(0)) return rc;
    if (N == 0) return 0;
    if (!init_out) return fail(QECMC_ERR_INVALID, "NULL buffer");
    const size_t nq = (size_t)a.nq;
    DevBuf dout, draw, deq;
    HIP_TRY(dout.alloc(N * nq));
    if (raw_out) HIP_TRY(draw.alloc(N * nq));
    if (eq_true_out) HIP_TRY(deq.alloc(N * 4));
    a.out = dout.as<uint8_t>(); a.raw = raw_out ? draw.as<uint8_t>() : nullptr; a.eq_true = eq_true_out ? deq.as<int32_t>() : nullptr;
    HIP_TRY(launch_generate(a, 0));
    HIP_TRY(hipMemcpy(init_out, dout.p, N * nq, hipMemcpyDeviceToHost));
    if (raw_out) HIP_TRY(hipMemcpy(raw_out, draw.p, N * nq, hipMemcpyDeviceToHost));
    if (eq_true_out) HIP_TRY(hipMemcpy(eq_true_out, deq.p, N * 4, hipMemcpyDeviceToHost));
    return 0;
}

// ---------------------------------------------------------------- chain / ladder
// (the drop-in calls gather their small buffers into one host block -- per thread, kept between calls -- and move it with one copy each way)
static std::vector<uint8_t> &staging()
{
    static thread_local std::vector<uint8_t> h;
    return h;
}
// (a large batch through these entry points must not leave its host copy behind in the thread)
struct StagingTrim {
    ~StagingTrim() { std::vector<uint8_t> &h = staging(); if (h.capacity() > (size_t(8) << 20)) std::vector<uint8_t>().swap(h); }
};

static int chain_update_impl(int code, int L, uint64_t N, uint8_t *states_inout, double p, double eta, int noise,
                             double p_logical, uint64_t iters, uint64_t seed, uint32_t first_syndrome, uint32_t slot, uint64_t k0,
                             uint8_t *accepted_out = nullptr)
{
    PRIM_PROLOGUE();
    if (!states_inout) return fail(QECMC_ERR_INVALID, "NULL buffer");
    if (noise == QECMC_NOISE_ALPHA) {      // `eta` carries alpha here
        if (!(p > 0.0) || !(p <= 1.0) || !(eta > 0.0)) return fail(QECMC_ERR_INVALID, "alpha noise needs pz_tilde in (0,1] and alpha > 0 (pz_tilde=%g alpha=%g)", p, eta);
        if (code == QECMC_TORIC || code == QECMC_PLANAR) return fail(QECMC_ERR_UNSUPPORTED, "alpha noise is built for the xzzx and rotated codes");
    } else
    if (noise) {
        if (!(p > 0.0) || !(p < 1.0) || !(eta > 0.0)) return fail(QECMC_ERR_INVALID, "biased noise needs p in (0,1) and eta > 0 (p=%g eta=%g)", p, eta);
        if (code == QECMC_TORIC || code == QECMC_PLANAR) return fail(QECMC_ERR_UNSUPPORTED, "biased noise is built for the xzzx and rotated codes (BASELINE config 4)");
    } else if (!(p > 0.0) || !(p <= 0.75)) return fail(QECMC_ERR_INVALID, "p=%g must be in (0, 0.75]", p);
    if (!(p_logical >= 0.0) || !(p_logical <= 1.0)) return fail(QECMC_ERR_INVALID, "p_logical=%g must be in [0,1]", p_logical);
    if (slot >= 0x100u) return fail(QECMC_ERR_INVALID, "slot %u collides with the swap stream id", slot);
    ChainArgs a;
    std::memset(&a, 0, sizeof a);
    const double f = noise ? 0.0 : chain_factor(p);
    std::vector<uint32_t> tbl(nq + 1, 0u);
    for (size_t d = 1; d <= nq && !noise; ++d) tbl[d] = thr32(std::pow(f, (double)d));
    for (int d = 1; d <= 4 && !noise; ++d) a.acc44[d] = thr44(std::pow(f, (double)d));
    const std::vector<double> bt = noise == QECMC_NOISE_ALPHA ? alpha_tables(p, eta, nq) : bias_tables(p, noise ? eta : 1.0, nq);
    // one device block, one copy each way: [bias table | acceptance table | states | accepted]
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };     // (every part on a 256-byte boundary, as hipMalloc would place it)
    const size_t o_tbl = up(bt.size() * 8), o_st = o_tbl + up(tbl.size() * 4), o_acc = o_st + up(N * nq), total = o_acc + (accepted_out ? N : 0);
    StagingTrim trim;
    std::vector<uint8_t> &h = staging();
    h.resize(o_acc);
    std::memcpy(h.data(), bt.data(), bt.size() * 8);
    std::memcpy(h.data() + o_tbl, tbl.data(), tbl.size() * 4);
    std::memcpy(h.data() + o_st, states_inout, N * nq);
    DevBuf d;
    HIP_TRY(d.alloc(total));
    HIP_TRY(hipMemcpy(d.p, h.data(), o_acc, hipMemcpyHostToDevice));
    if (accepted_out) a.accepted = d.as<uint8_t>() + o_acc;
    a.states = d.as<uint8_t>() + o_st; a.N = N; a.iters = iters; a.k0 = k0;
    a.thr_logical = p_logical > 0 ? thr64(p_logical) : 0;
    a.acc_tbl = reinterpret_cast<uint32_t *>(d.as<uint8_t>() + o_tbl); a.acc_all = !noise && f >= 1.0;
    a.first_syndrome = first_syndrome; a.slot = slot;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.L = L;
    a.code = code; a.noise = noise; a.bias_tbl = d.as<double>();
    HIP_TRY(launch_chain_update(a, 0));
    if (accepted_out) {
        h.resize(total - o_st);
        HIP_TRY(hipMemcpy(h.data(), d.as<uint8_t>() + o_st, total - o_st, hipMemcpyDeviceToHost));
        std::memcpy(states_inout, h.data(), N * nq);
        std::memcpy(accepted_out, h.data() + (o_acc - o_st), N);
    } else
        HIP_TRY(hipMemcpy(states_inout, d.as<uint8_t>() + o_st, N * nq, hipMemcpyDeviceToHost));
    return 0;
}

int qecmc_chain_update(int code, int L, uint64_t N, uint8_t *states_inout, double p, double p_logical,
                       uint64_t iters, uint64_t seed, uint32_t first_syndrome, uint32_t slot, uint64_t k0)
{
    return chain_update_impl(code, L, N, states_inout, p, 0.0, 0, p_logical, iters, seed, first_syndrome, slot, k0);
}

int qecmc_chain_update_biased(int code, int L, uint64_t N, uint8_t *states_inout, double p, double eta, double p_logical,
                              uint64_t iters, uint64_t seed, uint32_t first_syndrome, uint32_t slot, uint64_t k0)
{
    return chain_update_impl(code, L, N, states_inout, p, eta, 1, p_logical, iters, seed, first_syndrome, slot, k0);
}

int qecmc_chain_update_alpha(int code, int L, uint64_t N, uint8_t *states_inout, double pz_tilde, double alpha, double p_logical,
                             uint64_t iters, uint64_t seed, uint32_t first_syndrome, uint32_t slot, uint64_t k0, uint8_t *accepted_out)
{
    return chain_update_impl(code, L, N, states_inout, pz_tilde, alpha, QECMC_NOISE_ALPHA, p_logical, iters, seed, first_syndrome, slot, k0, accepted_out);
}

// Chain_xyz.update_chain_fast(iters), src/mcmc.py:106-114,162-173, on N independent chains: a stabilizer generator is proposed
// (planar_model._apply_random_stabilizer in the reference, the code's own here) and accepted with probability
// prod_i (p_i / (1 - sum p))^(change of n_i), i = x, y, z.  Draws as the non-top rule of qecmc_chain_update.
int qecmc_chain_update_xyz(int code, int L, uint64_t N, uint8_t *states_inout, const double *p_xyz, uint64_t iters, uint64_t seed,
                           uint32_t first_syndrome, uint32_t slot, uint64_t k0)
{
    PRIM_PROLOGUE();
    if (!states_inout || !p_xyz) return fail(QECMC_ERR_INVALID, "NULL buffer");
    const double tot = (p_xyz[0] + p_xyz[1]) + p_xyz[2];
    if (!(p_xyz[0] > 0) || !(p_xyz[1] > 0) || !(p_xyz[2] > 0) || !(tot < 1.0)) return fail(QECMC_ERR_INVALID, "p_xyz=(%g,%g,%g) must be positive with a sum below 1", p_xyz[0], p_xyz[1], p_xyz[2]);
    if (slot >= 0x100u) return fail(QECMC_ERR_INVALID, "slot %u collides with the swap stream id", slot);
    const double f[3] = {p_xyz[0] / (1.0 - tot), p_xyz[1] / (1.0 - tot), p_xyz[2] / (1.0 - tot)};          // mcmc.py:110
    std::vector<uint64_t> thr(729);
    for (int dx = -4; dx <= 4; ++dx)
        for (int dy = -4; dy <= 4; ++dy)
            for (int dz = -4; dz <= 4; ++dz)
                thr[((dx + 4) * 9 + (dy + 4)) * 9 + (dz + 4)] = thr44((std::pow(f[0], (double)dx) * std::pow(f[1], (double)dy)) * std::pow(f[2], (double)dz));   // :170
    ChainArgs a;
    std::memset(&a, 0, sizeof a);
    DevBuf dthr, dst;
    HIP_TRY(dthr.alloc(729 * 8)); HIP_TRY(dst.alloc(N * nq));
    HIP_TRY(hipMemcpy(dthr.p, thr.data(), 729 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dst.p, states_inout, N * nq, hipMemcpyHostToDevice));
    a.states = dst.as<uint8_t>(); a.N = N; a.iters = iters; a.k0 = k0; a.first_syndrome = first_syndrome; a.slot = slot;
    a.seed_lo = (uint32_t)seed; a.seed_hi = (uint32_t)(seed >> 32); a.L = L; a.code = code; a.noise = 0;
    a.xyz_thr = dthr.as<uint64_t>();
    HIP_TRY(launch_chain_update(a, 0));
    HIP_TRY(hipMemcpy(states_inout, dst.p, N * nq, hipMemcpyDeviceToHost));
    return 0;
}

// The step entry points' plan cache: keyed by the whole parameter block (the fields a plan does not depend on -- seed, first_syndrome,
// steps -- are constant for one ladder anyway), a handful of entries, least recently used out first.  Plans are immutable once
// built (a launch works on a copy of `args`), so concurrent callers may share one.
// (the key: exactly the fields build_plan reads, in a zeroed struct -- not the caller's block with its padding, seed, first_syndrome
// and step count, which would make a cache miss of every ladder with its own seed)
static qecmc_params plan_key(const qecmc_params &p)
{
    qecmc_params k;
    std::memset(&k, 0, sizeof k);
    k.abi_size = p.abi_size; k.code = p.code; k.L = p.L; k.Nc = p.Nc; k.noise = p.noise; k.scan = p.scan; k.conv_mode = p.conv_mode; k.device = p.device;
    k.iters = p.iters; k.tops_burn = p.tops_burn; k.TOPS = p.TOPS; k.SEQ = p.SEQ; k.replicas = p.replicas; k.eps = p.eps; k.p = p.p; k.eta = p.eta;
    k.alpha = p.alpha; k.p_logical = p.p_logical; k.flags = p.flags;
    return k;
}
static int cached_plan(const qecmc_params &pin, std::shared_ptr<const qecmc_plan> *out)
{
    const qecmc_params p = plan_key(pin);
    struct Entry { qecmc_params key; std::shared_ptr<const qecmc_plan> plan; uint64_t used; };
    static std::mutex mu;
    static std::vector<Entry> *cache = new std::vector<Entry>;     // (never destroyed, like the block pool)
    static uint64_t tick = 0;
    constexpr size_t kEntries = 8;
    {
        std::lock_guard<std::mutex> g(mu);
        for (Entry &e : *cache)
            if (std::memcmp(&e.key, &p, sizeof p) == 0) { e.used = ++tick; *out = e.plan; return 0; }
    }
    auto pl = std::make_shared<qecmc_plan>();
    if (int rc = build_plan(&pin, pl.get())) return rc;
    *out = pl;
    std::lock_guard<std::mutex> g(mu);
    if (cache->size() >= kEntries) {
        size_t lru = 0;
        for (size_t i = 1; i < cache->size(); ++i) if ((*cache)[i].used < (*cache)[lru].used) lru = i;
        // (its tables are freed when the last caller still stepping with it returns; every such call ends with a blocking copy)
        (*cache)[lru] = Entry{p, pl, ++tick};
    } else
        cache->push_back(Entry{p, pl, ++tick});
    return 0;
}

static int ladder_step_impl(const qecmc_params *params, uint64_t N, uint8_t *states_inout, uint8_t *flags_inout,
                            uint32_t *tops0_inout, uint16_t *neff_inout, uint64_t iters, uint64_t nsteps, uint64_t step0, uint64_t prop0)
{
    if (!params) return fail(QECMC_ERR_INVALID, "params is NULL");
    qecmc_params p = *params;
    p.iters = iters;
    p.conv_mode = QECMC_CONV_NONE;
    p.replicas = 0;
    if (int rc = validate_params(&p)) return rc;
    if (p.scan == QECMC_SCAN_COLOUR) return fail(QECMC_ERR_UNSUPPORTED, "scan = colour starts its ladders from seed configurations (qecmc_pteq_batch / qecmc_pteq_launch_dev)");
    if (!states_inout || !flags_inout || !tops0_inout) return fail(QECMC_ERR_INVALID, "NULL buffer");
    if ((p.noise == QECMC_NOISE_ALPHA) != (neff_inout != nullptr))
        return fail(QECMC_ERR_INVALID, "alpha-noise ladders step through qecmc_ladder_step_alpha (which carries the slots' n_eff), the others through qecmc_ladder_step");
    if (int rc = use_device(p.device)) return rc;
    if (N == 0) return 0;
    // a Python loop over Ladder.step presents the same parameters every call: the tables of the last few plans are kept
    std::shared_ptr<const qecmc_plan> plan;
    if (int rc = cached_plan(p, &plan)) return rc;
    const qecmc_plan &pl = *plan;
    const size_t nq = pl.args.nq, Nc = pl.args.Nc;
    // one device block, one copy each way: [n_eff records | tops0 | states | flags]
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };     // (every part on a 256-byte boundary, as hipMalloc would place it)
    const size_t o_t0 = neff_inout ? up(N * Nc * 4) : 0, o_st = o_t0 + up(N * 4), o_fl = o_st + up(N * Nc * nq), total = o_fl + N * Nc;
    StagingTrim trim;
    std::vector<uint8_t> &h = staging();
    h.resize(total);
    if (neff_inout) std::memcpy(h.data(), neff_inout, N * Nc * 4);
    std::memcpy(h.data() + o_t0, tops0_inout, N * 4);
    std::memcpy(h.data() + o_st, states_inout, N * Nc * nq);
    std::memcpy(h.data() + o_fl, flags_inout, N * Nc);
    DevBuf d;
    HIP_TRY(d.alloc(total));
    HIP_TRY(hipMemcpy(d.p, h.data(), total, hipMemcpyHostToDevice));
    LadderArgs a = pl.args;
    if (neff_inout) a.neff = d.as<uint32_t>();
    a.states = d.as<uint8_t>() + o_st; a.flags = d.as<uint8_t>() + o_fl; a.tops0 = reinterpret_cast<uint32_t *>(d.as<uint8_t>() + o_t0);
    a.N = N; a.first_syndrome = p.first_syndrome; a.step0 = step0; a.prop0 = prop0; a.nsteps = nsteps;
    a.seed_lo = (uint32_t)p.seed; a.seed_hi = (uint32_t)(p.seed >> 32);      // (the cached tables do not depend on the seed: patched per call)
    a.resume = 1; a.write_states = 1;
    HIP_TRY(launch_ladder_rs_toric(a, 0));
    HIP_TRY(hipMemcpy(h.data(), d.p, total, hipMemcpyDeviceToHost));
    if (neff_inout) std::memcpy(neff_inout, h.data(), N * Nc * 4);
    std::memcpy(tops0_inout, h.data() + o_t0, N * 4);
    std::memcpy(states_inout, h.data() + o_st, N * Nc * nq);
    std::memcpy(flags_inout, h.data() + o_fl, N * Nc);
    return 0;
}

int qecmc_ladder_step(const qecmc_params *params, uint64_t N, uint8_t *states_inout, uint8_t *flags_inout,
                      uint32_t *tops0_inout, uint64_t iters, uint64_t nsteps, uint64_t step0, uint64_t prop0)
{
    return ladder_step_impl(params, N, states_inout, flags_inout, tops0_inout, nullptr, iters, nsteps, step0, prop0);
}

int qecmc_ladder_step_alpha(const qecmc_params *params, uint64_t N, uint8_t *states_inout, uint8_t *flags_inout,
                            uint32_t *tops0_inout, uint16_t *neff_counts_inout, uint64_t iters, uint64_t nsteps,
                            uint64_t step0, uint64_t prop0)
{
    if (!neff_counts_inout) return fail(QECMC_ERR_INVALID, "NULL buffer");
    return ladder_step_impl(params, N, states_inout, flags_inout, tops0_inout, neff_counts_inout, iters, nsteps, step0, prop0);
}

// ---------------------------------------------------------------- PTEQ batch
int qecmc_plan_create(const qecmc_params *params, qecmc_plan **plan_out)
{
    if (!plan_out) return fail(QECMC_ERR_INVALID, "plan_out is NULL");
    *plan_out = nullptr;
    if (int rc = validate_params(params)) return rc;
    if (int rc = use_device(params->device)) return rc;
    qecmc_plan *pl = new (std::nothrow) qecmc_plan();
    if (!pl) return fail(QECMC_ERR_INVALID, "out of host memory");
    if (int rc = build_plan(params, pl)) { delete pl; return rc; }
    *plan_out = pl;
    return 0;
}

int qecmc_plan_destroy(qecmc_plan *plan)
{
    // the plan's tables may go back to the block pool and on to another call: wait for the launches that read them (what hipFree used
    // to do) -- on the plan's device, leaving the caller's current device as it was
    if (plan) {
        int cur = -1;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        (void)hipSetDevice(plan->prm.device);
        (void)hipDeviceSynchronize();
        if (have && cur != plan->prm.device) (void)hipSetDevice(cur);
    }
    delete plan;
    return 0;
}

int qecmc_plan_info(const qecmc_plan *plan, uint32_t *lds_bytes, uint32_t *block_threads, uint32_t *syndromes_per_block)
{
    if (!plan) return fail(QECMC_ERR_INVALID, "plan is NULL");
    if (lds_bytes) *lds_bytes = (uint32_t)plan->lds_bytes;
    if (block_threads) *block_threads = (uint32_t)plan->args.Nc * 64u;
    if (syndromes_per_block) *syndromes_per_block = kSynPerBlock;
    return 0;
}

// THE workspace formula of the criterion runs (the one place it lives): one log entry per (ladder step, column) -- the bottom chain's
// error count (u16), or for alpha noise the two counts behind n_eff (2 x u16) -- with one column per ladder (rounded up to whole
// 64-lane groups), or, when the launch runs on the plan's persistent grid with a work queue, one per lane of that grid.
static bool launch_takes_queue(const qecmc_plan *plan, bool wants_states_or_stats)
{
    return plan->queue_grid != 0 && !wants_states_or_stats && plan->prm.steps > 0;
}
// the persistent grid a scan = wave criterion launch of M ladders runs on, and the ladders each of its workgroups owns
static void wave_queue_shape(const qecmc_plan *plan, uint64_t M, uint32_t *grid, uint32_t *chunk)
{
    // (whole groups of 64 per workgroup: a batch that gives every ladder a lane of its own is laid out like a launch without the queue,
    // ladder l in lane l & 63 of workgroup l >> 6, and gives the same results whatever the grid)
    const uint64_t groups = (M + 63) / 64, g = std::max<uint64_t>(1, std::min<uint64_t>(plan->queue_grid, groups)), c = ((M + g - 1) / g + 63) / 64 * 64;
    *chunk = (uint32_t)c;
    *grid = (uint32_t)((M + c - 1) / c);
}
static uint64_t workspace_need(const qecmc_plan *plan, uint64_t N, bool queue)
{
    if (plan->prm.conv_mode != QECMC_CONV_ERROR_BASED) return 0;
    const uint64_t M = N * plan->args.replicas;
    uint64_t cols = (M + 63) / 64 * 64;
    if (queue && plan->args.scan == QECMC_SCAN_WAVE) {
        uint32_t grid, chunk;
        wave_queue_shape(plan, M, &grid, &chunk);
        cols = (uint64_t)grid * 64u;
    } else if (queue) {
        cols = std::min<uint64_t>(cols, (uint64_t)plan->queue_grid * 64u);
    }
    return (plan->prm.noise == QECMC_NOISE_ALPHA ? 4ull : 2ull) * cols * plan->prm.steps;
}

int qecmc_plan_workspace_bytes(const qecmc_plan *plan, uint64_t N, int with_final_states, uint64_t *bytes_out)
{
    if (!plan || !bytes_out) return fail(QECMC_ERR_INVALID, "NULL argument");
    // what qecmc_pteq_launch_dev(plan, N syndromes, d_final_states given or not) will ask for: a launch that may take the plan's work
    // queue logs one column per lane of the persistent grid, any other one column per ladder
    *bytes_out = workspace_need(plan, N, launch_takes_queue(plan, with_final_states != 0 || plan->d_swap_acc != nullptr));
    return 0;
}

int qecmc_pteq_launch_dev(qecmc_plan *plan, const void *d_init, uint64_t N, uint32_t first_syndrome, void *d_counts,
                          void *d_samples, void *d_tops0, void *d_steps_done, void *d_converged, void *d_final_states,
                          void *d_workspace, uint64_t workspace_bytes, void *hip_stream)
{
    if (!plan) return fail(QECMC_ERR_INVALID, "plan is NULL");
    if (N == 0) return 0;
    if (!d_init || !d_counts || !d_samples) return fail(QECMC_ERR_INVALID, "NULL device buffer");
    const uint64_t R = plan->args.replicas, M = N * R;              // ladders
    if (M + first_syndrome > 0xFFFFFFFFull) return fail(QECMC_ERR_INVALID, "global ladder index (first_syndrome + N * replicas) exceeds 32 bits");
    if (plan->args.scan == QECMC_SCAN_WAVE && (first_syndrome & 63u))
        return fail(QECMC_ERR_INVALID, "scan = wave: first_syndrome=%u must be a multiple of 64 (a wavefront shares its generator picks)", first_syndrome);
    const bool takes_queue = launch_takes_queue(plan, plan->d_swap_acc != nullptr || d_final_states != nullptr);
    {
        const uint64_t need = workspace_need(plan, N, takes_queue);
        if (need && !d_workspace) return fail(QECMC_ERR_INVALID, "conv_mode error_based needs the workspace of qecmc_plan_workspace_bytes()");
        if (workspace_bytes < need)
            return fail(QECMC_ERR_INVALID, "workspace of %llu bytes, this launch logs %llu (qecmc_plan_workspace_bytes: 2 or 4 bytes per ladder step and column)",
                        (unsigned long long)workspace_bytes, (unsigned long long)need);
    }
    if (R > 1 && (plan->d_swap_acc || plan->d_nerr_sum)) return fail(QECMC_ERR_INVALID, "qecmc_plan_set_stats is per ladder: not with replicas > 1");
    if ((plan->d_swap_acc || plan->d_nerr_sum) && (uint64_t)plan->args.nq * plan->prm.steps > 0xFFFFFFFFull)
        return fail(QECMC_ERR_INVALID, "qecmc_plan_set_stats: nq * steps exceeds the 32-bit error-count sums");
    hipStream_t strm = static_cast<hipStream_t>(hip_stream);
    if (R > 1) {   // the R ladders of a syndrome add into its outputs
        HIP_TRY(hipMemsetAsync(d_counts, 0, N * plan->args.ncls * 4, strm));
        HIP_TRY(hipMemsetAsync(d_samples, 0, N * 4, strm));
        if (d_tops0) HIP_TRY(hipMemsetAsync(d_tops0, 0, N * 4, strm));
        if (d_steps_done) HIP_TRY(hipMemsetAsync(d_steps_done, 0, N * 4, strm));
        if (d_converged) HIP_TRY(hipMemsetAsync(d_converged, 1, N, strm));
    }
    LadderArgs a = plan->args;
    a.swap_acc = plan->d_swap_acc ? plan->d_swap_acc : nullptr;
    a.nerr_sum = plan->d_nerr_sum;
    if (a.nerr_sum && !a.swap_acc) return fail(QECMC_ERR_INVALID, "qecmc_plan_set_stats: d_nerr_sums needs d_swap_accepts");
    a.init = static_cast<const uint8_t *>(d_init);
    a.counts = static_cast<uint32_t *>(d_counts);
    a.samples = static_cast<uint32_t *>(d_samples);
    a.tops0 = static_cast<uint32_t *>(d_tops0);
    a.steps_done = static_cast<uint32_t *>(d_steps_done);
    a.converged = static_cast<uint8_t *>(d_converged);
    a.nlog = static_cast<uint16_t *>(d_workspace);
    a.states = static_cast<uint8_t *>(d_final_states);
    a.write_states = d_final_states != nullptr;
    a.N = M; a.first_syndrome = first_syndrome;
    a.step0 = 0; a.prop0 = 0; a.nsteps = plan->prm.steps; a.resume = 0;
    if (plan->args.scan == QECMC_SCAN_WAVE && plan->prm.conv_mode != QECMC_CONV_NONE && !takes_queue)
        return fail(QECMC_ERR_UNSUPPORTED, "scan = wave runs the criterion on its persistent grid, where a ladder's lane is reused when it has stopped: no final "
                    "states or per-ladder statistics with conv_mode error_based (and steps must be > 0)");
    if (takes_queue && plan->args.scan == QECMC_SCAN_WAVE) {
        // (the workgroups of the persistent grid own contiguous shares of the batch: no global counter)
        wave_queue_shape(plan, M, &a.grid_cap, &a.wu_chunk);
    } else if (takes_queue) {
        // (one launch at a time per plan: the counter belongs to the plan)
        HIP_TRY(hipMemsetAsync(plan->queue.p, 0, sizeof(uint32_t), strm));
        a.queue = plan->queue.as<uint32_t>();
        a.grid_cap = plan->queue_grid;
    }
    HIP_TRY(launch_ladder_rs_toric(a, strm));
    return 0;
}

int qecmc_plan_set_stats(qecmc_plan *plan, void *d_swap_accepts, void *d_nerr_sums)
{
    if (!plan) return fail(QECMC_ERR_INVALID, "plan is NULL");
    if (d_nerr_sums && !d_swap_accepts) return fail(QECMC_ERR_INVALID, "d_nerr_sums needs d_swap_accepts");
    if (d_swap_accepts && plan->args.Nc < 2) return fail(QECMC_ERR_INVALID, "swap statistics need Nc >= 2");
    if (d_swap_accepts && (plan->args.scan == QECMC_SCAN_COLOUR || plan->args.scan == QECMC_SCAN_WAVE)) return fail(QECMC_ERR_UNSUPPORTED, "swap statistics are not collected by the scan = colour / wave kernels");
    if (d_swap_accepts && plan->lds_bytes + ladder_stats_lds_bytes(plan->args.Nc) > 160 * 1024)
        return fail(QECMC_ERR_UNSUPPORTED, "no LDS left for the statistics counters at this L / Nc");
    plan->d_swap_acc = static_cast<uint32_t *>(d_swap_accepts);
    plan->d_nerr_sum = static_cast<uint32_t *>(d_nerr_sums);
    return 0;
}

int qecmc_pteq_resume_dev(qecmc_plan *plan, void *d_states, void *d_flags, void *d_tops0, uint64_t N,
                          uint32_t first_syndrome, uint64_t step0, void *d_counts, void *d_samples, void *hip_stream)
{
    if (!plan) return fail(QECMC_ERR_INVALID, "plan is NULL");
    if (N == 0) return 0;
    if (!d_states || !d_flags || !d_tops0) return fail(QECMC_ERR_INVALID, "NULL device buffer");
    if ((d_counts == nullptr) != (d_samples == nullptr)) return fail(QECMC_ERR_INVALID, "d_counts and d_samples go together");
    if (plan->prm.conv_mode != QECMC_CONV_NONE) return fail(QECMC_ERR_INVALID, "qecmc_pteq_resume_dev runs fixed-length chunks: conv_mode must be NONE");
    if (plan->args.replicas > 1) return fail(QECMC_ERR_INVALID, "qecmc_pteq_resume_dev continues single ladders: replicas must be <= 1");
    if (plan->args.noise == QECMC_NOISE_ALPHA) return fail(QECMC_ERR_UNSUPPORTED, "alpha-noise ladders carry n_eff: continue them with qecmc_ladder_step_alpha");
    if (plan->args.scan == QECMC_SCAN_COLOUR) return fail(QECMC_ERR_UNSUPPORTED, "scan = colour starts its ladders from seed configurations: no chunked continuation");
    if (N + first_syndrome > 0xFFFFFFFFull) return fail(QECMC_ERR_INVALID, "global syndrome index exceeds 32 bits");
    if (plan->args.scan == QECMC_SCAN_WAVE && (first_syndrome & 63u))
        return fail(QECMC_ERR_INVALID, "scan = wave: first_syndrome=%u must be a multiple of 64 (a wavefront shares its generator picks)", first_syndrome);
    LadderArgs a = plan->args;
    a.states = static_cast<uint8_t *>(d_states); a.flags = static_cast<uint8_t *>(d_flags); a.tops0 = static_cast<uint32_t *>(d_tops0);
    a.counts = static_cast<uint32_t *>(d_counts); a.samples = static_cast<uint32_t *>(d_samples);
    a.N = N; a.first_syndrome = first_syndrome;
    a.step0 = step0; a.prop0 = step0 * plan->prm.iters; a.nsteps = plan->prm.steps;
    a.resume = 1; a.write_states = 1; a.accumulate = 1;
    HIP_TRY(launch_ladder_rs_toric(a, static_cast<hipStream_t>(hip_stream)));
    return 0;
}

int qecmc_pteq_batch(const qecmc_params *params, const uint8_t *init, uint64_t N, uint32_t *counts_out,
                     uint32_t *samples_out, uint32_t *tops0_out, uint32_t *steps_done_out, uint8_t *converged_out,
                     uint8_t *final_states_out, qecmc_stats *stats_out)
{
    return qecmc_pteq_batch_stats(params, init, N, counts_out, samples_out, tops0_out, steps_done_out, converged_out,
                                  final_states_out, nullptr, nullptr, stats_out);
}

int qecmc_pteq_batch_stats(const qecmc_params *params, const uint8_t *init, uint64_t N, uint32_t *counts_out,
                           uint32_t *samples_out, uint32_t *tops0_out, uint32_t *steps_done_out, uint8_t *converged_out,
                           uint8_t *final_states_out, uint32_t *swap_accepts_out, uint32_t *nerr_sums_out, qecmc_stats *stats_out)
{
    const auto t0 = std::chrono::steady_clock::now();
    qecmc_plan *pl = nullptr;
    if (int rc = qecmc_plan_create(params, &pl)) return rc;
    struct Guard { qecmc_plan *p; ~Guard() { qecmc_plan_destroy(p); } } guard{pl};   // (synchronises: an error return may leave the launch running)
    if (N == 0) return 0;
    if (!init || !counts_out || !samples_out) return fail(QECMC_ERR_INVALID, "NULL buffer");
    const size_t nq = pl->args.nq, Nc = pl->args.Nc, ncls = pl->args.ncls, R = pl->args.replicas;
    // (the work-queue kernels log one column per lane of the persistent grid, not per ladder: workspace_need knows)
    const uint64_t ws_bytes = workspace_need(pl, N, launch_takes_queue(pl, swap_accepts_out || nerr_sums_out || final_states_out));
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (ws_bytes > free_b / 2)
        return fail(QECMC_ERR_INVALID, "conv_mode error_based needs %llu bytes of workspace (2*N*steps), %zu free: lower "
                    "`steps` or the batch size", (unsigned long long)ws_bytes, free_b);
    DevBuf di, dc, ds, dt, dsd, dcv, df, dw;
    HIP_TRY(di.alloc(N * nq)); HIP_TRY(dc.alloc(N * ncls * 4)); HIP_TRY(ds.alloc(N * 4)); HIP_TRY(dt.alloc(N * 4));
    HIP_TRY(dsd.alloc(N * 4)); HIP_TRY(dcv.alloc(N));
    if (ws_bytes) HIP_TRY(dw.alloc(ws_bytes));
    if (final_states_out) HIP_TRY(df.alloc(N * R * Nc * nq));
    DevBuf dsa, dns;
    if (swap_accepts_out || nerr_sums_out) {
        HIP_TRY(dsa.alloc(N * (Nc > 1 ? Nc - 1 : 1) * 4));
        if (nerr_sums_out) HIP_TRY(dns.alloc(N * Nc * 4));
        if (int rc = qecmc_plan_set_stats(pl, dsa.p, nerr_sums_out ? dns.p : nullptr)) return rc;
    }
    HIP_TRY(hipMemcpy(di.p, init, N * nq, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, 0));
    int rc = qecmc_pteq_launch_dev(pl, di.p, N, params->first_syndrome, dc.p, ds.p, dt.p, dsd.p, dcv.p,
                                   final_states_out ? df.p : nullptr, ws_bytes ? dw.p : nullptr, ws_bytes, nullptr);
    if (rc) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return rc; }
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIP_TRY(hipMemcpy(counts_out, dc.p, N * ncls * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(samples_out, ds.p, N * 4, hipMemcpyDeviceToHost));
    if (tops0_out) HIP_TRY(hipMemcpy(tops0_out, dt.p, N * 4, hipMemcpyDeviceToHost));
    if (steps_done_out) HIP_TRY(hipMemcpy(steps_done_out, dsd.p, N * 4, hipMemcpyDeviceToHost));
    if (converged_out) HIP_TRY(hipMemcpy(converged_out, dcv.p, N, hipMemcpyDeviceToHost));
    if (final_states_out) HIP_TRY(hipMemcpy(final_states_out, df.p, N * R * Nc * nq, hipMemcpyDeviceToHost));
    if (swap_accepts_out) HIP_TRY(hipMemcpy(swap_accepts_out, dsa.p, N * (Nc - 1) * 4, hipMemcpyDeviceToHost));
    if (nerr_sums_out) HIP_TRY(hipMemcpy(nerr_sums_out, dns.p, N * Nc * 4, hipMemcpyDeviceToHost));
    if (stats_out) {
        stats_out->proposals = N * R * Nc * params->iters * params->steps;   // upper bound when the criterion stops early
        stats_out->swap_tests = N * R * (Nc - 1) * params->steps;
        stats_out->kernel_ms = ms;
        stats_out->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

int qecmc_ptdc_batch(const qecmc_params *params, const uint8_t *init, uint64_t N, int32_t droplets, uint32_t flags,
                     uint32_t *hist_out, uint32_t *m_out, qecmc_stats *stats_out)
{
    return qecmc_ptdc_batch_conv(params, init, N, droplets, flags, 0.0, hist_out, m_out, nullptr, stats_out);
}

int qecmc_ptdc_batch_conv(const qecmc_params *params, const uint8_t *init, uint64_t N, int32_t droplets, uint32_t flags,
                          double conv_mult, uint32_t *hist_out, uint32_t *m_out, uint32_t *steps_done_out, qecmc_stats *stats_out)
{
    return qecmc_ptdc_batch_xyz(params, init, N, droplets, flags, conv_mult, nullptr, hist_out, m_out, steps_done_out, nullptr, nullptr, stats_out);
}

int qecmc_ptdc_batch_xyz(const qecmc_params *params, const uint8_t *init, uint64_t N, int32_t droplets, uint32_t flags,
                         double conv_mult, const double *p_xyz_sampling, uint32_t *hist_out, uint32_t *m_out,
                         uint32_t *steps_done_out, uint32_t *xyz_out, uint32_t *xyz_count_out, qecmc_stats *stats_out)
{
    const auto t0 = std::chrono::steady_clock::now();
    if (!params) return fail(QECMC_ERR_INVALID, "params is NULL");
    if (!(conv_mult >= 0.0) || !std::isfinite(conv_mult)) return fail(QECMC_ERR_INVALID, "conv_mult=%g must be finite and >= 0", conv_mult);
    qecmc_params p = *params;
    p.p_logical = 0.0;                                   // Ladder(p_sampling, code, Nc): decoders.py:182,196
    p.conv_mode = QECMC_CONV_NONE;
    p.replicas = 0;                                      // (droplets are this entry point's own replica dimension)
    std::vector<uint64_t> xyz_thr;
    if (p_xyz_sampling) {
        // Chain_xyz (mcmc.py:106-114): factors = p_xyz / (1 - p_xyz.sum()), accept iff u < (factors ** change).prod() (:170)
        const double *q = p_xyz_sampling;
        const double tot = (q[0] + q[1]) + q[2];
        if (!(q[0] > 0) || !(q[1] > 0) || !(q[2] > 0) || !(tot < 1.0)) return fail(QECMC_ERR_INVALID, "p_xyz_sampling=(%g,%g,%g) must be positive with a sum below 1", q[0], q[1], q[2]);
        if (p.Nc != 1) return fail(QECMC_ERR_INVALID, "Chain_xyz is a single chain (mcmc.py:106): Nc=%d must be 1", p.Nc);
        if (p.code == QECMC_TORIC) return fail(QECMC_ERR_UNSUPPORTED, "Chain_xyz is built for the planar, xzzx and rotated codes (the reference's runs the planar stencil, mcmc.py:164)");
        const double f[3] = {q[0] / (1.0 - tot), q[1] / (1.0 - tot), q[2] / (1.0 - tot)};
        xyz_thr.resize(729);
        for (int dx = -4; dx <= 4; ++dx)
            for (int dy = -4; dy <= 4; ++dy)
                for (int dz = -4; dz <= 4; ++dz) {
                    const double w = (std::pow(f[0], (double)dx) * std::pow(f[1], (double)dy)) * std::pow(f[2], (double)dz);
                    xyz_thr[((dx + 4) * 9 + (dy + 4)) * 9 + (dz + 4)] = thr44(w);      // u < w  <=>  v44 < ceil(w 2^44)
                }
        p.p = tot <= 0.75 ? tot : 0.75;                                           // unused by the rule; keeps the plan's tables valid
    }
    if (p.noise == QECMC_NOISE_ALPHA) {
        // STDC_droplet_alpha (decoders.py:510-534): Chain_alpha single chains, `update_chain(5)` per step
        if (p.Nc != 1) return fail(QECMC_ERR_UNSUPPORTED, "the unique-chain estimators run Chain_alpha as single chains (decoders.py:510): Nc=%d must be 1", p.Nc);
        if (p_xyz_sampling) return fail(QECMC_ERR_INVALID, "p_xyz_sampling and alpha noise exclude each other");
    } else
    if (p.noise != QECMC_NOISE_DEPOLARIZING) return fail(QECMC_ERR_UNSUPPORTED, "PTDC is defined for the depolarizing ladder (decoders.py:168)");
    if (p.scan != QECMC_SCAN_RANDOM) return fail(QECMC_ERR_UNSUPPORTED, "PTDC runs the reference's random-scan ladder");
    if (droplets < 1) return fail(QECMC_ERR_INVALID, "droplets=%d must be >= 1", droplets);
    if (flags & ~3u) return fail(QECMC_ERR_INVALID, "unknown flags 0x%x", flags);
    const bool init_per_droplet = flags & QECMC_PTDC_INIT_PER_DROPLET, per_rung = flags & QECMC_PTDC_SET_PER_RUNG;
    qecmc_plan *pl = nullptr;
    if (int rc = qecmc_plan_create(&p, &pl)) return rc;
    struct Guard { qecmc_plan *p; ~Guard() { qecmc_plan_destroy(p); } } guard{pl};   // (synchronises: an error return may leave the launch running)
    if (N == 0) return 0;
    if (!init || !hist_out) return fail(QECMC_ERR_INVALID, "NULL buffer");
    const size_t nq = pl->args.nq, Nc = pl->args.Nc, ncls = pl->args.ncls, D = (size_t)droplets;
    const uint64_t M = N * ncls * D;                                  // ladders
    const uint64_t sets = per_rung ? M * Nc : N * ncls;               // PTRC: one per (ladder, rung); PTDC: one per (syndrome, class)
    if (M + p.first_syndrome > 0xFFFFFFFFull) return fail(QECMC_ERR_INVALID, "first_syndrome + N * classes * droplets = %llu ladders exceed the 32-bit syndrome index", (unsigned long long)(M + p.first_syndrome));
    uint64_t cap = 16;
    while (cap < 2 * p.steps * (per_rung ? 1 : Nc * D)) cap <<= 1;    // twice the insertions one set can see
    if (per_rung) conv_mult = 0.0;                                    // PTRC_droplet's stop is commented out (decoders.py:627-630)
    const bool own = conv_mult != 0.0 && D > 1;                       // the stop looks at each droplet's own dictionary
    uint64_t own_cap = 16;
    while (own_cap < 2 * p.steps * Nc) own_cap <<= 1;
    const uint64_t maxu = p.steps * Nc * D;                           // distinct chains a (syndrome, class) set can hold
    if (xyz_out && per_rung) return fail(QECMC_ERR_INVALID, "xyz_out is defined for the per-class sets, not with QECMC_PTDC_SET_PER_RUNG");
    if (xyz_out && nq > 1023) return fail(QECMC_ERR_UNSUPPORTED, "xyz_out packs counts in 10 bits: nq=%zu", nq);
    const uint64_t need = sets * cap * 8 + M * nq + sets * (nq + 1) * 8 + (own ? M * own_cap * 8 : 0) + M * 4 + (xyz_out ? sets * (maxu + 1) * 4 : 0);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (need > free_b - free_b / 8)
        return fail(QECMC_ERR_INVALID, "PTDC needs %llu bytes of device memory (%llu sets x %llu keys), %zu free: lower N, droplets or steps",
                    (unsigned long long)need, (unsigned long long)sets, (unsigned long long)cap, free_b);
    // one start per ladder; the kernel copies it into every rung and sets the top flag (Ladder.__init__, mcmc.py:72-75)
    std::vector<uint8_t> st((size_t)M * nq);
    for (uint64_t sc = 0; sc < N * ncls; ++sc)
        for (size_t d = 0; d < D; ++d)
            std::memcpy(&st[(sc * D + d) * nq], init + (init_per_droplet ? sc * D + d : sc) * nq, nq);
    DevBuf ds, dtab, dh, dm, down, dsd, dxyz, dxc, dthr;
    HIP_TRY(ds.alloc(st.size()));
    if (xyz_out) {
        HIP_TRY(dxyz.alloc(sets * maxu * 4)); HIP_TRY(hipMemset(dxyz.p, 0xFF, sets * maxu * 4));
        HIP_TRY(dxc.alloc(sets * 4)); HIP_TRY(hipMemset(dxc.p, 0, sets * 4));
    }
    if (!xyz_thr.empty()) { HIP_TRY(dthr.alloc(729 * 8)); HIP_TRY(hipMemcpy(dthr.p, xyz_thr.data(), 729 * 8, hipMemcpyHostToDevice)); }
    if (own) { HIP_TRY(down.alloc(M * own_cap * 8)); HIP_TRY(hipMemset(down.p, 0, M * own_cap * 8)); }
    if (steps_done_out) HIP_TRY(dsd.alloc(M * 4));
    HIP_TRY(dtab.alloc(sets * cap * 8)); HIP_TRY(dh.alloc(sets * (nq + 1) * 4));
    if (m_out) { HIP_TRY(dm.alloc(sets * (nq + 1) * 4)); HIP_TRY(hipMemset(dm.p, 0, sets * (nq + 1) * 4)); }
    HIP_TRY(hipMemcpy(ds.p, st.data(), st.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dtab.p, 0, sets * cap * 8)); HIP_TRY(hipMemset(dh.p, 0, sets * (nq + 1) * 4));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, 0));
    // ONE launch: the ladder kernel inserts every rung's configuration into its set after every step (USET instantiation)
    LadderArgs a = pl->args;
    a.init = ds.as<uint8_t>();
    a.N = M; a.first_syndrome = p.first_syndrome; a.nsteps = p.steps; a.step0 = 0; a.prop0 = 0; a.resume = 0; a.write_states = 0;
    a.uset_tab = reinterpret_cast<unsigned long long *>(dtab.p); a.uset_cap = cap; a.uset_hist = dh.as<uint32_t>();
    a.uset_mhist = m_out ? dm.as<uint32_t>() : nullptr; a.uset_D = (uint32_t)D; a.uset_per_rung = per_rung ? 1 : 0;
    a.uset_conv_mult = conv_mult; a.uset_own = own ? reinterpret_cast<unsigned long long *>(down.p) : nullptr; a.uset_own_cap = own_cap;
    a.steps_done = steps_done_out ? dsd.as<uint32_t>() : nullptr;
    a.uset_xyz = xyz_out ? dxyz.as<uint32_t>() : nullptr; a.uset_xyz_cnt = xyz_out ? dxc.as<uint32_t>() : nullptr; a.uset_xyz_stride = maxu;
    a.xyz_thr = xyz_thr.empty() ? nullptr : dthr.as<uint64_t>();
    {
        const hipError_t e = launch_ladder_rs_toric(a, 0);
        if (e != hipSuccess) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); return fail(QECMC_ERR_HIP, "PTDC launch: %s", hipGetErrorString(e)); }
    }
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    HIP_TRY(hipMemcpy(hist_out, dh.p, sets * (nq + 1) * 4, hipMemcpyDeviceToHost));
    if (m_out) HIP_TRY(hipMemcpy(m_out, dm.p, sets * (nq + 1) * 4, hipMemcpyDeviceToHost));
    if (steps_done_out) HIP_TRY(hipMemcpy(steps_done_out, dsd.p, M * 4, hipMemcpyDeviceToHost));
    if (xyz_out) HIP_TRY(hipMemcpy(xyz_out, dxyz.p, sets * maxu * 4, hipMemcpyDeviceToHost));
    if (xyz_out && xyz_count_out) HIP_TRY(hipMemcpy(xyz_count_out, dxc.p, sets * 4, hipMemcpyDeviceToHost));
    if (stats_out) {
        stats_out->proposals = M * Nc * p.iters * p.steps;
        stats_out->swap_tests = M * (Nc - 1) * p.steps;
        stats_out->kernel_ms = ms;
        stats_out->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return 0;
}

}  // extern "C"
