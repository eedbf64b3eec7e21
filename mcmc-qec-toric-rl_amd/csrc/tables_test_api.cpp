// C test API over the host-side table builders (tables.hpp), compiled ALONE by g++ -- no HIP runtime, no device -- with
// -fsanitize=address,undefined (make tables_asan) or plainly (make tables): tests/test_host_tables.py checks every table the
// plan precomputes for the kernels against values the CPU oracle computes.  Not part of libqecmc.so.
#include "tables.hpp"

#include <cstring>

using namespace qecmc;
using namespace qecmc::tables;

namespace {
template <class T> int put(const std::vector<T> &v, T *out, int cap)
{
    if ((int)v.size() > cap) return -(int)v.size();
    std::memcpy(out, v.data(), v.size() * sizeof(T));
    return (int)v.size();
}
std::vector<uint32_t> gen_table(int code, int L) { return code == QECMC_TORIC ? toric_generator_table(L) : surf_generator_table(code, L); }
}  // namespace

extern "C" {
int qt_generator_table(int code, int L, uint32_t *out, int cap) { return put(gen_table(code, L), out, cap); }
int qt_logical_masks(int code, int L, int W, uint32_t *out, int cap)
{
    return put(code == QECMC_TORIC ? toric_logical_masks(L, W) : surf_logical_masks(code, L, W), out, cap);
}
int qt_wave_descriptors(int code, int L, uint32_t *out, int cap) { return put(wave_descriptors(gen_table(code, L)), out, cap); }
uint64_t qt_thr64(double v) { return thr64(v); }
uint64_t qt_thr44(double v) { return thr44(v); }
uint32_t qt_thr32(double v) { return thr32(v); }
double qt_chain_factor(double p) { return chain_factor(p); }
int qt_ladder(double p_bottom, double p_top, int Nc, double *pl, double *pd)
{
    std::vector<double> a, b;
    ladder_probabilities(p_bottom, p_top, Nc, a, b);
    std::memcpy(pl, a.data(), a.size() * sizeof(double));
    if (!b.empty()) std::memcpy(pd, b.data(), b.size() * sizeof(double));
    return (int)b.size();
}
int qt_bias_tables(int alpha_model, double p, double eta_or_alpha, int nq, double *out, int cap)
{
    return put(alpha_model ? alpha_tables(p, eta_or_alpha, (size_t)nq) : bias_tables(p, eta_or_alpha, (size_t)nq), out, cap);
}
int qt_patterns(int code, int L, uint8_t *gen_type, int cap_types, uint32_t *patterns, int cap_patterns)
{
    std::vector<uint8_t> gt;
    std::vector<uint32_t> pat;
    generator_patterns(gen_table(code, L), gt, pat);
    if (put(gt, gen_type, cap_types) < 0) return -1;
    return put(pat, patterns, cap_patterns);
}
int qt_count_change(const uint32_t *patterns, int n, uint32_t *out, int cap)
{
    return put(count_change_table(std::vector<uint32_t>(patterns, patterns + n)), out, cap);
}
int qt_swap_thresholds(const double *pdiff, int n, int nq, uint64_t *out, int cap)
{
    return put(swap_thresholds(std::vector<double>(pdiff, pdiff + n), nq), out, cap);
}
int qt_colour_phases(int code, int L, uint16_t *out, int cap)
{
    int n_phases = 0;
    const std::vector<uint16_t> ph = colour_phases(gen_table(code, L), n_phases);
    return put(ph, out, cap) < 0 ? -1 : n_phases;
}
}
