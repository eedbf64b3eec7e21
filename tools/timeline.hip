// Diagnostic (not part of the product): per-workgroup start/end stamps + CU placement of the
// ladder kernel, to see how evenly one launch fills the chip.  Unity build of the library
// sources with QECMC_TIMELINE defined.
#define QECMC_TIMELINE 1
#include "../mcmc-qec-toric-rl_amd/csrc/capi.hip"
#include "../mcmc-qec-toric-rl_amd/csrc/ladder_rs.hip"
#include "../mcmc-qec-toric-rl_amd/csrc/ladder_toric.hip"
#include "../mcmc-qec-toric-rl_amd/csrc/primitives.hip"
namespace qecmc {   // the families this tool does not launch
hipError_t launch_ladder_surf(const LadderArgs &, hipStream_t) { return hipErrorInvalidValue; }
hipError_t launch_ladder_sweep(const LadderArgs &, hipStream_t) { return hipErrorInvalidValue; }
hipError_t launch_ladder_biased(const LadderArgs &, hipStream_t) { return hipErrorInvalidValue; }
hipError_t launch_ladder_uset(const LadderArgs &, hipStream_t) { return hipErrorInvalidValue; }
hipError_t launch_ladder_colour(const LadderArgs &, hipStream_t) { return hipErrorInvalidValue; }
}
#include <map>
#include <algorithm>

int main(int argc, char **argv)
{
    const uint64_t N = argc > 1 ? atoll(argv[1]) : 65536;
    const int steps = argc > 2 ? atoi(argv[2]) : 500;
    qecmc_params p; memset(&p, 0, sizeof p);
    p.abi_size = sizeof p; p.code = QECMC_TORIC; p.L = 9; p.Nc = 8; p.p = 0.15; p.p_logical = 0.5; p.iters = 10;
    p.steps = steps; p.tops_burn = 2; p.seed = 1;
    qecmc_plan *pl = nullptr;
    if (qecmc_plan_create(&p, &pl)) { printf("plan: %s\n", qecmc_last_error()); return 1; }
    const size_t nq = 162;
    std::vector<uint8_t> init(N * nq);
    uint32_t s = 12345;
    for (auto &v : init) { s = s * 1664525u + 1013904223u; v = (s >> 24) < 38 ? 1 + (s >> 8) % 3 : 0; }
    uint8_t *di; uint32_t *dc, *ds, *dt; uint64_t *dbg;
    const unsigned grid = (unsigned)((N + 63) / 64);
    hipMalloc(&di, N * nq); hipMalloc(&dc, N * 64); hipMalloc(&ds, N * 4); hipMalloc(&dt, N * 4); hipMalloc(&dbg, grid * 32);
    hipMemcpy(di, init.data(), N * nq, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        LadderArgs a = pl->args;
        a.init = di; a.counts = dc; a.samples = ds; a.tops0 = dt; a.states = nullptr; a.write_states = 0; a.N = N;
        a.nsteps = steps; a.dbg = dbg;
        hipMemset(dbg, 0, grid * 32);
        launch_ladder_rs_toric(a, 0);
        hipDeviceSynchronize();
    }
    std::vector<uint64_t> h(grid * 4);
    hipMemcpy(h.data(), dbg, grid * 32, hipMemcpyDeviceToHost);
    uint64_t t0 = ~0ull, t1 = 0;
    for (unsigned b = 0; b < grid; ++b) { t0 = std::min(t0, h[b * 4]); t1 = std::max(t1, h[b * 4 + 2]); }
    printf("grid %u kernel span %.3f ms (100 MHz ticks)\n", grid, (t1 - t0) / 1e5);
    std::map<uint32_t, std::vector<unsigned>> percu;
    double dur_sum = 0, dur_min = 1e30, dur_max = 0;
    std::vector<double> starts, ends, durs;
    for (unsigned b = 0; b < grid; ++b) {
        const uint32_t hw = (uint32_t)h[b * 4 + 1], xcc = (uint32_t)(h[b * 4 + 1] >> 32) & 0xF;
        const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        percu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
        const double d = (h[b * 4 + 2] - h[b * 4]) / 1e5;
        dur_sum += d; dur_min = std::min(dur_min, d); dur_max = std::max(dur_max, d);
        starts.push_back((h[b * 4] - t0) / 1e5); ends.push_back((h[b * 4 + 2] - t0) / 1e5); durs.push_back(d);
    }
    std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end()); std::sort(durs.begin(), durs.end());
    printf("block duration ms: min %.3f mean %.3f max %.3f; p10 %.3f p50 %.3f p90 %.3f\n", dur_min, dur_sum / grid, dur_max,
           durs[grid / 10], durs[grid / 2], durs[grid * 9 / 10]);
    printf("start ms: p50 %.3f p90 %.3f max %.3f; end ms: p10 %.3f p50 %.3f p90 %.3f max %.3f\n", starts[grid / 2],
           starts[grid * 9 / 10], starts[grid - 1], ends[grid / 10], ends[grid / 2], ends[grid * 9 / 10], ends[grid - 1]);
    std::map<size_t, int> hist;
    for (auto &kv : percu) hist[kv.second.size()]++;
    printf("distinct (xcc,se,sh,cu) ids: %zu; blocks-per-id histogram:", percu.size());
    for (auto &kv : hist) printf(" %zu:%d", kv.first, kv.second);
    printf("\n");
    return 0;
}
