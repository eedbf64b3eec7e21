// Micro-benchmark: LDS-array cost of the operations the proposal loop is made of (gfx950): lane-private ds_read_b32,
// ds_write_b32 and ds_xor_b32 (no return), and random 16-byte table reads (ds_read_b128).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(512) void k(uint32_t *out, int iters, uint32_t seed)
{
    __shared__ uint32_t buf[8 * 16 * 64];      // [wave][16 words][64 lanes]
    __shared__ uint4 tab[256];
    __shared__ uint2 tab2[256];
    __shared__ uint32_t tab1[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8 * 16 * 64; i += blockDim.x) buf[i] = i * 2654435761u;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) { tab[i] = make_uint4(i, i + 1, i + 2, i + 3); tab2[i] = make_uint2(i, i + 5); tab1[i] = i * 7; }
    __syncthreads();
    uint32_t *p = buf + wave * 16 * 64 + lane;
    uint32_t acc = seed, r = seed * 747796405u + lane * 2891336453u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) acc += ((volatile uint32_t *)p)[u * 64];                           // ds_read_b32
            else if (KIND == 1) ((volatile uint32_t *)p)[u * 64] = acc + u;                   // ds_write_b32
            else if (KIND == 5) { r = r * 1664525u + 1013904223u; const uint2 e = tab2[(r >> 24) % 162u]; acc += e.x ^ e.y; }    // random ds_read_b64
            else if (KIND == 7) { r = r * 1664525u + 1013904223u; const uint32_t g = (r >> 24) % 162u; const uint2 e0 = tab2[g], e1 = tab2[g + 81]; acc += e0.x ^ e1.y; }   // two 8-byte tables: ds_read2_b64 offset1:81
            else if (KIND == 8) { r = r * 1664525u + 1013904223u; const uint32_t g = (r >> 24) % 81u; const uint2 e0 = tab2[2 * g], e1 = tab2[2 * g + 1]; acc += e0.x ^ e1.y; }   // adjacent pair: ds_read2_b64 offset1:1
            else if (KIND == 6) { r = r * 1664525u + 1013904223u; acc += tab1[(r >> 24) % 162u]; }                               // random ds_read_b32
            else if (KIND == 2) __hip_atomic_fetch_xor(p + u * 64, acc | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_xor_b32
            else if (KIND == 3) { r = r * 1664525u + 1013904223u; const uint4 e = tab[(r >> 24) % 162u]; acc += e.x ^ e.w; }     // random ds_read_b128
            else if (KIND == 4) { r = r * 1664525u + 1013904223u; acc += r >> 24; }           // the index arithmetic alone
        }
        if (KIND == 2) acc += i;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + p[0];
}

template <int KIND>
int run(const char *name)
{
    uint32_t *out;
    const int blocks = 256 * 4, threads = 512, iters = 2000;     // 4 workgroups of 8 waves per CU: 8 waves per SIMD
    CHECK(hipMalloc(&out, blocks * threads * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1u);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)blocks * threads / 64, insts = waves * iters * 16.0;
    // LDS cycles per wave-instruction per CU, assuming 256 CUs at 2.4 GHz
    printf("%-24s %8.3f ms  %.2f cycles per wave-instruction per CU\n", name, ms, ms * 1e-3 * 2.4e9 * 256 / insts);
    CHECK(hipFree(out));
    return 0;
}

int main()
{
    run<0>("ds_read_b32 lane-private");
    run<1>("ds_write_b32 lane-private");
    run<2>("ds_xor_b32 lane-private");
    run<3>("ds_read_b128 random/162");
    run<5>("ds_read_b64 random/162");
    run<6>("ds_read_b32 random/162");
    run<7>("ds_read2_b64 split tables");
    run<8>("ds_read2_b64 adjacent");
    run<4>("(index arithmetic only)");
    return 0;
}
