// Micro-benchmark: issue cost of the integer instructions Philox4x32 is made of (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a ^ 0x1234567, d = b + 77;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) {          // mad_u64_u32 (4 independent chains)
                uint64_t p0 = (uint64_t)a * 0xD2511F53u, p1 = (uint64_t)b * 0xCD9E8D57u;
                uint64_t p2 = (uint64_t)c * 0xD2511F53u, p3 = (uint64_t)d * 0xCD9E8D57u;
                a = (uint32_t)(p0 >> 32) ^ (uint32_t)p0; b = (uint32_t)(p1 >> 32) ^ (uint32_t)p1;
                c = (uint32_t)(p2 >> 32) ^ (uint32_t)p2; d = (uint32_t)(p3 >> 32) ^ (uint32_t)p3;
            } else if (KIND == 1) {   // xor/add only (same count of simple ops: 8)
                a = (a ^ b) + 1; b = (b ^ c) + 2; c = (c ^ d) + 3; d = (d ^ a) + 4;
            } else if (KIND == 2) {   // mul_lo only
                a = a * 0xD2511F53u + 1; b = b * 0xCD9E8D57u + 1; c = c * 0xD2511F53u + 1; d = d * 0xCD9E8D57u + 1;
            } else if (KIND == 3) {   // mul_hi only
                a = __umulhi(a, 0xD2511F53u) + c; b = __umulhi(b, 0xCD9E8D57u) + d; c = __umulhi(c, 0xD2511F53u) + a; d = __umulhi(d, 0xCD9E8D57u) + b;
            } else if (KIND == 4) {   // 24-bit mul
                a = __umul24(a, 0x511F53u) + c; b = __umul24(b, 0x9E8D57u) + d; c = __umul24(c, 0x511F53u) + a; d = __umul24(d, 0x9E8D57u) + b;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int KIND>
int run(const char *name, int per_iter_ops)
{
    uint32_t *out;
    const int blocks = 256 * 8, threads = 256, iters = 2000;
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
    const double waves = (double)blocks * threads / 64, insts = waves * iters * 16.0 * per_iter_ops;
    // cycles per wave-instruction per SIMD, assuming 1024 SIMDs at 2.4 GHz
    printf("%-14s %8.3f ms  %.2f cycles/wave-inst/SIMD (@2.4GHz, %d insts per unrolled step)\n", name, ms,
           ms * 1e-3 * 2.4e9 * 1024 / insts, per_iter_ops);
    CHECK(hipFree(out));
    return 0;
}

int main()
{
    run<0>("mad_u64_u32+xor", 8);   // 4 mad + 4 xor
    run<1>("xor+add", 8);
    run<2>("mul_lo+add", 4);        // v_mad_u32_u24? compiler may fuse; see ISA
    run<3>("mul_hi+add", 8);
    run<4>("mul24+add", 4);
    return 0;
}
