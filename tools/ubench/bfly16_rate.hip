// Microbenchmark #4: issue rate of the block-asm signed-digit butterfly of csrc/ntt16_kernels.hip (16 registers per thread, scalar
// twiddles, no LDS / global memory in the loop): one butterfly at a time vs two interleaved chains, at 2 / 4 / 8 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o bfly16_rate bfly16_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mkhe-kklss_amd/csrc/modarith.h"
using namespace mkhe;
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
struct MC { i32 q0, q1; u32 ninv; };

__device__ __forceinline__ i64 mm1(i64 a, i32 w0, i32 w1, const MC& c) {
    const u32 al = lo32((u64)a); const i32 a0 = (i32)al; const i32 a1 = (i32)(hi32((u64)a) + (al >> 31));
    i64 acc = (i64)a0 * w0; i32 m, m2; u64 k;
    asm("v_mul_lo_u32 %1, %3, %7\n\tv_mad_i64_i32 %0, %2, %1, %8, %0\n\tv_ashrrev_i64 %0, 32, %0\n\tv_mad_i64_i32 %0, %2, %4, %6, %0\n\t"
        "v_mad_i64_i32 %0, %2, %1, %9, %0\n\tv_mad_i64_i32 %0, %2, %5, %10, %0"
        : "+v"(acc), "=&v"(m), "=&s"(k) : "v"(lo32((u64)acc)), "v"(a0), "v"(a1), "s"(w1), "s"(c.ninv), "s"(c.q0), "s"(c.q1), "s"(w0));
    asm("v_mul_lo_u32 %1, %3, %6\n\tv_mad_i64_i32 %0, %2, %1, %7, %0\n\tv_ashrrev_i64 %0, 32, %0\n\tv_mad_i64_i32 %0, %2, %4, %5, %0\n\t"
        "v_mad_i64_i32 %0, %2, %1, %8, %0"
        : "+v"(acc), "=&v"(m2), "=&s"(k) : "v"(lo32((u64)acc)), "v"(a1), "s"(w1), "s"(c.ninv), "s"(c.q0), "s"(c.q1));
    return acc;
}
// two independent products, chains interleaved instruction by instruction
__device__ __forceinline__ void mm2(i64 a, i64 b, i32 w0, i32 w1, i32 v0, i32 v1, const MC& c, i64& ra, i64& rb) {
    const u32 al = lo32((u64)a); const i32 a0 = (i32)al; const i32 a1 = (i32)(hi32((u64)a) + (al >> 31));
    const u32 bl = lo32((u64)b); const i32 b0 = (i32)bl; const i32 b1 = (i32)(hi32((u64)b) + (bl >> 31));
    i64 x = (i64)a0 * w0, y = (i64)b0 * v0; i32 m, n; u64 k;
    asm("v_mul_lo_u32 %2, %5, %13\n\tv_mul_lo_u32 %3, %6, %13\n\t"
        "v_mad_i64_i32 %0, %4, %2, %14, %0\n\tv_mad_i64_i32 %1, %4, %3, %14, %1\n\t"
        "v_ashrrev_i64 %0, 32, %0\n\tv_ashrrev_i64 %1, 32, %1\n\t"
        "v_mad_i64_i32 %0, %4, %7, %11, %0\n\tv_mad_i64_i32 %1, %4, %9, %12, %1\n\t"
        "v_mad_i64_i32 %0, %4, %2, %15, %0\n\tv_mad_i64_i32 %1, %4, %3, %15, %1\n\t"
        "v_mad_i64_i32 %0, %4, %8, %16, %0\n\tv_mad_i64_i32 %1, %4, %10, %17, %1"
        : "+v"(x), "+v"(y), "=&v"(m), "=&v"(n), "=&s"(k)
        : "v"(lo32((u64)x)), "v"(lo32((u64)y)), "v"(a0), "v"(a1), "v"(b0), "v"(b1), "s"(w1), "s"(v1), "s"(c.ninv), "s"(c.q0), "s"(c.q1), "s"(w0), "s"(v0));
    asm("v_mul_lo_u32 %2, %5, %11\n\tv_mul_lo_u32 %3, %6, %11\n\t"
        "v_mad_i64_i32 %0, %4, %2, %12, %0\n\tv_mad_i64_i32 %1, %4, %3, %12, %1\n\t"
        "v_ashrrev_i64 %0, 32, %0\n\tv_ashrrev_i64 %1, 32, %1\n\t"
        "v_mad_i64_i32 %0, %4, %7, %9, %0\n\tv_mad_i64_i32 %1, %4, %8, %10, %1\n\t"
        "v_mad_i64_i32 %0, %4, %2, %13, %0\n\tv_mad_i64_i32 %1, %4, %3, %13, %1"
        : "+v"(x), "+v"(y), "=&v"(m), "=&v"(n), "=&s"(k)
        : "v"(lo32((u64)x)), "v"(lo32((u64)y)), "v"(a1), "v"(b1), "s"(w1), "s"(v1), "s"(c.ninv), "s"(c.q0), "s"(c.q1));
    ra = x; rb = y;
}
template <int IL, int THREADS, int WPE>
__global__ void __launch_bounds__(THREADS, WPE) k(u64* out, const u64* tw, u64 qs, u32 ninv, int reps, unsigned long long* clk) {
    u64 x[16];
    MC c; c.q0 = (i32)lo32(qs); c.q1 = (i32)hi32(qs); c.ninv = ninv;
    asm("" : "+s"(c.q0), "+s"(c.q1), "+s"(c.ninv));
    for (int r = 0; r < 16; ++r) x[r] = (threadIdx.x * 977 + r * 131 + 7);
    i32 w0[8], w1[8];
    for (int i = 0; i < 8; ++i) { u64 w = ((const __attribute__((address_space(4))) u64*)tw)[i]; w0[i] = (i32)lo32(w); w1[i] = (i32)hi32(w); asm("" : "+s"(w0[i]), "+s"(w1[i])); }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int B = 3; B >= 0; --B) {
#pragma unroll
            for (int g = 0; g < 8; g += IL) {
                const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1)), i1 = i0 | (1 << B);
                if (IL == 1) {
                    const i64 T = mm1((i64)x[i1], w0[g >> B], w1[g >> B], c); const i64 u = (i64)x[i0];
                    x[i0] = (u64)(u + T); x[i1] = (u64)(u - T);
                } else {
                    const int g2 = g + 1, j0 = ((g2 >> B) << (B + 1)) | (g2 & ((1 << B) - 1)), j1 = j0 | (1 << B);
                    i64 T, S; mm2((i64)x[i1], (i64)x[j1], w0[g >> B], w1[g >> B], w0[g2 >> B], w1[g2 >> B], c, T, S);
                    const i64 u = (i64)x[i0], v = (i64)x[j0];
                    x[i0] = (u64)(u + T); x[i1] = (u64)(u - T); x[j0] = (u64)(v + S); x[j1] = (u64)(v - S);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int r = 0; r < 16; ++r) x[r] = (u64)((i64)(x[r] << 8) >> 8);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    u64 acc = 0; for (int r = 0; r < 16; ++r) acc += x[r];
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int IL, int THREADS, int WPE> int run(const char* name, int blocks) {
    u64 *out, *tw; unsigned long long* clk;
    CHECK(hipMalloc(&out, (size_t)blocks * THREADS * 8)); CHECK(hipMalloc(&tw, 1024 * 8)); CHECK(hipMalloc(&clk, blocks * 16));
    const u64 q = 0x3fffffffd60001ull; u64 qi = q; for (int i = 0; i < 6; ++i) qi *= 2 - q * qi;
    u64 h[1024]; for (int i = 0; i < 1024; ++i) h[i] = sd_split((0x123456789abcdefull * (i + 1)) % q);
    CHECK(hipMemcpy(tw, h, sizeof(h), hipMemcpyHostToDevice));
    const int reps = 400;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<IL, THREADS, WPE><<<blocks, THREADS>>>(out, tw, sd_split(q), (u32)(0 - qi), reps, clk); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); k<IL, THREADS, WPE><<<blocks, THREADS>>>(out, tw, sd_split(q), (u32)(0 - qi), reps, clk); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long hc[16384]; CHECK(hipMemcpy(hc, clk, blocks * 16, hipMemcpyDeviceToHost));
    double ticks = 0, rt = 0; for (int i = 0; i < blocks; ++i) { ticks += hc[2 * i]; rt += hc[2 * i + 1]; }
    double ghz = ticks / rt / 10.0;
    double waves_per_simd = (double)blocks * THREADS / 64 / (256.0 * 4.0);
    double ns = ms * 1e6 / (waves_per_simd * reps * 32.0);
    printf("%-44s %7.3f ms  clock %.2f GHz  %6.2f ns per wave-butterfly per SIMD (%5.1f cyc)  => %5.1f us per 2^15 limb per CU\n",
           name, ms, ghz, ns, ns * ghz, ns * 240 * 16 / 4 / 1000.0);
    return 0;
}
int main() {
    run<1, 1024, 8>("1 chain,  8 waves/SIMD", 512);
    run<2, 1024, 8>("2 chains, 8 waves/SIMD", 512);
    run<1, 1024, 8>("1 chain,  4 waves/SIMD", 256);
    run<2, 1024, 8>("2 chains, 4 waves/SIMD", 256);
    run<1, 512, 8>("1 chain,  2 waves/SIMD", 256);
    run<2, 512, 8>("2 chains, 2 waves/SIMD", 256);
    run<1, 256, 8>("1 chain,  1 wave/SIMD", 256);
    run<2, 256, 8>("2 chains, 1 wave/SIMD", 256);
    return 0;
}
