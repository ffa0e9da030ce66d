// Microbenchmark #5: the butterfly with the ONE-round product of round 2 (csrc/ntt16_kernels.hip mm31) against the two-round signed-digit
// Montgomery product (mm1 of bfly16_rate.hip).  The twiddle w enters as two pre-reduced constants u = w * 2^31 mod q and
// u' = w * 2^63 mod q (balanced, radix-2^31 digits): a * w = a0 * u + a1 * u' (92 bits) needs ONE Montgomery round of radix 2^31 where
// a 128-bit product needs two: 8 multiplier-class + 1 plain instruction instead of 12.
// hipcc --offload-arch=gfx950 -O3 -o bfly31_rate bfly31_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mkhe-kklss_amd/csrc/modarith.h"
using namespace mkhe;
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
struct MC { i32 q0, q1; u32 ninv; };

__device__ __forceinline__ i64 mm_old(i64 a, i32 w0, i32 w1, const MC& c) {
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
// one round: u0, u1 / v0, v1 = radix-2^31 digits of u = w 2^31 mod q and u' = w 2^63 mod q; q0, q1 = radix-2^31 digits of q
__device__ __forceinline__ i64 mm_new(i64 a, i32 u0, i32 u1, i32 v0, i32 v1, const MC& c) {
    const u32 al = lo32((u64)a); const i32 a0 = (i32)al; const i32 a1 = (i32)(hi32((u64)a) + (al >> 31));
    i32 s0 = u0, t0 = v0;
    asm("" : "+s"(s0), "+s"(t0));
    i64 acc = (i64)a0 * s0;                                // v_mad_i64_i32
    acc = (i64)a1 * t0 + acc;                              // + a1 * u'0
    i32 m; u64 k;
    asm("v_mul_lo_u32 %1, %3, %7\n\t"                     // lo(acc) * -q^-1
        "v_bfe_i32 %1, %1, 0, 31\n\t"                     // balanced 31-bit digit
        "v_mad_i64_i32 %0, %2, %1, %8, %0\n\t"            // + m * q0: low 31 bits zero
        "v_ashrrev_i64 %0, 31, %0\n\t"
        "v_mad_i64_i32 %0, %2, %4, %6, %0\n\t"            // + a0 * u1
        "v_mad_i64_i32 %0, %2, %5, %10, %0\n\t"           // + a1 * u'1
        "v_mad_i64_i32 %0, %2, %1, %9, %0"                  // + m * q1
        : "+v"(acc), "=&v"(m), "=&s"(k)
        : "v"(lo32((u64)acc)), "v"(a0), "v"(a1), "s"(u1), "s"(c.ninv), "s"(c.q0), "s"(c.q1), "s"(v1));
    return acc;
}
template <int NEW, int THREADS, int WPE>
__global__ void __launch_bounds__(THREADS, WPE) k(u64* out, const u64* tw, u64 qs, u32 ninv, int reps, unsigned long long* clk) {
    u64 x[16];
    MC c; c.q0 = (i32)lo32(qs); c.q1 = (i32)hi32(qs); c.ninv = ninv;
    asm("" : "+s"(c.q0), "+s"(c.q1), "+s"(c.ninv));
    for (int r = 0; r < 16; ++r) x[r] = (threadIdx.x * 977 + r * 131 + 7);
    i32 w0[8], w1[8], v0[8], v1[8];
    for (int i = 0; i < 8; ++i) {
        u64 w = ((const __attribute__((address_space(4))) u64*)tw)[2 * i], v = ((const __attribute__((address_space(4))) u64*)tw)[2 * i + 1];
        w0[i] = (i32)lo32(w); w1[i] = (i32)hi32(w); v0[i] = (i32)lo32(v); v1[i] = (i32)hi32(v);
        asm("" : "+s"(w0[i]), "+s"(w1[i]), "+s"(v0[i]), "+s"(v1[i]));
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int B = 3; B >= 0; --B) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1)), i1 = i0 | (1 << B);
                const i64 T = NEW ? mm_new((i64)x[i1], w0[g >> B], w1[g >> B], v0[g >> B], v1[g >> B], c) : mm_old((i64)x[i1], w0[g >> B], w1[g >> B], c);
                const i64 u = (i64)x[i0];
                x[i0] = (u64)(u + T); x[i1] = (u64)(u - T);
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
template <int NEW, int THREADS, int WPE> int run(const char* name, int blocks) {
    u64 *out, *tw; unsigned long long* clk;
    CHECK(hipMalloc(&out, (size_t)blocks * THREADS * 8)); CHECK(hipMalloc(&tw, 1024 * 8)); CHECK(hipMalloc(&clk, blocks * 16));
    const u64 q = 0x3fffffffd60001ull; u64 qi = q; for (int i = 0; i < 6; ++i) qi *= 2 - q * qi;
    u64 h[1024]; for (int i = 0; i < 1024; ++i) h[i] = sd_split((0x123456789abcdefull * (i + 1)) % q);
    CHECK(hipMemcpy(tw, h, sizeof(h), hipMemcpyHostToDevice));
    const int reps = 400;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<NEW, THREADS, WPE><<<blocks, THREADS>>>(out, tw, sd_split(q), (u32)(0 - qi), reps, clk); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); k<NEW, THREADS, WPE><<<blocks, THREADS>>>(out, tw, sd_split(q), (u32)(0 - qi), reps, clk); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
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
    run<0, 1024, 8>("two rounds (12 + 2), 8 waves/SIMD", 512);
    run<1, 1024, 8>("one round  (8 + 3),  8 waves/SIMD", 512);
    run<0, 1024, 8>("two rounds (12 + 2), 4 waves/SIMD", 256);
    run<1, 1024, 8>("one round  (8 + 3),  4 waves/SIMD", 256);
    return 0;
}
