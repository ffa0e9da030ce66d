// Microbenchmark #3: VALU floor of the register-resident radix-2 stages used by the NTT kernels
// (no LDS, no global memory inside the loop).  1024-thread workgroups, one per CU (4 waves/SIMD)
// and 8 waves/SIMD for comparison.  Prints ns per wave-butterfly per SIMD and the implied
// time per 2^15-point limb (240 butterflies x 16 waves / 4 SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mkhe-kklss_amd/csrc/modarith.h"
using namespace mkhe;
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

// the butterflies of csrc/ntt_kernels.hip (signed-digit products; twiddles / q in signed-split form)
__device__ __forceinline__ void bf_nr(u64& U, u64& V, u64 w, u64 q, u64 q2, u32 ninv) {
    const i64 Tm = mont_mul_sd((i64)V, w, sd_split(q), ninv); const i64 u = (i64)U; U = (u64)(u + Tm); V = (u64)(u - Tm);
}
__device__ __forceinline__ void bf_cs(u64& U, u64& V, u64 w, u64 q, u64 q2, u32 ninv) {
    u64 Tm = mont_mul_sdu(V, w, sd_split(q), q, ninv); u64 u = csub(U, q2); U = u + Tm; V = u + (q2 - Tm);
}
template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS) k(u64* out, const u64* tw, u64 q, u32 ninv, int reps, unsigned long long* clk) {
    u64 x[32], w[16];
    const u64 q2 = 2 * q;
    for (int r = 0; r < 32; ++r) x[r] = (threadIdx.x * 977 + r * 131 + 7) % q;
    for (int k2 = 0; k2 < 16; ++k2) w[k2] = sd_split(tw[(threadIdx.x * 16 + k2) & 1023]);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int B = 4; B >= 0; --B) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1)), i1 = i0 | (1 << B);
                if (MODE == 1) bf_nr(x[i0], x[i1], w[g >> B], q, q2, ninv); else bf_cs(x[i0], x[i1], w[g >> B], q, q2, ninv);
            }
        }
        if (MODE == 1) { for (int r = 0; r < 32; ++r) x[r] = (u64)((i64)(x[r] << 8) >> 8); }   // keep |x| < 2^55 (cheap, not counted)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    u64 acc = 0; for (int r = 0; r < 32; ++r) acc += x[r];
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int MODE, int THREADS> int run(const char* name, int blocks) {
    u64 *out, *tw; unsigned long long* clk;
    CHECK(hipMalloc(&out, (size_t)blocks * THREADS * 8)); CHECK(hipMalloc(&tw, 1024 * 8)); CHECK(hipMalloc(&clk, blocks * 16));
    u64 h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (0x123456789abcdefull * (i + 1)) % 0x3fffffffd60001ull;
    CHECK(hipMemcpy(tw, h, sizeof(h), hipMemcpyHostToDevice));
    const u64 q = 0x3fffffffd60001ull; u64 qi = q; for (int i = 0; i < 6; ++i) qi *= 2 - q * qi;
    const int reps = 200;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<MODE, THREADS><<<blocks, THREADS>>>(out, tw, q, (u32)(0 - qi), reps, clk); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); k<MODE, THREADS><<<blocks, THREADS>>>(out, tw, q, (u32)(0 - qi), reps, clk); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long hc[8192]; CHECK(hipMemcpy(hc, clk, blocks * 16, hipMemcpyDeviceToHost));
    double ticks = 0, rt = 0; for (int i = 0; i < blocks; ++i) { ticks += hc[2 * i]; rt += hc[2 * i + 1]; }
    double ghz = ticks / rt / 10.0;                      // s_memrealtime = 100 MHz
    double waves_per_simd = (double)blocks * THREADS / 64 / (256.0 * 4.0);
    double bfly_per_wave = (double)reps * 80;
    double ns = ms * 1e6 / (waves_per_simd * bfly_per_wave);
    printf("%-34s %7.3f ms  clock %.2f GHz  %6.2f ns per wave-butterfly per SIMD (%5.1f cyc)  => %5.1f us per 2^15 limb per CU\n",
           name, ms, ghz, ns, ns * ghz, ns * 240 * 16 / 4 / 1000.0);
    return 0;
}
int main() {
    run<1, 1024>("no-csub, 4 waves/SIMD", 256);
    run<0, 1024>("csub,    4 waves/SIMD", 256);
    run<1, 256>("no-csub, 8 waves/SIMD (256thr x8)", 256 * 8);
    run<0, 256>("csub,    8 waves/SIMD (256thr x8)", 256 * 8);
    run<1, 1024>("no-csub, 4 waves/SIMD again", 256);
    return 0;
}
