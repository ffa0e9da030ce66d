// Microbenchmark #6 (round 3): the butterfly on the one-round product with an UNSIGNED low data digit ("mm30u") against mm31 of round 2.
// a = hi 2^32 + lo exactly (hi signed, lo unsigned): no digit fix-up (v_lshrrev + v_add) in front of the product.  The twiddle enters as
//   u = w 2^30 mod q in [0, q)  (radix-2^30 digits u0, u1 >= 0: they meet lo in v_mad_u64_u32) and
//   v = w 2^62 mod q, balanced  (radix-2^30 digits, signed: they meet hi in v_mad_i64_i32),
// one Montgomery round of radix 2^30: C = lo u0 + hi v0; m = bal30(lo(C) * -q^-1); T = ((C + m p0) >> 30) + lo u1 + hi v1 + m p1 = a w mod q,
// T in (-q, 5q) for |a| < 2^62: the never-reduced values grow by up to 5q per stage, which the 54-bit primes (2^62.9 / q = 477) can afford.
// Also checks mm30u against a host computation.   hipcc --offload-arch=gfx950 -O3 -o bfly30u_rate bfly30u_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mkhe-kklss_amd/csrc/modarith.h"
using namespace mkhe;
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
struct MC { i32 q0, q1; u32 ninv; };

__device__ __forceinline__ i64 mm31(i64 a, i32 u0, i32 u1, i32 v0, i32 v1, const MC& c) {
    const u32 al = lo32((u64)a); const i32 a0 = (i32)al; const i32 a1 = (i32)(hi32((u64)a) + (al >> 31));
    i32 s0 = u0, t0 = v0;
    asm("" : "+s"(s0), "+s"(t0));
    i64 acc = (i64)a0 * s0;
    acc = (i64)a1 * t0 + acc;
    i32 m; u64 k;
    asm("v_mul_lo_u32 %1, %3, %7\n\tv_bfe_i32 %1, %1, 0, 31\n\tv_mad_i64_i32 %0, %2, %1, %8, %0\n\tv_ashrrev_i64 %0, 31, %0\n\t"
        "v_mad_i64_i32 %0, %2, %4, %6, %0\n\tv_mad_i64_i32 %0, %2, %5, %10, %0\n\tv_mad_i64_i32 %0, %2, %1, %9, %0"
        : "+v"(acc), "=&v"(m), "=&s"(k)
        : "v"(lo32((u64)acc)), "v"(a0), "v"(a1), "s"(u1), "s"(c.ninv), "s"(c.q0), "s"(c.q1), "s"(v1));
    return acc;
}
// one asm block: 6 mads + mul_lo + ashr (8 multiplier-class) + bfe
__device__ __forceinline__ i64 mm30u(i64 a, i32 u0, i32 u1, i32 v0, i32 v1, const MC& c) {
    i64 acc; i32 m; u64 k;
    // (the chain is cut after the low column: an asm block can only name whole operands, and the multiply by -q^-1 reads the low word)
    asm("v_mad_u64_u32 %0, %1, %2, %4, 0\n\t"              // lo * u0
        "v_mad_i64_i32 %0, %1, %3, %5, %0"                   // + hi * v0
        : "=&v"(acc), "=&s"(k) : "v"(lo32((u64)a)), "v"(hi32((u64)a)), "s"(u0), "s"(v0));
    asm("v_mul_lo_u32 %1, %3, %8\n\t"                      // lo(C) * -q^-1
        "v_bfe_i32 %1, %1, 0, 30\n\t"                      // balanced 30-bit digit
        "v_mad_i64_i32 %0, %2, %1, %9, %0\n\t"             // + m * p0: low 30 bits zero
        "v_ashrrev_i64 %0, 30, %0\n\t"
        "v_mad_u64_u32 %0, %2, %4, %6, %0\n\t"             // + lo * u1
        "v_mad_i64_i32 %0, %2, %5, %7, %0\n\t"             // + hi * v1
        "v_mad_i64_i32 %0, %2, %1, %10, %0"                  // + m * p1
        : "+v"(acc), "=&v"(m), "=&s"(k)
        : "v"(lo32((u64)acc)), "v"(lo32((u64)a)), "v"(hi32((u64)a)), "s"(u1), "s"(v1), "s"(c.ninv), "s"(c.q0), "s"(c.q1));
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
                const i64 T = NEW ? mm30u((i64)x[i1], w0[g >> B], w1[g >> B], v0[g >> B], v1[g >> B], c) : mm31((i64)x[i1], w0[g >> B], w1[g >> B], v0[g >> B], v1[g >> B], c);
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
// correctness: T = mm30u(a, w) for random a (|a| < 2^62) and w: T = a w (mod q) and -q < T < 5q
__global__ void chk(const i64* a, const u64* uv, u64 qs30, u32 ninv30, i64* T, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    MC c; c.q0 = (i32)lo32(qs30); c.q1 = (i32)hi32(qs30); c.ninv = ninv30;
    const u64 u = ((const __attribute__((address_space(4))) u64*)uv)[0], v = ((const __attribute__((address_space(4))) u64*)uv)[1];
    T[i] = mm30u(a[i], (i32)lo32(u), (i32)hi32(u), (i32)lo32(v), (i32)hi32(v), c);
}
static u64 mulmod_h(u64 a, u64 b, u64 q) { return (u64)((unsigned __int128)a * b % q); }
// radix-2^30 digit pair packed as (hi digit << 32) | (u32)lo digit; balanced: lo digit in [-2^29, 2^29)
static u64 split30(i64 x, bool balanced) {
    i64 lo = x & ((1ll << 30) - 1);
    if (balanced && lo >= (1ll << 29)) lo -= 1ll << 30;
    i64 hi = (x - lo) >> 30;
    return ((u64)(u32)(i32)hi << 32) | (u32)(i32)lo;
}
int check() {
    const u64 q = 0x3fffffffd60001ull;
    const int n = 1 << 16;
    static i64 ha[1 << 16], hT[1 << 16];
    u64 s = 12345;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < n; ++i) { ha[i] = (i64)(rnd() >> 2) * ((rnd() & 1) ? 1 : -1); if (i < 8) ha[i] = (i & 1 ? -1 : 1) * (((i64)1 << 62) - 1 - i); if (i >= 8 && i < 16) ha[i] = (i64)0xffffffffull << (i & 7); }
    const u64 w = rnd() % q;
    u64 r30 = (1ull << 30) % q, r62 = (1ull << 62) % q;
    u64 u = mulmod_h(w, r30, q), vv = mulmod_h(w, r62, q);
    i64 vb = vv > q / 2 ? (i64)vv - (i64)q : (i64)vv;
    u64 huv[2] = {split30((i64)u, false), split30(vb, true)};
    u64 qinv = q; for (int i = 0; i < 6; ++i) qinv *= 2 - q * qinv;
    const u32 ninv30 = (u32)(0 - qinv) & ((1u << 30) - 1);
    i64* da; i64* dT; u64* duv;
    CHECK(hipMalloc(&da, n * 8)); CHECK(hipMalloc(&dT, n * 8)); CHECK(hipMalloc(&duv, 16));
    CHECK(hipMemcpy(da, ha, n * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(duv, huv, 16, hipMemcpyHostToDevice));
    chk<<<n / 256, 256>>>(da, duv, split30((i64)q, true), ninv30, dT, n);
    CHECK(hipMemcpy(hT, dT, n * 8, hipMemcpyDeviceToHost));
    int bad = 0; double lo = 0, hi = 0;
    for (int i = 0; i < n; ++i) {
        __int128 e = ((__int128)ha[i] * w) % (__int128)q; if (e < 0) e += q;
        __int128 g = (__int128)hT[i] % (__int128)q; if (g < 0) g += q;
        double rel = (double)hT[i] / (double)q;
        if (rel < lo) lo = rel; if (rel > hi) hi = rel;
        if (e != g || rel <= -1.0 || rel >= 5.0) { if (bad < 5) printf("MISMATCH i=%d a=%lld T=%lld\n", i, (long long)ha[i], (long long)hT[i]); ++bad; }
    }
    printf("mm30u check: %d values, %d bad, T/q in [%.3f, %.3f]\n", n, bad, lo, hi);
    return bad;
}
template <int NEW, int THREADS, int WPE> int run(const char* name, int blocks) {
    u64 *out, *tw; unsigned long long* clk;
    CHECK(hipMalloc(&out, (size_t)blocks * THREADS * 8)); CHECK(hipMalloc(&tw, 1024 * 8)); CHECK(hipMalloc(&clk, blocks * 16));
    const u64 q = 0x3fffffffd60001ull; u64 qi = q; for (int i = 0; i < 6; ++i) qi *= 2 - q * qi;
    u64 h[1024]; for (int i = 0; i < 1024; ++i) h[i] = NEW ? split30((i64)((0x123456789abcdefull * (i + 1)) % q), false) : sd_split((0x123456789abcdefull * (i + 1)) % q);
    CHECK(hipMemcpy(tw, h, sizeof(h), hipMemcpyHostToDevice));
    const int reps = 400;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const u64 qs = NEW ? split30((i64)q, true) : sd_split(q);
    k<NEW, THREADS, WPE><<<blocks, THREADS>>>(out, tw, qs, (u32)(0 - qi), reps, clk); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); k<NEW, THREADS, WPE><<<blocks, THREADS>>>(out, tw, qs, (u32)(0 - qi), reps, clk); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
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
    if (check()) return 1;
    run<0, 1024, 8>("mm31  (8 + 3 + add/sub),  8 waves/SIMD", 512);
    run<1, 1024, 8>("mm30u (8 + 1 + add/sub),  8 waves/SIMD", 512);
    run<0, 1024, 8>("mm31  (8 + 3 + add/sub),  4 waves/SIMD", 256);
    run<1, 1024, 8>("mm30u (8 + 1 + add/sub),  4 waves/SIMD", 256);
    return 0;
}
