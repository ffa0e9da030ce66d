// Microbenchmark #7 (round 3, experiment): the butterfly in double precision for moduli below 2^46.5 (the 45-bit primes of PN16QP1761), against
// mm30u.  FP64 FMA issues at full rate on CDNA4 where the 32 x 32 -> 64 multiply-adds of the integer product issue at a quarter of it; a modular
// product of exact integers held in doubles is an error-free transformation:
//   h = a w (rounded), l = fma(a, w, -h) (the exact rest), k = rint(h / q) (via the rounded reciprocal: off by at most 2), r = fma(-k, q, h) (exact),
//   T = r + l = a w - k q exactly, |T| <= 2.5 q  --  6 instructions + add / subtract.
// The question is what the chip's clock does under them at the 1400 W cap.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o bflyf64_rate bflyf64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef unsigned long long u64;
typedef long long i64;

__device__ __forceinline__ double mmf(double a, double w, double q, double qinv) {
    const double h = a * w;
    const double l = __builtin_fma(a, w, -h);
    const double k = __builtin_rint(h * qinv);
    const double r = __builtin_fma(-k, q, h);
    return r + l;
}
template <int THREADS, int WPE>
__global__ void __launch_bounds__(THREADS, WPE) k(double* out, const double* tw, double q, double qinv, int reps, unsigned long long* clk) {
    double x[16];
    for (int r = 0; r < 16; ++r) x[r] = (double)(threadIdx.x * 977 + r * 131 + 7);
    double w[8];
    for (int i = 0; i < 8; ++i) { w[i] = ((const __attribute__((address_space(4))) double*)tw)[i]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int B = 3; B >= 0; --B) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1)), i1 = i0 | (1 << B);
                const double T = mmf(x[i1], w[g >> B], q, qinv);
                const double u = x[i0];
                x[i0] = u + T; x[i1] = u - T;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int r = 0; r < 16; ++r) x[r] = __builtin_fma(-__builtin_rint(x[r] * qinv), q, x[r]);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double acc = 0; for (int r = 0; r < 16; ++r) acc += x[r];
    out[blockIdx.x * THREADS + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
__global__ void chk(const double* a, double w, double q, double qinv, double* T, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) T[i] = mmf(a[i], w, q, qinv);
}
int check() {
    const u64 q = 0x200000860001ull;                 // a 45-bit prime of PN16QP1761
    const int n = 1 << 16;
    static double ha[1 << 16], hT[1 << 16];
    u64 s = 12345;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < n; ++i) { ha[i] = (double)(i64)(rnd() >> 12) * ((rnd() & 1) ? 1.0 : -1.0); if (i < 8) ha[i] = (i & 1 ? -1.0 : 1.0) * (9007199254740991.0 - i); }
    const u64 w = rnd() % q;
    double *da, *dT;
    CHECK(hipMalloc(&da, n * 8)); CHECK(hipMalloc(&dT, n * 8));
    CHECK(hipMemcpy(da, ha, n * 8, hipMemcpyHostToDevice));
    chk<<<n / 256, 256>>>(da, (double)w, (double)q, 1.0 / (double)q, dT, n);
    CHECK(hipMemcpy(hT, dT, n * 8, hipMemcpyDeviceToHost));
    int bad = 0; double lo = 0, hi = 0;
    for (int i = 0; i < n; ++i) {
        __int128 e = ((__int128)(i64)ha[i] * (i64)w) % (__int128)q; if (e < 0) e += q;
        const double t = hT[i];
        __int128 g = (__int128)(i64)t % (__int128)q; if (g < 0) g += q;
        const double rel = t / (double)q;
        if (rel < lo) lo = rel; if (rel > hi) hi = rel;
        if (t != std::floor(t) || e != g || std::fabs(rel) > 2.5) { if (bad < 5) printf("MISMATCH i=%d a=%.0f T=%.0f\n", i, ha[i], t); ++bad; }
    }
    printf("mmf check: %d values (|a| up to 2^53), %d bad, T/q in [%.3f, %.3f]\n", n, bad, lo, hi);
    return bad;
}
template <int THREADS, int WPE> int run(const char* name, int blocks) {
    double *out, *tw; unsigned long long* clk;
    CHECK(hipMalloc(&out, (size_t)blocks * THREADS * 8)); CHECK(hipMalloc(&tw, 1024 * 8)); CHECK(hipMalloc(&clk, blocks * 16));
    const u64 q = 0x200000860001ull;
    double h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (double)((0x123456789abcdefull * (i + 1)) % q);
    CHECK(hipMemcpy(tw, h, sizeof(h), hipMemcpyHostToDevice));
    const int reps = 400;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int warm = 0; warm < 40; ++warm) k<THREADS, WPE><<<blocks, THREADS>>>(out, tw, (double)q, 1.0 / (double)q, reps, clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int it = 0; it < 20; ++it) k<THREADS, WPE><<<blocks, THREADS>>>(out, tw, (double)q, 1.0 / (double)q, reps, clk);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    static unsigned long long hc[16384]; CHECK(hipMemcpy(hc, clk, blocks * 16, hipMemcpyDeviceToHost));
    double ticks = 0, rt = 0; for (int i = 0; i < blocks; ++i) { ticks += hc[2 * i]; rt += hc[2 * i + 1]; }
    double ghz = ticks / rt / 10.0;
    double waves_per_simd = (double)blocks * THREADS / 64 / (256.0 * 4.0);
    double ns = ms * 1e6 / (waves_per_simd * reps * 32.0);
    printf("%-44s %7.3f ms  clock %.2f GHz  %6.2f ns per wave-butterfly per SIMD (%5.1f cyc)\n", name, ms, ghz, ns, ns * ghz);
    return 0;
}
int main() {
    if (check()) return 1;
    run<1024, 8>("f64 butterfly (6 + add/sub), 8 waves/SIMD, steady state", 512);
    run<1024, 8>("f64 butterfly (6 + add/sub), 4 waves/SIMD, steady state", 256);
    return 0;
}
