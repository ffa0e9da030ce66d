// HBM read ceiling for the streaming kernels (inner_product_kernel, ext_inner*): what does a kernel that ONLY reads reach on this part, with the
// access pattern of those kernels -- S concurrent streams per thread (one per digit / operand, 4 MB apart), 16 bytes per lane and load, the sum
// written once per thread?  Compare with their 5.8-5.9 TB/s (profiles/README.md).   hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int S, bool NT, int U>
__global__ void __launch_bounds__(256) rd(const u64* __restrict__ a, u64* __restrict__ out, long stream_words, long n_per_stream) {
    // thread t of the grid reads words 2 t, 2 t + 1 of every stream (like a coefficient pair of every digit)
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (2 * t >= n_per_stream) return;
    u64 s0 = 0, s1 = 0;
#pragma unroll U
    for (int i = 0; i < S; ++i) {
        const u64x2* p = (const u64x2*)(a + (long)i * stream_words + 2 * t);
        const u64x2 v = NT ? __builtin_nontemporal_load(p) : *p;
        s0 += v.x; s1 ^= v.y;
    }
    u64x2 r; r.x = s0; r.y = s1;
    *(u64x2*)(out + 2 * t) = r;
}
// contiguous: block b reads one contiguous region of S * 4 KiB (S consecutive 4 KiB rows), i.e. DRAM-page-friendly
template <int S, bool NT, int U>
__global__ void __launch_bounds__(256) rdc(const u64* __restrict__ a, u64* __restrict__ out, long stream_words, long n_per_stream) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (2 * t >= n_per_stream) return;
    const u64* base = a + (long)blockIdx.x * (512L * S) + 2 * threadIdx.x;
    u64 s0 = 0, s1 = 0;
#pragma unroll U
    for (int i = 0; i < S; ++i) {
        const u64x2* p = (const u64x2*)(base + 512L * i);
        const u64x2 v = NT ? __builtin_nontemporal_load(p) : *p;
        s0 += v.x; s1 ^= v.y;
    }
    u64x2 r; r.x = s0; r.y = s1;
    *(u64x2*)(out + 2 * t) = r;
}
template <int S, bool NT, int U> void runc(const char* name, const u64* a, u64* out, long stream_words, long n) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = (int)((n / 2 + 255) / 256);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rdc<S, NT, U>), dim3(blocks), dim3(256), 0, 0, a, out, stream_words, n);
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((rdc<S, NT, U>), dim3(blocks), dim3(256), 0, 0, a, out, stream_words, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 8.0 * n * (S + 1);
    printf("%-44s rows    %3d  %7.1f MB per launch  %8.1f us  %7.1f GB/s\n", name, S, bytes / 1e6, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e9);
}
template <int S, bool NT, int U> void run(const char* name, const u64* a, u64* out, long stream_words, long n) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = (int)((n / 2 + 255) / 256);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((rd<S, NT, U>), dim3(blocks), dim3(256), 0, 0, a, out, stream_words, n);
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((rd<S, NT, U>), dim3(blocks), dim3(256), 0, 0, a, out, stream_words, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 8.0 * n * (S + 1);
    printf("%-44s streams %3d  %7.1f MB per launch  %8.1f us  %7.1f GB/s\n", name, S, bytes / 1e6, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e9);
}
int main() {
    const long n = 1 << 19 << 4;            // words per stream per launch: 16 limbs of 2^15 words... 8M words = 64 MB per stream
    const int SMAX = 70;
    const long stream_words = n;              // streams back to back
    u64 *a, *out;
    CK(hipMalloc(&a, (size_t)SMAX * stream_words * 8)); CK(hipMalloc(&out, (size_t)n * 8));
    CK(hipMemset(a, 1, (size_t)SMAX * stream_words * 8));
    run<1, false, 1>("1 stream, cached loads (fits the Infinity Cache)", a, out, stream_words, n);
    // one long stream: every thread reads S consecutive 16-byte words 4 KiB apart inside ONE contiguous region (a grid-strided copy-like read)
    run<70, true, 8>("70 x 64 MB as one region (stride = stream)", a, out, stream_words, n);
    run<14, false, 4>("14 streams, cached loads, unroll 4", a, out, stream_words, n);
    run<14, true, 4>("14 streams, nontemporal, unroll 4", a, out, stream_words, n);
    run<14, true, 14>("14 streams, nontemporal, unroll 14", a, out, stream_words, n);
    run<28, true, 4>("28 streams, nontemporal, unroll 4", a, out, stream_words, n);
    run<28, true, 8>("28 streams, nontemporal, unroll 8", a, out, stream_words, n);
    run<70, true, 8>("70 streams, nontemporal, unroll 8", a, out, stream_words, n);
    run<70, false, 8>("70 streams, cached, unroll 8", a, out, stream_words, n);
    runc<70, true, 8>("contiguous 280 KiB per block, nontemporal", a, out, stream_words, n);
    runc<14, true, 14>("contiguous 56 KiB per block, nontemporal", a, out, stream_words, n);
    runc<70, false, 8>("contiguous 280 KiB per block, cached", a, out, stream_words, n);
    return 0;
}
