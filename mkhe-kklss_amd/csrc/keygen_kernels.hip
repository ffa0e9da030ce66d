// keygen_kernels.hip -- see keygen_kernels.h.  Streaming kernels: 256 threads, grid.x strides over the coefficients,
// grid.y = limb, grid.z = digit / polynomial; per-modulus constants are wave-uniform (SGPRs).
#include "keygen_kernels.h"

namespace mkhe {

constexpr int KG_THREADS = 256;
static int kg_bx(int N) { return (N + KG_THREADS - 1) / KG_THREADS; }

__global__ void __launch_bounds__(KG_THREADS) small_expand_kernel(u64* dst, const i32* small, const Mod* mods, int limbs, int N) {
    const int j = blockIdx.y, c = blockIdx.z;
    const u64 q = mods[j].q;
    for (int n = blockIdx.x * KG_THREADS + threadIdx.x; n < N; n += gridDim.x * KG_THREADS) {
        const i32 s = small[(long)c * N + n];
        dst[((long)c * limbs + j) * N + n] = s < 0 ? q - (u64)(-(i64)s) : (u64)s;
    }
}
void launch_small_expand(u64* dst, const i32* small, const Mod* mods, int count, int limbs, int N, hipStream_t st) {
    hipLaunchKernelGGL(small_expand_kernel, dim3(kg_bx(N), limbs, count), dim3(KG_THREADS), 0, st, dst, small, mods, limbs, N);
}

__global__ void __launch_bounds__(KG_THREADS) keygen_combine_kernel(KeygenArgs a) {
    const int j = blockIdx.y, i = blockIdx.z;
    const Mod md = a.mods[j];
    const u64 q = md.q;
    const u64 g = a.g ? a.g[(long)i * a.mtot + j] : 0;
    const long base = ((long)i * a.mtot + j) * a.N, lim = (long)j * a.N;
    for (int n = blockIdx.x * KG_THREADS + threadIdx.x; n < a.N; n += gridDim.x * KG_THREADS) {
        u64 v = a.out[base + n];
        if (a.mform_e) v = mont_mul(v, md.r2, q, md.ninv32);
        if (g) v = csub(v + mont_mul(a.skA[lim + n], g, q, md.ninv32), q);
        if (a.crs) {
            const u64 t = mont_mul(a.crs[base + n], a.skB[lim + n], q, md.ninv32);
            v = csub(a.sign > 0 ? v + t : v + (q - t), q);
        }
        if (a.neg) v = q - v;
        a.out[base + n] = v;
    }
}
void launch_keygen_combine(const KeygenArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(keygen_combine_kernel, dim3(kg_bx(a.N), a.mtot, a.beta), dim3(KG_THREADS), 0, st, a);
}

__global__ void __launch_bounds__(KG_THREADS) permute_ntt_kernel(u64* dst, const u64* src, int logN, u64 galEl) {
    const int N = 1 << logN, j = blockIdx.y;
    const u64 mask = 2 * (u64)N - 1;
    for (int n = blockIdx.x * KG_THREADS + threadIdx.x; n < N; n += gridDim.x * KG_THREADS) {
        const u64 t1 = 2 * (u64)(__brev((u32)n) >> (32 - logN)) + 1;
        const u64 t2 = ((galEl * t1 & mask) - 1) >> 1;
        const u32 idx = __brev((u32)t2) >> (32 - logN);
        dst[(long)j * N + n] = src[(long)j * N + idx];
    }
}
void launch_permute_ntt(u64* dst, const u64* src, int limbs, int logN, u64 galEl, hipStream_t st) {
    hipLaunchKernelGGL(permute_ntt_kernel, dim3(kg_bx(1 << logN), limbs), dim3(KG_THREADS), 0, st, dst, src, logN, galEl);
}

// Philox4x32-10 (Salmon et al., SC'11): 10 rounds of two 32x32 -> 64 multiplies and a key bump
__device__ __forceinline__ void philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1, u32 (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const u64 p0 = (u64)0xD2511F53u * c0, p1 = (u64)0xCD9E8D57u * c2;
        const u32 n0 = hi32(p1) ^ c1 ^ k0, n2 = hi32(p0) ^ c3 ^ k1;
        c1 = lo32(p1); c3 = lo32(p0); c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__global__ void __launch_bounds__(KG_THREADS) crs_expand_kernel(u64* out, const Mod* mods, u64 seed, i32 idx, int mtot, int N) {
    const int j = blockIdx.y, i = blockIdx.z;
    const Mod md = mods[j];
    const u64 q = md.q;
    const u64 mask = ~0ull >> __clzll((long long)q);
    const u32 row = (u32)(i * mtot + j);
    for (int n = blockIdx.x * KG_THREADS + threadIdx.x; n < N; n += gridDim.x * KG_THREADS) {
        u64 v;
        for (u32 block = 0;; ++block) {            // every candidate is accepted with probability q / 2^bitlen(q) > 1/2
            u32 o[4];
            philox4x32_10((u32)n, row, (u32)idx, block, lo32(seed), hi32(seed), o);
            const u64 c0 = (((u64)o[1] << 32) | o[0]) & mask, c1 = (((u64)o[3] << 32) | o[2]) & mask;
            if (c0 < q) { v = c0; break; }
            if (c1 < q) { v = c1; break; }
        }
        out[((long)i * mtot + j) * N + n] = mont_mul(v, md.r2, q, md.ninv32);
    }
}
void launch_crs_expand(u64* out, const Mod* mods, u64 seed, i32 idx, int beta, int mtot, int N, hipStream_t st) {
    hipLaunchKernelGGL(crs_expand_kernel, dim3(kg_bx(N), mtot, beta), dim3(KG_THREADS), 0, st, out, mods, seed, idx, mtot, N);
}

}  // namespace mkhe
