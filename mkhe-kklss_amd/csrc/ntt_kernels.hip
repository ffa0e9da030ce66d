// ntt_kernels.hip -- see ntt_kernels.h for the design.
#include "ntt_kernels.h"

namespace mkhe {

// ------------------------------------------------------------------ layouts
template <int LOGN> struct Geo {
    static constexpr int N = 1 << LOGN;
    static constexpr int T = N / 32;                  // threads per limb
    static constexpr int BT = T < 64 ? 64 : T;        // block size
    static constexpr int LPB = BT / T;                // limbs per block
    static constexpr int MIDTOP = (LOGN - 6) < 9 ? (LOGN - 6) : 9;   // highest bit handled by the middle phase
    static constexpr bool HAS_MID = MIDTOP >= 5;
};

template <int LOGN> __device__ __forceinline__ int posA(int t, int r) { return (r << (LOGN - 5)) | t; }
__device__ __forceinline__ int posB(int t, int r) { return ((t >> 5) << 10) | (r << 5) | (t & 31); }
__device__ __forceinline__ int posC(int t, int r) { return (t << 5) | r; }
__device__ __forceinline__ int swz(int p) { return p ^ ((p >> 5) & 31); }

enum Layout { LA = 0, LB = 1, LC = 2 };
template <int LOGN, int L> __device__ __forceinline__ int pos(int t, int r) {
    if constexpr (L == LA) return posA<LOGN>(t, r);
    else if constexpr (L == LB) return posB(t, r);
    else return posC(t, r);
}

// Re-distribute the 32 registers of every thread from layout FROM to layout TO through LDS,
// one 32-bit plane at a time.
template <int LOGN, int FROM, int TO>
__device__ __forceinline__ void exchange(u64 (&x)[32], u32* lds, int t) {
    u32 lo[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) lds[swz(pos<LOGN, FROM>(t, r))] = lo32(x[r]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; ++r) lo[r] = lds[swz(pos<LOGN, TO>(t, r))];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; ++r) lds[swz(pos<LOGN, FROM>(t, r))] = hi32(x[r]);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = ((u64)lds[swz(pos<LOGN, TO>(t, r))] << 32) | lo[r];
    __syncthreads();
}

// ------------------------------------------------------------------ butterflies
// Forward (Harvey lazy): U,V in [0,4q) -> X,Y in [0,4q).
__device__ __forceinline__ void bfly_fwd(u64& U, u64& V, u64 w, u64 q, u64 q2, u32 ninv) {
    u64 Tm = mont_mul_lazy(V, w, q, ninv);
    u64 u = csub(U, q2);
    U = u + Tm;
    V = u + (q2 - Tm);
}
// Inverse (Gentleman-Sande): U,V in [0,2q) -> X,Y in [0,2q).
__device__ __forceinline__ void bfly_inv(u64& U, u64& V, u64 w, u64 q, u64 q2, u32 ninv) {
    u64 s = csub(U + V, q2);
    u64 d = U + q2 - V;
    U = s;
    V = mont_mul_lazy(d, w, q, ninv);
}

// One radix-2 stage on register bit B; tw points at the run of (16 >> B) twiddles.
template <int B, bool INV>
__device__ __forceinline__ void stage(u64 (&x)[32], const u64* __restrict__ tw, u64 q, u64 q2, u32 ninv) {
    constexpr int NW = 16 >> B;
    u64 w[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] = tw[k];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
        const int i1 = i0 | (1 << B);
        if constexpr (INV) bfly_inv(x[i0], x[i1], w[g >> B], q, q2, ninv);
        else bfly_fwd(x[i0], x[i1], w[g >> B], q, q2, ninv);
    }
}

template <int LOGN>
__device__ __forceinline__ void job_pointers(const NttBatch& b, int job, const u64*& src, u64*& dst, int& m, int& outer) {
    outer = job / b.inner_count;
    const int s = job - outer * b.inner_count;
    m = b.map[s];
    src = b.src + (long)outer * b.src_outer + (long)(b.src_mapped ? m : s) * b.src_inner;
    dst = b.dst + (long)outer * b.dst_outer + (long)(b.dst_mapped ? m : s) * b.dst_inner;
}

// ------------------------------------------------------------------ forward kernel
template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::BT) ntt_fwd_kernel(NttBatch b) {
    using G = Geo<LOGN>;
    extern __shared__ __attribute__((aligned(16))) u32 lds_all[];
    const int sub = threadIdx.x / G::T, t = threadIdx.x % G::T;
    u32* lds = lds_all + sub * G::N;
    int job = blockIdx.x * G::LPB + sub;
    const bool active = job < b.njobs;
    if (!active) job = b.njobs - 1;            // keep every lane in the barriers; results discarded
    const u64* src; u64* dst; int m, outer;
    job_pointers<LOGN>(b, job, src, dst, m, outer);
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const u64* psi = b.psi + (long)m * G::N;

    u64 x[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = src[posA<LOGN>(t, r)];
    if (b.reduce_in) {
        // digit of a foreign modulus (Decompose, alpha = 1): bring it below 4q when needed
        const u64 qs = b.mods[b.reduce_src_mod_is_outer ? outer : m].q;
        if (qs > 4 * q) {
#pragma unroll
            for (int r = 0; r < 32; ++r) x[r] = mont_mul_lazy(x[r], md.r1, q, ninv);
        }
    }
    // phase 1: index bits n-1 .. n-5 (register bits 4..0); twiddles psi[2^(s-1) + k] are wave-uniform
    stage<4, false>(x, psi + 1, q, q2, ninv);
    stage<3, false>(x, psi + 2, q, q2, ninv);
    stage<2, false>(x, psi + 4, q, q2, ninv);
    stage<1, false>(x, psi + 8, q, q2, ninv);
    stage<0, false>(x, psi + 16, q, q2, ninv);

    if constexpr (G::HAS_MID) {
        exchange<LOGN, LA, LB>(x, lds, t);
        const int hi = t >> 5;
        // phase 2: index bits MIDTOP .. 5 (register bit B = beta - 5)
        if constexpr (G::MIDTOP >= 9) stage<4, false>(x, psi + (G::N >> 10) + (hi << 0), q, q2, ninv);
        if constexpr (G::MIDTOP >= 8) stage<3, false>(x, psi + (G::N >> 9) + (hi << 1), q, q2, ninv);
        if constexpr (G::MIDTOP >= 7) stage<2, false>(x, psi + (G::N >> 8) + (hi << 2), q, q2, ninv);
        if constexpr (G::MIDTOP >= 6) stage<1, false>(x, psi + (G::N >> 7) + (hi << 3), q, q2, ninv);
        stage<0, false>(x, psi + (G::N >> 6) + (hi << 4), q, q2, ninv);
    }
    exchange<LOGN, LB, LC>(x, lds, t);
    // phase 3: index bits 4..0
    stage<4, false>(x, psi + (G::N >> 5) + (t << 0), q, q2, ninv);
    stage<3, false>(x, psi + (G::N >> 4) + (t << 1), q, q2, ninv);
    stage<2, false>(x, psi + (G::N >> 3) + (t << 2), q, q2, ninv);
    stage<1, false>(x, psi + (G::N >> 2) + (t << 3), q, q2, ninv);
    stage<0, false>(x, psi + (G::N >> 1) + (t << 4), q, q2, ninv);
    // canonical output (lattigo: final BRedAdd)
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = csub(csub(x[r], q2), q);
    exchange<LOGN, LC, LB>(x, lds, t);
    if (active) {
#pragma unroll
        for (int r = 0; r < 32; ++r) dst[posB(t, r)] = x[r];
    }
}

// ------------------------------------------------------------------ inverse kernel
template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::BT) ntt_inv_kernel(NttBatch b) {
    using G = Geo<LOGN>;
    extern __shared__ __attribute__((aligned(16))) u32 lds_all[];
    const int sub = threadIdx.x / G::T, t = threadIdx.x % G::T;
    u32* lds = lds_all + sub * G::N;
    int job = blockIdx.x * G::LPB + sub;
    const bool active = job < b.njobs;
    if (!active) job = b.njobs - 1;
    const u64* src; u64* dst; int m, outer;
    job_pointers<LOGN>(b, job, src, dst, m, outer);
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const u64* psi = b.psi + (long)m * G::N;

    u64 x[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = src[posB(t, r)];
    exchange<LOGN, LB, LC>(x, lds, t);
    // phase A: index bits 0..4
    stage<0, true>(x, psi + (G::N >> 1) + (t << 4), q, q2, ninv);
    stage<1, true>(x, psi + (G::N >> 2) + (t << 3), q, q2, ninv);
    stage<2, true>(x, psi + (G::N >> 3) + (t << 2), q, q2, ninv);
    stage<3, true>(x, psi + (G::N >> 4) + (t << 1), q, q2, ninv);
    stage<4, true>(x, psi + (G::N >> 5) + (t << 0), q, q2, ninv);
    exchange<LOGN, LC, LB>(x, lds, t);
    if constexpr (G::HAS_MID) {
        const int hi = t >> 5;
        stage<0, true>(x, psi + (G::N >> 6) + (hi << 4), q, q2, ninv);
        if constexpr (G::MIDTOP >= 6) stage<1, true>(x, psi + (G::N >> 7) + (hi << 3), q, q2, ninv);
        if constexpr (G::MIDTOP >= 7) stage<2, true>(x, psi + (G::N >> 8) + (hi << 2), q, q2, ninv);
        if constexpr (G::MIDTOP >= 8) stage<3, true>(x, psi + (G::N >> 9) + (hi << 1), q, q2, ninv);
        if constexpr (G::MIDTOP >= 9) stage<4, true>(x, psi + (G::N >> 10) + (hi << 0), q, q2, ninv);
        exchange<LOGN, LB, LA>(x, lds, t);
    }
    // phase C: index bits n-5 .. n-2, then the last stage with N^-1 folded in
    stage<0, true>(x, psi + 16, q, q2, ninv);
    stage<1, true>(x, psi + 8, q, q2, ninv);
    stage<2, true>(x, psi + 4, q, q2, ninv);
    stage<3, true>(x, psi + 2, q, q2, ninv);
    const u64 ninvR = b.aux[2 * m], w1n = b.aux[2 * m + 1];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        u64 U = x[g], V = x[g + 16];
        u64 s = mont_mul_lazy(U + V, ninvR, q, ninv);
        u64 d = mont_mul_lazy(U + q2 - V, w1n, q, ninv);
        x[g] = b.lazy_out ? s : csub(s, q);
        x[g + 16] = b.lazy_out ? d : csub(d, q);
    }
    if (active) {
#pragma unroll
        for (int r = 0; r < 32; ++r) dst[posA<LOGN>(t, r)] = x[r];
    }
}

// ------------------------------------------------------------------ launchers
template <int LOGN> static void launch_fwd_t(const NttBatch& b, hipStream_t st) {
    using G = Geo<LOGN>;
    static bool attr = false;
    const size_t lds = (size_t)G::LPB * G::N * sizeof(u32);
    if (!attr) { (void)hipFuncSetAttribute((const void*)ntt_fwd_kernel<LOGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    const int blocks = (b.njobs + G::LPB - 1) / G::LPB;
    hipLaunchKernelGGL(ntt_fwd_kernel<LOGN>, dim3(blocks), dim3(G::BT), lds, st, b);
}
template <int LOGN> static void launch_inv_t(const NttBatch& b, hipStream_t st) {
    using G = Geo<LOGN>;
    static bool attr = false;
    const size_t lds = (size_t)G::LPB * G::N * sizeof(u32);
    if (!attr) { (void)hipFuncSetAttribute((const void*)ntt_inv_kernel<LOGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    const int blocks = (b.njobs + G::LPB - 1) / G::LPB;
    hipLaunchKernelGGL(ntt_inv_kernel<LOGN>, dim3(blocks), dim3(G::BT), lds, st, b);
}

void launch_ntt_fwd(int logN, const NttBatch& b, hipStream_t st) {
    if (b.njobs <= 0) return;
    switch (logN) {
        case 10: launch_fwd_t<10>(b, st); break;
        case 11: launch_fwd_t<11>(b, st); break;
        case 12: launch_fwd_t<12>(b, st); break;
        case 13: launch_fwd_t<13>(b, st); break;
        case 14: launch_fwd_t<14>(b, st); break;
        case 15: launch_fwd_t<15>(b, st); break;
        default: break;
    }
}
void launch_ntt_inv(int logN, const NttBatch& b, hipStream_t st) {
    if (b.njobs <= 0) return;
    switch (logN) {
        case 10: launch_inv_t<10>(b, st); break;
        case 11: launch_inv_t<11>(b, st); break;
        case 12: launch_inv_t<12>(b, st); break;
        case 13: launch_inv_t<13>(b, st); break;
        case 14: launch_inv_t<14>(b, st); break;
        case 15: launch_inv_t<15>(b, st); break;
        default: break;
    }
}

}  // namespace mkhe
