// h16_core.h -- the pieces of the H16 forward / inverse NTT kernels (ntt16_kernels.hip) that the fused Decompose + inner-product kernel
// (ntt16_f2_kernels.hip) shares with them: geometry, the never-reduced butterflies on the one-round products, the radix-2 stages on register
// bits, and the four LDS re-distributions of a 2^14-point pass.  Device code only; derivations in ntt16_kernels.hip.
#pragma once
#include "ntt_kernels.h"
#include "h16_arith.h"
#include <utility>

namespace mkhe {
namespace h16 {

constexpr int NN = 1 << 15, HH = 1 << 14, NT = 1024;
constexpr int WSTR = 1088;                 // LDS words per wave region (1024 + padding of the wave-local layouts)
constexpr int LDS_WORDS = 16 * WSTR;

// stage-0 source loads: SG pairs in flight per thread (4 instead of 8 changes neither the spills the compiler places around stage 0
// nor the time: 224 vs 222 us per average launch)
constexpr int SG = 8;
__device__ __forceinline__ void ld_wait(u64 (&a)[4], u64 (&b)[4]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
}
__device__ __forceinline__ void ld_wait(u64 (&a)[8], u64 (&b)[8]) { ld_wait16(a, b); }
// pipelined stage-0 loads (round 3): PIPE_P pairs requested up front, the rest one by one behind every second consumed pair / the stash;
// pipe_issued(r) = pairs requested by the time pair r is waited for
#ifndef MKHE_H16_PIPE_P
#define MKHE_H16_PIPE_P 11
#endif
constexpr int PIPE_P = MKHE_H16_PIPE_P;
constexpr int pipe_issued(int r) {
    int n = PIPE_P;
    for (int k = 0; k < r && k < 16; ++k) if (n < 16 && ((k & 1) == 1 || k >= 7)) ++n;      // after pair k has been consumed
    return n;
}
template <int... I, class F> __device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int K> __device__ __forceinline__ void ld_wait_pair(u64& a, u64& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(K)); }
// Signed butterfly, never reduced: X = U + w V, Y = U - w V with the product as a signed representative; |x| grows by less than
// q/2 + |x|/16 per stage.  Both modulus classes use it: MODE 1 (31q < 2^62: the 54-bit primes) runs all 15 stages without any
// reduction, MODE 0 (q < 2^60) interposes the partial reduction above after at most 8 stages (limb() below) instead of the
// conditional subtractions of a Harvey butterfly (+50 % instructions per butterfly in the first version of this kernel).
template <bool SW> __device__ __forceinline__ void bfly(u64& U, u64& V, u64 ws, const MC& c) {
#ifdef MKHE_H16_X_NOBFLY
    if ((MKHE_H16_X_NOBFLY >> (SW ? 0 : 1)) & 1) { U += ws; return; }      // timing experiment only: butterflies with scalar / per-lane twiddles removed
#endif
    const i64 T = mm<SW>((i64)V, ws, c);
    const i64 u = (i64)U;
    U = (u64)(u + T);
    V = (u64)(u - T);
}
// the same butterfly on the one-round product; tw[0] = u, tw[1] = v
template <bool SW, bool UC> __device__ __forceinline__ void bfly31(u64& U, u64& V, const u64* tw, const MC& c) {
#ifdef MKHE_H16_X_NOBFLY
    if ((MKHE_H16_X_NOBFLY >> (SW ? 0 : 1)) & 1) { U += tw[0] + tw[1]; return; }      // timing experiment only
#endif
    const i64 T = UC ? mm30u<SW>((i64)V, tw[0], tw[1], c) : mm31<SW>((i64)V, tw[0], tw[1], c);
    const i64 u = (i64)U;
    U = (u64)(u + T);
    V = (u64)(u - T);
}
__device__ __forceinline__ void reduce_all(u64 (&x)[16], const MC& c, bool big) {
    if (big) {
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = (u64)pred((i64)x[r], c);
    }
}

// one radix-2 stage on register bit B of the 16 registers; tw: the 8 >> B twiddles of this thread for the stage
template <bool SW, int B> __device__ __forceinline__ void stage(u64 (&x)[16], const u64* tw, const MC& c) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
        bfly<SW>(x[i0], x[i0 | (1 << B)], tw[g >> B], c);
#ifndef MKHE_H16_NO_SCHEDBAR
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

// the same stage with scalar twiddle PAIRS: butterflies G0 .. G0 + NG - 1, tw[2 i], tw[2 i + 1] = (u, v) of twiddle (G0 >> B) + i
template <bool UC, int B, int G0 = 0, int NG = 8> __device__ __forceinline__ void stage31(u64 (&x)[16], const u64* tw, const MC& c) {
#pragma unroll
    for (int g = G0; g < G0 + NG; ++g) {
        const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
        bfly31<true, UC>(x[i0], x[i0 | (1 << B)], tw + 2 * ((g >> B) - (G0 >> B)), c);
#ifndef MKHE_H16_NO_SCHEDBAR
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

// butterfly number g (0..7) of the stage on register bit B, per-lane twiddle
template <int B> __device__ __forceinline__ void bfly1(u64 (&x)[16], int g, u64 w, const MC& c) {
    const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
    bfly<false>(x[i0], x[i0 | (1 << B)], w, c);
#ifndef MKHE_H16_NO_SCHEDBAR
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// the same on the one-round product: tw = the (u, v) pair of this lane's twiddle; RING pairs (4 VGPRs each) are live in phases C / D
#ifndef MKHE_H16_RING
#define MKHE_H16_RING 3
#endif
// U class, phase D on the one-round product too (pairs of constants per lane): measured slower (10 spilled VGPRs around it: 372 against 349 us
// for the 1792-limb launch), so phase D keeps the two-round product of the balanced path (8-byte twiddles) for both classes
#ifndef MKHE_H16_UD31
#define MKHE_H16_UD31 0
#endif
constexpr int RING = MKHE_H16_RING;
template <bool UC, int B> __device__ __forceinline__ void bfly1_31(u64 (&x)[16], int g, const u64* tw, const MC& c) {
    const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
    bfly31<false, UC>(x[i0], x[i0 | (1 << B)], tw, c);
#ifndef MKHE_H16_NO_SCHEDBAR
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// ------------------------------------------------------------------ LDS re-distributions
// word offsets of register r in the four exchanges (write side, read side); bases are per thread (below)
enum { X_AB = 0, X_BC = 1, X_CD = 2, X_DE = 3 };
template <int X> constexpr int woff(int r) {
    return X == X_AB ? r * WSTR : X == X_BC ? 68 * r : X == X_CD ? 65 * (r >> 2) + 260 * (r & 3) : r;
}
template <int X> constexpr int roff(int r) {
    return X == X_AB ? 64 * r : X == X_BC ? 4 * r : X == X_CD ? 260 * (r >> 2) + (r & 3) : 66 * r;
}
__device__ __forceinline__ int de_w(int c) { return 16 * c + (c >> 1); }     // {0, 16, 33, 49}
template <int X> __device__ __forceinline__ int wbase(int wv, int l) {
    if constexpr (X == X_AB) return wv * 64 + l;
    else if constexpr (X == X_DE) return wv * WSTR + 66 * (l >> 2) + de_w(l & 3);
    else return wv * WSTR + l;
}
template <int X> __device__ __forceinline__ int rbase(int wv, int l) {
    if constexpr (X == X_AB) return wv * WSTR + l;
    else if constexpr (X == X_BC) return wv * WSTR + 68 * (l >> 2) + (l & 3);
    else if constexpr (X == X_CD) return wv * WSTR + 4 * (l >> 2) + 65 * (l & 3);
    else return wv * WSTR + de_w(l >> 4) + (l & 15);
}
// Round 3: the WRITE side of the A -> B, B -> C and C -> D re-distributions is lane-linear (word address = wave-uniform base + lane +
// constant(register)), which is what ds_write_addtid_b32 does without an address register: LDS address = M0 + 16-bit offset + 4 * lane.
// Its data path costs 2 cycles per wave-instruction where ds_write_b32 (address + data VGPR) costs 4 (MI355X_MICROARCH.md, LDS), and the
// exchanges are bound by exactly that path when the sixteen waves of a workgroup go through them together.
#ifndef MKHE_H16_ADDTID
#define MKHE_H16_ADDTID 1
#endif
#ifndef MKHE_H16_PRIO
#define MKHE_H16_PRIO 1
#endif
#ifndef MKHE_H16_URED
#define MKHE_H16_URED 1          // 0: the round-2 rule (reduce digits above 4q at the load) for the U class too (A/B)
#endif
template <int X, int R0> __device__ __forceinline__ void addtid_write8(const u32 (&w)[8], unsigned base_bytes) {
    asm volatile("s_mov_b32 m0, %8\n\t"
                 "s_nop 0\n\t"                              // (SALU write of M0 -> add-TID LDS instruction: one wait state; hazards inside an asm block are ours)
                 "ds_write_addtid_b32 %0 offset:%9\n\t"
                 "ds_write_addtid_b32 %1 offset:%10\n\t"
                 "ds_write_addtid_b32 %2 offset:%11\n\t"
                 "ds_write_addtid_b32 %3 offset:%12\n\t"
                 "ds_write_addtid_b32 %4 offset:%13\n\t"
                 "ds_write_addtid_b32 %5 offset:%14\n\t"
                 "ds_write_addtid_b32 %6 offset:%15\n\t"
                 "ds_write_addtid_b32 %7 offset:%16"
                 : : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "s"(base_bytes),
                     "n"(4 * woff<X>(R0 + 0)), "n"(4 * woff<X>(R0 + 1)), "n"(4 * woff<X>(R0 + 2)), "n"(4 * woff<X>(R0 + 3)),
                     "n"(4 * woff<X>(R0 + 4)), "n"(4 * woff<X>(R0 + 5)), "n"(4 * woff<X>(R0 + 6)), "n"(4 * woff<X>(R0 + 7))
                 : "m0", "memory");
}
template <int X, bool HI> __device__ __forceinline__ void write_plane(const u64 (&x)[16], u32* lds, int wv, int l) {
    typedef __attribute__((address_space(3))) u32* lptr;
    if constexpr (MKHE_H16_ADDTID && X != X_DE) {
        static_assert(4 * woff<X>(15) <= 65535, "16-bit offset field");
        const unsigned base = (unsigned)(unsigned long)(lptr)lds + 4u * (unsigned)wbase<X>(wv, 0);      // wave-uniform
        u32 a[8], b[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { a[r] = HI ? hi32(x[r]) : lo32(x[r]); b[r] = HI ? hi32(x[8 + r]) : lo32(x[8 + r]); }
        addtid_write8<X, 0>(a, base);
        addtid_write8<X, 8>(b, base);
    } else {
        lptr wr = (lptr)lds + wbase<X>(wv, l);
#pragma unroll
        for (int r = 0; r < 16; ++r) wr[woff<X>(r)] = HI ? hi32(x[r]) : lo32(x[r]);
    }
}
template <int X> __device__ __forceinline__ void exchange(u64 (&x)[16], u32* lds, int wv) {
    constexpr bool CROSS = X == X_AB;
#ifdef MKHE_H16_X_NOXCHG
    if ((MKHE_H16_X_NOXCHG >> X) & 1) return;      // timing experiment only (wrong results): cost of this re-distribution
#endif
    // the (loop-invariant) LDS bases are recomputed next to their use instead of living in VGPRs across the whole job
    const int l = lane_id();
    typedef __attribute__((address_space(3))) u32* lptr;
    typedef volatile __attribute__((address_space(3))) u32* vlptr;
    // The reads are volatile so that they stay single ds_read_b32: merged into ds_read2_b32 the two words of one instruction
    // (same plane, two different coefficients) land in a consecutive register pair and every coefficient then needs two v_mov to
    // get its own (low, high) pair back -- 32 VALU instructions per re-distribution, in a kernel that is VALU-issue bound.
    vlptr rd = (vlptr)((lptr)lds + rbase<X>(wv, l));
    if constexpr (CROSS) __syncthreads();          // every wave is done reading its region (previous pass)
    write_plane<X, false>(x, lds, wv, l);
    xsync<CROSS>();
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = (x[r] & 0xffffffff00000000ull) | rd[roff<X>(r)];
    xsync<CROSS>();
    write_plane<X, true>(x, lds, wv, l);
    xsync<CROSS>();
#ifdef MKHE_H16_FLOW
    // experiment: the high words are requested in the order in which the first stage of the next phase consumes them (pairs r, r + 8) and
    // nothing waits for all of them: the compiler's counted lgkmcnt waits let the first butterflies start while the later words are in flight
    // (a wave's LDS operations execute in order, so the next write to this region -- a whole phase later -- needs no fence)
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int r = (k >> 1) | ((k & 1) << 3); x[r] = ((u64)rd[roff<X>(r)] << 32) | lo32(x[r]); }
#else
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = ((u64)rd[roff<X>(r)] << 32) | lo32(x[r]);
    if constexpr (!CROSS) xsync<false>();
#endif
}

}  // namespace h16
}  // namespace mkhe
