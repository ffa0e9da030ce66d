// ntt_kernels.hip -- see ntt_kernels.h for the design.
//
// v2 structure notes (measured on MI355X, profiles/r1_*):
//  * one runtime loop over the three radix-32 phases shares the unrolled stage code, so the whole
//    kernel stays inside the instruction cache (the fully unrolled v1 was ~72 KB of straight-line code);
//  * only the A<->B re-distribution crosses waves (s_barrier); B<->C is half-wave local and uses
//    wave-level ordering only, so waves drift apart and LDS traffic overlaps other waves' VALU work;
//  * forward butterflies skip the conditional subtraction entirely when 34q < 2^63 (all 54-bit
//    primes of the shipped parameter sets): values then grow by 2q per stage and are normalised
//    once at the end with a Montgomery product by R mod q;
//  * all pointers are address_space(1) so loads are global_load (vmcnt only), never flat_load.
#include "ntt_kernels.h"
#include <stdexcept>
#include <cstdlib>
#include <mutex>
#ifndef MKHE_SB_MASK
#define MKHE_SB_MASK 1
#endif
#ifndef MKHE_TW_CHUNK
#define MKHE_TW_CHUNK 4
#endif

namespace mkhe {

typedef const __attribute__((address_space(1))) u64* gcptr;
typedef __attribute__((address_space(1))) u64* gptr;

// ------------------------------------------------------------------ layouts
template <int LOGN> struct Geo {
    static constexpr int N = 1 << LOGN;
    static constexpr int T = N / 32;                  // threads per limb
    static constexpr int BT = T < 64 ? 64 : T;        // block size
    static constexpr int LPB = BT / T;                // limbs per block
    static constexpr int MIDTOP = (LOGN - 6) < 9 ? (LOGN - 6) : 9;   // highest bit handled by the middle phase
    static constexpr int MIDB = MIDTOP - 5;           // highest register bit with work in the middle phase (-1: none)
    static constexpr bool HAS_MID = MIDTOP >= 5;
};

template <int LOGN> __device__ __forceinline__ int posA(int t, int r) { return (r << (LOGN - 5)) | t; }
__device__ __forceinline__ int posB(int t, int r) { return ((t >> 5) << 10) | (r << 5) | (t & 31); }
__device__ __forceinline__ int posC(int t, int r) { return (t << 5) | r; }

enum Layout { LA = 0, LB = 1, LC = 2 };

// LDS image of a limb plane: word address of coefficient p is p + (p >> 5) (one pad word per 32), so
// that every layout addresses its 32 registers as base(thread) + r * stride with a compile-time stride:
//   A: p = r*2^(n-5) + t        -> base t + (t>>5),          stride 2^(n-5) + 2^(n-10)
//   B: p = hi*1024 + r*32 + lo  -> base hi*1056 + lo,        stride 33
//   C: p = t*32 + r             -> base 33*t,                stride 1
// i.e. the LDS instructions carry immediate offsets and no per-access address arithmetic is issued.
// Conflict-free for all three under the 32-bank rule of ds_read_b32 / ds_write_b32 (lanes of a 32-lane
// group differ in t (A), lo (B: consecutive words) or t (C: stride 33 = 1 mod 32)).
template <int LOGN, int L> __device__ __forceinline__ int lds_base(int t) {
    if constexpr (L == LA) return t + (t >> 5);
    else if constexpr (L == LB) return (t >> 5) * 1056 + (t & 31);
    else return 33 * t;
}
template <int LOGN, int L> constexpr int lds_stride() {
    return L == LA ? ((1 << (LOGN - 5)) + (1 << (LOGN - 10))) : (L == LB ? 33 : 1);
}
template <int LOGN> constexpr int lds_words() { return (1 << LOGN) + (1 << (LOGN - 5)); }

// ordering point between LDS phases: workgroup barrier when the exchange crosses waves, otherwise
// only a compiler/wave-level fence (DS operations of one wave execute in issue order).
template <bool CROSS> __device__ __forceinline__ void lds_sync() {
    if constexpr (CROSS) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
}

// Re-distribute the 32 registers of every thread from layout FROM to layout TO through LDS,
// one 32-bit plane at a time (about N*4 bytes of LDS per limb).
template <int LOGN, int FROM, int TO, bool CROSS>
__device__ __forceinline__ void exchange(u64 (&x)[32], u32* lds, int t) {
#ifdef MKHE_X_NOXCHG
    return;      // timing experiment only (wrong results): cost of the LDS re-distribution
#endif
    // the thread index is made opaque here so that the (loop-invariant) LDS base addresses are recomputed
    // next to their use instead of being hoisted out of the persistent loop and spilled (VGPR budget 128)
    asm volatile("" : "+v"(t));
    u32* wr = lds + lds_base<LOGN, FROM>(t);
    u32* rd = lds + lds_base<LOGN, TO>(t);
    constexpr int SW = lds_stride<LOGN, FROM>(), SR = lds_stride<LOGN, TO>();
#pragma unroll
    for (int r = 0; r < 32; ++r) wr[r * SW] = lo32(x[r]);
    lds_sync<CROSS>();
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = (x[r] & 0xffffffff00000000ull) | rd[r * SR];
    lds_sync<CROSS>();
#pragma unroll
    for (int r = 0; r < 32; ++r) wr[r * SW] = hi32(x[r]);
    lds_sync<CROSS>();
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = ((u64)rd[r * SR] << 32) | lo32(x[r]);
    // no trailing workgroup barrier: the next LDS writer of any region is either its own half wave
    // (wave-local exchanges, ordered by issue order) or sits behind the explicit barrier of the next limb
    if constexpr (!CROSS) lds_sync<false>();
}

// XOR-swizzled variant (word address p ^ ((p >> 5) & 31), N words per plane, two VALU per access): used by the
// inverse kernel, whose register allocation is tighter with per-access addresses than with the base +
// immediate form above (scratch 60 B vs 288 B per lane, 68 vs 90 us per limb on MI355X).
template <int LOGN, int L> __device__ __forceinline__ int pos(int t, int r) {
    if constexpr (L == LA) return posA<LOGN>(t, r);
    else if constexpr (L == LB) return posB(t, r);
    else return posC(t, r);
}
__device__ __forceinline__ int swz(int p) { return p ^ ((p >> 5) & 31); }
template <int LOGN, int FROM, int TO, bool CROSS>
__device__ __forceinline__ void exchange_xor(u64 (&x)[32], u32* lds, int t) {
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int r = 0; r < 32; ++r) lds[swz(pos<LOGN, FROM>(t, r))] = lo32(x[r]);
    lds_sync<CROSS>();
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = (x[r] & 0xffffffff00000000ull) | lds[swz(pos<LOGN, TO>(t, r))];
    lds_sync<CROSS>();
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int r = 0; r < 32; ++r) lds[swz(pos<LOGN, FROM>(t, r))] = hi32(x[r]);
    lds_sync<CROSS>();
    asm volatile("" : "+v"(t));
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = ((u64)lds[swz(pos<LOGN, TO>(t, r))] << 32) | lo32(x[r]);
    if constexpr (!CROSS) lds_sync<false>();
}

// ------------------------------------------------------------------ butterflies
// Twiddles (ws) and the modulus (qs) are in signed-split form, see mont_mul_sd.
// Forward, reduced every stage (Harvey): U,V in [0,4q) -> [0,4q).
__device__ __forceinline__ void bfly_fwd_cs(u64& U, u64& V, u64 ws, u64 q, u64 q2, u64 qs, u32 ninv) {
    u64 Tm = mont_mul_sdu(V, ws, qs, q, ninv);
    u64 u = csub(U, q2);
    U = u + Tm;
    V = u + (q2 - Tm);
}
// Forward, never reduced, SIGNED values: |x| grows by less than q per stage (34q < 2^63 class): one add, one sub.
__device__ __forceinline__ void bfly_fwd_nr(u64& U, u64& V, u64 ws, u64 qs, u32 ninv) {
    const i64 Tm = mont_mul_sd((i64)V, ws, qs, ninv);
    const i64 u = (i64)U;
    U = (u64)(u + Tm);
    V = (u64)(u - Tm);
}
// Inverse (Gentleman-Sande): U,V in [0,2q) -> [0,2q).
__device__ __forceinline__ void bfly_inv(u64& U, u64& V, u64 ws, u64 q, u64 q2, u64 qs, u32 ninv) {
    u64 s = csub(U + V, q2);
    u64 d = U + q2 - V;
    U = s;
    V = mont_mul_sdu(d, ws, qs, q, ninv);
}

// One radix-2 stage on register bit B; tw points at the run of (16 >> B) twiddles.
// MODE: 0 forward reduced, 1 forward never-reduced, 2 inverse.
template <int B, int MODE>
__device__ __forceinline__ void stage(u64 (&x)[32], gcptr tw, u64 q, u64 q2, u64 qs, u32 ninv) {
    constexpr int NW = 16 >> B;                 // twiddles of this stage
#ifdef MKHE_NO_CHUNK
    constexpr int CH = NW;
#else
    constexpr int CH = NW < MKHE_TW_CHUNK ? NW : MKHE_TW_CHUNK;         // twiddles held at a time (one chunk in use + one in flight:
#endif
    constexpr int NCH = NW / CH;                //  <= 16 VGPRs instead of 32 at B = 0; the budget is 128)
    constexpr int BPC = 16 / NCH;               // butterflies per chunk
    u64 wc[CH], wn[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) wc[k] = tw[k];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c + 1 < NCH) {
#pragma unroll
            for (int k = 0; k < CH; ++k) wn[k] = tw[(c + 1) * CH + k];
        }
#pragma unroll
        for (int gg = 0; gg < BPC; ++gg) {
            const int g = c * BPC + gg;
            const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
            const int i1 = i0 | (1 << B);
            const u64 w = wc[(g >> B) - c * CH];
            if constexpr (MODE == 2) bfly_inv(x[i0], x[i1], w, q, q2, qs, ninv);
            else if constexpr (MODE == 1) bfly_fwd_nr(x[i0], x[i1], w, qs, ninv);
            else bfly_fwd_cs(x[i0], x[i1], w, q, q2, qs, ninv);
            // at 4 waves/SIMD the other waves hide the latency of one wave's dependency chain;
            // interleaving more than two butterflies only adds live temporaries
#ifndef MKHE_NO_SCHEDBAR
            if ((gg & MKHE_SB_MASK) == MKHE_SB_MASK) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        if (c + 1 < NCH) {
#pragma unroll
            for (int k = 0; k < CH; ++k) wc[k] = wn[k];
        }
    }
    // keep the next stage's twiddle loads from being hoisted above this stage (register pressure)
    asm volatile("" ::: "memory");
}

// the (up to) five stages of one phase; stage on register bit B uses
// psi[(base >> B) + (prefix << (4 - B)) + k].  FWD: most significant bit first; INV: least first.
template <int MODE>
__device__ __forceinline__ void phase(u64 (&x)[32], gcptr psi, int base, int prefix, int maxB, u64 q, u64 q2, u64 qs, u32 ninv) {
    if constexpr (MODE != 2) {
        if (maxB >= 4) stage<4, MODE>(x, psi + (base >> 4) + prefix, q, q2, qs, ninv);
        if (maxB >= 3) stage<3, MODE>(x, psi + (base >> 3) + (prefix << 1), q, q2, qs, ninv);
        if (maxB >= 2) stage<2, MODE>(x, psi + (base >> 2) + (prefix << 2), q, q2, qs, ninv);
        if (maxB >= 1) stage<1, MODE>(x, psi + (base >> 1) + (prefix << 3), q, q2, qs, ninv);
        if (maxB >= 0) stage<0, MODE>(x, psi + base + (prefix << 4), q, q2, qs, ninv);
    } else {
        if (maxB >= 0) stage<0, MODE>(x, psi + base + (prefix << 4), q, q2, qs, ninv);
        if (maxB >= 1) stage<1, MODE>(x, psi + (base >> 1) + (prefix << 3), q, q2, qs, ninv);
        if (maxB >= 2) stage<2, MODE>(x, psi + (base >> 2) + (prefix << 2), q, q2, qs, ninv);
        if (maxB >= 3) stage<3, MODE>(x, psi + (base >> 3) + (prefix << 1), q, q2, qs, ninv);
        if (maxB >= 4) stage<4, MODE>(x, psi + (base >> 4) + prefix, q, q2, qs, ninv);
    }
}

// The small per-launch lists are read straight from the kernarg segment (scalar loads): indexing the
// by-value struct dynamically would make the compiler copy all of it to scratch.
typedef const __attribute__((address_space(4))) NttBatch* kargptr;
// NttBatch::vi_parts[item] through a scalar dword load (a 16-bit load of a kernel argument is a VECTOR memory instruction, with a full vmcnt wait behind it)
__device__ __forceinline__ unsigned vi_parts_of(kargptr kb, int item) {
    return (((const __attribute__((address_space(4))) unsigned*)kb->vi_parts)[item >> 1] >> (16 * (item & 1))) & 0xffffu;
}
// Returns the number of source limbs to add up (1 everywhere but in the Q slots of a merged inverse launch, NttBatch::vi: the
// group's members and its Q-only extra summand; 0 = this job does not exist: a P slot of a member the group does not have).
template <bool VI = false>
__device__ __forceinline__ int job_pointers(const NttBatch& b, int job, gcptr& src, gptr& dst, int& m, int& outer, int* pm = nullptr) {
    kargptr kb = (kargptr)__builtin_amdgcn_kernarg_segment_ptr();
    // slot-major job order: consecutive workgroups share a modulus, so at any time the chip works
    // on 2-3 twiddle tables that stay resident in every XCD's L2
    const int s = job / b.nouter;
    outer = job - s * b.nouter;
    m = kb->mod[s];
    if (VI && b.vi) {
        const int cnt = kb->vi_cnt[outer];
        const int k = s < b.vi_q ? 0 : (s - b.vi_q) / b.vi_np;
        const bool exists = k < cnt;
        const unsigned mem = kb->vi_mem[outer];
        const int item = (int)((mem >> (exists ? 8 * k : 0)) & 255u);
        const long off = (long)item * b.src_outer + (long)m * b.src_inner;
        src = (gcptr)(b.src + off);
        dst = (gptr)(b.dst + off);
        if (pm) *pm = s < b.vi_q ? -1 : k;
        if (!exists) return 0;
        if (s >= b.vi_q) return 1 + (int)(vi_parts_of(kb, item) & 255u);               // P slot: the member's own parts
        int n = kb->vi_extra[outer] != nullptr ? 1 : 0;
        for (int j = 0; j < cnt; ++j) n += 1 + (int)(vi_parts_of(kb, (mem >> (8 * j)) & 255u) & 255u);
        return n;
    }
    if (pm) *pm = -1;
    const int p = kb->pos[s];
    const u64* sbase = b.src; u64* dbase = b.dst;
    if (b.nitems > 0) {
        const int item = outer / b.outers_per_item;
        outer -= item * b.outers_per_item;
        sbase = kb->src_items[item]; dbase = kb->dst_items[item];
    }
    src = (gcptr)(sbase + (long)outer * b.src_outer + (long)(b.src_mapped ? m : p) * b.src_inner);
    dst = (gptr)(dbase + (long)outer * b.dst_outer + (long)(b.dst_mapped ? m : p) * b.dst_inner);
    return 1;
}
// word offset of summand k of group g relative to `first`, limb m of the group's first member (merged inverse launches);
// k = member count: the Q-only extra summand
__device__ __forceinline__ long vi_member_offset(const NttBatch& b, int g, int k, int m, gcptr first) {
    kargptr kb = (kargptr)__builtin_amdgcn_kernarg_segment_ptr();
    if (k >= kb->vi_cnt[g]) return (long)(((u64)kb->vi_extra[g] - (u64)first) >> 3) + (long)m * b.src_inner;
    const unsigned mem = kb->vi_mem[g];
    return ((long)((mem >> (8 * k)) & 255u) - (long)(mem & 255u)) * b.src_outer;
}
// the same with products that arrive in parts (NttBatch::vi_parts): summand k >= 1 of the job whose first summand is `first` (limb m of the group's
// first member, or of member pm for a P slot).  Q slots (pm < 0): the members in turn, each followed by its further parts, then the Q-only extra.
__device__ __forceinline__ long vi_summand_offset(const NttBatch& b, int g, int pm, int k, int m, gcptr first) {
    kargptr kb = (kargptr)__builtin_amdgcn_kernarg_segment_ptr();
    const unsigned mem = kb->vi_mem[g];
    if (pm >= 0) {
        const int item = (int)((mem >> (8 * pm)) & 255u);
        return ((long)(vi_parts_of(kb, item) >> 8) + (k - 1) - (long)item) * b.src_outer;
    }
    const int cnt = kb->vi_cnt[g], item0 = (int)(mem & 255u);
    int kk = k;
    for (int j = 0; j < cnt; ++j) {
        const int item = (int)((mem >> (8 * j)) & 255u);
        const unsigned parts = vi_parts_of(kb, item);
        const int np_ = 1 + (int)(parts & 255u);
        if (kk < np_) return ((long)(kk == 0 ? item : (int)(parts >> 8) + kk - 1) - (long)item0) * b.src_outer;
        kk -= np_;
    }
    return (long)(((u64)kb->vi_extra[g] - (u64)first) >> 3) + (long)m * b.src_inner;
}

// ------------------------------------------------------------------ forward kernel
// phases: 0 = index bits n-1..n-5 (layout A), 1 = bits MIDTOP..5 (layout B), 2 = bits 4..0 (layout C)
// MODE 1: moduli with 34q < 2^63, no reduction inside; MODE 0: reduced every stage.
// DEC: fused gadget digit spread of Decompose (input limb re-read under a foreign modulus).
#ifdef MKHE_PHASE_TRACE
#define MKHE_STAMP(k) do { if (b.trace && (threadIdx.x & 63) == 0 && active) b.trace[((long)job * 16 + (threadIdx.x >> 6)) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MKHE_STAMP(k) do { } while (0)
#endif
// one limb (or half limb of a split launch): load, [digit reduction], three radix-32 phases, normalisation, store
template <int LOGN, int MODE, bool DEC>
__device__ __forceinline__ void ntt_fwd_job(const NttBatch& b, int job, bool active, u32* lds, int t) {
    using G = Geo<LOGN>;
#ifdef MKHE_PHASE_TRACE
    if (b.trace && (threadIdx.x & 63) == 0 && active) {
        u64* tw = b.trace + ((long)job * 16 + (threadIdx.x >> 6)) * 16;
        tw[12] = __builtin_amdgcn_s_memrealtime();
        tw[14] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
    }
#endif
    MKHE_STAMP(0);
    gcptr src; gptr dst; int m, outer;
    const int half = b.split ? (job & 1) : 0;
    job_pointers(b, b.split ? (job >> 1) : job, src, dst, m, outer);
    src += half * G::N; dst += half * G::N;
    const int root = b.split ? 2 + half : 1;          // twiddle index of a sub-transform group: root * m' + i'
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    gcptr psi = (gcptr)(b.psi + ((long)m * G::N << b.split));

    u64 x[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = src[posA<LOGN>(t, r)];
#ifdef MKHE_PHASE_TRACE
    if (b.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); MKHE_STAMP(1); }      // loads landed
#endif
    if constexpr (DEC) {
        // digit of a foreign modulus (Decompose, alpha = 1): bring it below 4q when needed
        int sm = m;
        if (b.reduce_src_mod_is_outer == 1) sm = outer;
        else if (b.reduce_src_mod_is_outer == 2) sm = ((kargptr)__builtin_amdgcn_kernarg_segment_ptr())->outer_mod[outer];
        const u64 qs = b.mods[sm].q << (b.src_lazy ? 2 : 0);       // bound of the digit values (< 2^63)
        if (qs > 4 * q) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                if constexpr (MODE == 1) x[r] = (u64)mont_mul_sd((i64)x[r], md.r1s, md.qs, ninv);     // signed, |.| < q
                else x[r] = mont_mul_sdu(x[r], md.r1s, md.qs, q, ninv);                                // [0,2q)
            }
        }
    }
#pragma unroll 1
    for (int ph = 0; ph < 3; ++ph) {
        int base = 16, prefix = 0, maxB = 4;
        if (ph == 1) { base = G::N >> 6; prefix = t >> 5; maxB = G::MIDB; }
        if (ph == 2) { base = G::N >> 1; prefix = t; }
        phase<MODE>(x, psi, base * root, prefix, maxB, q, q2, md.qs, ninv);
        MKHE_STAMP(2 + 2 * ph);                  // 2, 4, 6: end of the butterflies of phase ph
        if (ph == 0) {
            if constexpr (G::HAS_MID) {
                __syncthreads();     // every wave is done with the LDS of the previous limb
                exchange<LOGN, LA, LB, true>(x, lds, t);
            }
        } else if (ph == 1) exchange<LOGN, LB, LC, false>(x, lds, t);
        MKHE_STAMP(3 + 2 * ph);                  // 3, 5: end of the re-distribution after phase ph
    }
    // canonical output (lattigo: final BRedAdd)
    if constexpr (MODE == 1) {
        // MODE 1 values are signed with |x| < 4q + 15 * 0.65q < 14q
        if (!b.skip_norm) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const i64 y = mont_mul_sd((i64)x[r], md.r1s, md.qs, ninv);             // (-0.6q, 0.6q)
                x[r] = (u64)(y + ((y >> 63) & (i64)q));                               // canonical
            }
        } else {
            const i64 bias = (i64)(q << 4);                                            // same residue, positive: (2q, 30q)
#pragma unroll
            for (int r = 0; r < 32; ++r) x[r] = (u64)((i64)x[r] + bias);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 32; ++r) x[r] = csub(csub(x[r], q2), q);
    }
    MKHE_STAMP(7);                               // normalisation done
    // back to layout B for 256-B contiguous stores
    exchange<LOGN, LC, LB, false>(x, lds, t);
    MKHE_STAMP(8);
    if (active) {
#pragma unroll
        for (int r = 0; r < 32; ++r) dst[posB(t, r)] = x[r];
    }
    MKHE_STAMP(9);                               // stores issued
#ifdef MKHE_PHASE_TRACE
    if (b.trace && (threadIdx.x & 63) == 0 && active) b.trace[((long)job * 16 + (threadIdx.x >> 6)) * 16 + 13] = __builtin_amdgcn_s_memrealtime();
#endif
}

// MODE 2 (mixed launch): the modulus class is looked up per job (NttBatch::small_slots), so that the limbs of both classes
// share ONE persistent grid -- launched as two kernels the big-modulus class only gets the CUs when the other class's
// persistent workgroups exit, and then runs a ragged second round on a quarter of the chip.
template <int LOGN, int MODE, bool DEC>
__global__ void __launch_bounds__(Geo<LOGN>::BT) ntt_fwd_kernel(NttBatch b) {
    using G = Geo<LOGN>;
    extern __shared__ __attribute__((aligned(16))) u32 lds_all[];
    const int sub = G::LPB == 1 ? 0 : threadIdx.x / G::T, t = G::LPB == 1 ? threadIdx.x : threadIdx.x % G::T;
    u32* lds = lds_all + sub * lds_words<LOGN>();
    const int njobs = (b.nslots * b.nouter) << b.split;
#ifdef MKHE_R1_PRIO      // experiment (round 4): static wave priorities inside the only resident workgroup, so that one wave group's memory / LDS phases run under the other's butterflies
    {
        const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        if (MKHE_R1_PRIO == 2) { if (w < 8) __builtin_amdgcn_s_setprio(3); }
        else { const int g = w >> 2; if (g == 0) __builtin_amdgcn_s_setprio(3); else if (g == 1) __builtin_amdgcn_s_setprio(2); else if (g == 2) __builtin_amdgcn_s_setprio(1); }
    }
#endif
    // persistent workgroups: each one walks the job list with stride gridDim.x.  A wave that finishes its
    // part of a limb starts loading the next limb at once; the only workgroup-wide rendezvous are the
    // barriers around the cross-wave exchange.
#pragma unroll 1
    for (int jb = blockIdx.x * G::LPB; jb < njobs; jb += gridDim.x * G::LPB) {
        int job = jb + sub;
        const bool active = job < njobs;
        if (!active) job = njobs - 1;            // keep every lane in the barriers; results discarded
        if constexpr (G::LPB == 1) job = __builtin_amdgcn_readfirstlane(job);    // provably wave-uniform: modulus constants and pointers stay in SGPRs
        if constexpr (MODE == 2) {
            const int slot = (b.split ? (job >> 1) : job) / b.nouter;
            if ((b.small_slots >> slot) & 1) ntt_fwd_job<LOGN, 1, DEC>(b, job, active, lds, t);
            else ntt_fwd_job<LOGN, 0, DEC>(b, job, active, lds, t);
        } else ntt_fwd_job<LOGN, MODE, DEC>(b, job, active, lds, t);
    }
}

// ------------------------------------------------------------------ inverse kernel
// VI: merged launches (NttBatch::vi) -- a separate instantiation, the member sum at the load costs the plain kernel registers
template <int LOGN, bool VI>
__global__ void __launch_bounds__(Geo<LOGN>::BT) ntt_inv_kernel(NttBatch b) {
    using G = Geo<LOGN>;
    extern __shared__ __attribute__((aligned(16))) u32 lds_all[];
    const int sub = G::LPB == 1 ? 0 : threadIdx.x / G::T, t = G::LPB == 1 ? threadIdx.x : threadIdx.x % G::T;
    u32* lds = lds_all + sub * lds_words<LOGN>();
    const int njobs = (b.nslots * b.nouter) << b.split;
#pragma unroll 1
    for (int jb = blockIdx.x * G::LPB; jb < njobs; jb += gridDim.x * G::LPB) {
    int job = jb + sub;
    const bool active = job < njobs;
    if (!active) job = njobs - 1;
    if constexpr (G::LPB == 1) job = __builtin_amdgcn_readfirstlane(job);    // provably wave-uniform: modulus constants and pointers stay in SGPRs
    gcptr src; gptr dst; int m, outer;
    const int half = b.split ? (job & 1) : 0;
    int pm = -1;
    const int nsum_ = job_pointers<VI>(b, b.split ? (job >> 1) : job, src, dst, m, outer, &pm);
    const int nsum = VI ? nsum_ : 1;
    if (nsum == 0) {                             // merged launch: a P slot of a member this group does not have
        if constexpr (G::LPB == 1) continue;     // (the whole workgroup shares the job)
    }
    const bool exists = active && nsum != 0;
    src += half * G::N; dst += half * G::N;
    const int root = b.split ? 2 + half : 1;
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    gcptr psi = (gcptr)(b.psi + ((long)m * G::N << b.split));

    u64 x[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = src[posB(t, r)];
    if constexpr (VI) {
        // merged launch, Q slot: the canonical sum of this limb over the group's members
#pragma unroll 1
        for (int k = 1; k < nsum; ++k) {
            gcptr sk = src + vi_summand_offset(b, outer, pm, k, m, src - half * G::N);
#pragma unroll
            for (int r = 0; r < 32; ++r) x[r] = csub(x[r] + sk[posB(t, r)], q);
        }
    }
    if constexpr (G::HAS_MID) __syncthreads();   // other waves may still read this half-wave's LDS region (previous limb)
    exchange_xor<LOGN, LB, LC, false>(x, lds, t);
    // phases: 0 = index bits 0..4 (layout C), 1 = bits 5..MIDTOP (layout B); the top phase follows
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        int base = G::N >> 1, prefix = t, maxB = 4;
        if (ph == 1) { base = G::N >> 6; prefix = t >> 5; maxB = G::MIDB; }
        phase<2>(x, psi, base * root, prefix, maxB, q, q2, md.qs, ninv);
        if (ph == 0) exchange_xor<LOGN, LC, LB, false>(x, lds, t);
        else { if constexpr (G::HAS_MID) exchange_xor<LOGN, LB, LA, true>(x, lds, t); }
    }
    // top phase: index bits n-5 .. n-2, then the last stage with N^-1 folded in
    phase<2>(x, psi, 16 * root, 0, 3, q, q2, md.qs, ninv);
    // aux: 6 words per modulus = {N^-1 * R, psiinv[1] * N^-1 * R} for the whole transform, then the same pair with
    // psiinv[2 + half] for the two sub-transforms of a split limb (N = size of the whole limb in both cases)
    const int ax = 6 * m + (b.split ? 2 + 2 * half : 0);
    const u64 ninvR = b.aux[ax], w1n = b.aux[ax + 1];
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        u64 U = x[g], V = x[g + 16];
        u64 s = mont_mul_sdu(U + V, ninvR, md.qs, q, ninv);          // aux constants are in signed-split form
        u64 d = mont_mul_sdu(U + q2 - V, w1n, md.qs, q, ninv);
        x[g] = (b.lazy_out | b.split) ? s : csub(s, q);
        x[g + 16] = (b.lazy_out | b.split) ? d : csub(d, q);
    }
    if (exists) {
#pragma unroll
        for (int r = 0; r < 32; ++r) dst[posA<LOGN>(t, r)] = x[r];
    }
    }
}

// ------------------------------------------------------------------ N = 2^16: cross-half radix-2 passes
// The limb does not fit one workgroup's registers (64 coefficients per thread).  The first Cooley-Tukey stage
// (pairs (j, j + N/2), twiddle psi[1]) is a streaming pass src -> dst; the remaining 15 stages are two independent
// 2^15-point sub-transforms run in place by the register-resident kernel (NttBatch::split).  Mirror image for the
// inverse: sub-transforms first, then the last Gentleman-Sande stage (N^-1 already folded into the sub-transforms).
constexpr int SPLIT_THREADS = 256;
template <bool DEC>
__global__ void __launch_bounds__(SPLIT_THREADS) ntt_split_fwd_kernel(NttBatch b, int logN) {
    const int N = 1 << logN, H = N >> 1;
    gcptr src; gptr dst; int m, outer;
    job_pointers(b, blockIdx.y, src, dst, m, outer);
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const u64 w = b.psi[(long)m * N + 1];
    bool red = false;
    if constexpr (DEC) {
        int sm = m;
        if (b.reduce_src_mod_is_outer == 1) sm = outer;
        else if (b.reduce_src_mod_is_outer == 2) sm = ((kargptr)__builtin_amdgcn_kernarg_segment_ptr())->outer_mod[outer];
        const u64 qs = b.mods[sm].q << (b.src_lazy ? 2 : 0);
        red = qs > 4 * q;
    }
    for (int j = blockIdx.x * SPLIT_THREADS + threadIdx.x; j < H; j += gridDim.x * SPLIT_THREADS) {
        u64 U = src[j], V = src[j + H];
        if (red) { U = mont_mul_sdu(U, md.r1s, md.qs, q, ninv); V = mont_mul_sdu(V, md.r1s, md.qs, q, ninv); }
        else U = csub(U, q2);                                 // inputs < 4q
        const u64 Tm = mont_mul_sdu(V, w, md.qs, q, ninv);
        dst[j] = U + Tm;                                      // < 4q: what the sub-transforms accept
        dst[j + H] = U + (q2 - Tm);
    }
}
__global__ void __launch_bounds__(SPLIT_THREADS) ntt_split_inv_kernel(NttBatch b, int logN) {
    const int N = 1 << logN, H = N >> 1;
    gcptr src; gptr dst; int m, outer;
    if (!job_pointers<true>(b, blockIdx.y, src, dst, m, outer)) return;
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const u64 w = b.psi[(long)m * N + 1];
    for (int j = blockIdx.x * SPLIT_THREADS + threadIdx.x; j < H; j += gridDim.x * SPLIT_THREADS) {
        const u64 U = dst[j], V = dst[j + H];                 // [0,2q)
        u64 s = csub(U + V, q2);
        u64 d = mont_mul_sdu(U + q2 - V, w, md.qs, q, ninv);
        if (!b.lazy_out) { s = csub(s, q); d = csub(d, q); }
        dst[j] = s;
        dst[j + H] = d;
    }
}
// ------------------------------------------------------------------ low-latency sub-transforms (small launches)
// A launch with far fewer limbs than the chip has CUs is bound by the latency of ONE register-resident sub-transform:
// 32 coefficients per thread = 200+ dependent butterflies on one or two waves per SIMD.  For those launches the
// sub-transforms run on 2^13 coefficients with 8 per thread instead (1024 threads, 4 waves per SIMD, 52 butterflies
// each): radix-8 phases in registers, the data re-distributed through LDS (64-bit words, skewed) between phases, first
// phase straight from global memory, last phase straight to it.  N = 2^14: one streaming pass + 2 sub-transforms per
// limb; N = 2^15: two streaming passes + 4.  Same values in and out as the register-resident path.
constexpr int SM_LOGM = 13, SM_M = 1 << SM_LOGM, SM_T = 1024, SM_E = SM_M / SM_T;
constexpr int SM_LDS_WORDS = SM_M + (SM_M >> 3);
__device__ __forceinline__ int sm_pad(int p) { return p + ((p >> 4) << 1); }     // 16 consecutive words, then 2 words of skew

// One phase: NB stages S0 .. S0+NB-1 of the 2^13-point sub-transform on units of 2^NB coefficients p + a * gl.
// MODE 0 / 1: forward (Harvey / signed never-reduced), 2: inverse (stages in descending order).
// FIN 1: forward normalisation on the way out (canonical, or +16q when skip_norm); FIN 2: inverse, times N^-1.
// nsum > 1 (first inverse phase of a merged launch, NttBatch::vi): the input is the canonical sum of nsum limbs, gsrc + sum_off[k].
// LOGM: log2 of the sub-transform.  13: 1024 threads x 8 coefficients (N = 2^15: four per limb behind the radix-4 pass).  12 (round 3): 512 threads
// x 8 coefficients, 36 KB of LDS -- N = 2^14 limbs as FOUR sub-transforms behind the radix-4 pass instead of two of 2^13 points behind a radix-2
// pass: the launches of that ring (PN14QP439, the cnn) are a few dozen limbs, a kernel lasts as long as ONE workgroup, and a workgroup of half
// the size with half the work finishes sooner on twice as many CUs.
template <int LOGM> struct SmGeo { static constexpr int T = LOGM == 11 ? 256 : LOGM == 12 ? 512 : SM_T, E = (1 << LOGM) / T, LDSW = (1 << LOGM) + (1 << (LOGM - 3)); };
// PRE (the 2^12- and 2^11-point kernels): the twiddles of ALL phases were requested at the start of the kernel (sm_tw_load: they do not depend on
// the data; 28 words per thread), so that a phase does not begin with a global round trip behind its barrier -- these kernels last as long as one
// workgroup, and a workgroup's time is its phase structure, not its instruction count.
template <int S0, int NB, int LOGM>
__device__ __forceinline__ void sm_tw_load(gcptr psi, int root, int t, u64* tw) {
    constexpr int LGL = LOGM - S0 - NB, UPT = SmGeo<LOGM>::E >> NB, NE = 1 << NB;
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
        const int high = (k * SmGeo<LOGM>::T + t) >> LGL;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < (1 << i); ++j) tw[k * (NE - 1) + (1 << i) - 1 + j] = psi[(root << (S0 + i)) + (high << i) + j];
    }
}
template <int S0, int NB, bool FROM_G, bool TO_G, int MODE, int FIN, int LOGM = SM_LOGM, bool PRE = false>
__device__ __forceinline__ void sm_phase(gcptr gsrc, gptr gdst, u64* lds, gcptr psi, int root, int t, const Mod& md, u64 fin_c, int skip_norm,
                                         int nsum = 1, const long* sum_off = nullptr, const u64* tw = nullptr) {
    constexpr int LGL = LOGM - S0 - NB, GL = 1 << LGL, UPT = SmGeo<LOGM>::E >> NB, NE = 1 << NB;
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
#pragma unroll
    for (int k = 0; k < UPT; ++k) {
        const int u = k * SmGeo<LOGM>::T + t;
        const int low = u & (GL - 1), high = u >> LGL;
        const int p = (high << (LGL + NB)) | low;
        u64 x[NE];
#pragma unroll
        for (int a = 0; a < NE; ++a) x[a] = FROM_G ? gsrc[p + a * GL] : lds[sm_pad(p + a * GL)];
        if constexpr (FROM_G && MODE == 2) {
#pragma unroll
            for (int k = 1; k < VI_SUMS; ++k)
                if (k < nsum) {
#pragma unroll
                    for (int a = 0; a < NE; ++a) x[a] = csub(x[a] + gsrc[sum_off[k] + p + a * GL], q);
                }
        }
#pragma unroll
        for (int ii = 0; ii < NB; ++ii) {
            const int i = MODE == 2 ? NB - 1 - ii : ii;
            const int half = 1 << (NB - 1 - i);
#pragma unroll
            for (int a = 0; a < NE; ++a) {
                if (a & half) continue;
                const u64 w = PRE ? tw[k * (NE - 1) + (1 << i) - 1 + (a >> (NB - i))] : psi[(root << (S0 + i)) + (high << i) + (a >> (NB - i))];
                if constexpr (MODE == 2) bfly_inv(x[a], x[a + half], w, q, q2, md.qs, ninv);
                else if constexpr (MODE == 1) bfly_fwd_nr(x[a], x[a + half], w, md.qs, ninv);
                else bfly_fwd_cs(x[a], x[a + half], w, q, q2, md.qs, ninv);
            }
        }
#pragma unroll
        for (int a = 0; a < NE; ++a) {
            u64 v = x[a];
            if constexpr (FIN == 1) {
                if constexpr (MODE == 1) {
                    if (!skip_norm) { const i64 y = mont_mul_sd((i64)v, md.r1s, md.qs, ninv); v = (u64)(y + ((y >> 63) & (i64)q)); }
                    else v = (u64)((i64)v + (i64)(q << 4));
                } else v = csub(csub(v, q2), q);
            } else if constexpr (FIN == 2) v = mont_mul_sdu(v, fin_c, md.qs, q, ninv);          // [0,2q), N^-1 folded in
            if constexpr (TO_G) gdst[p + a * GL] = v; else lds[sm_pad(p + a * GL)] = v;
        }
    }
}

template <int MODE, int LOGM = SM_LOGM>
__global__ void __launch_bounds__(SmGeo<LOGM>::T) ntt_fwd_lds_kernel(NttBatch b, int d) {
    extern __shared__ __attribute__((aligned(16))) u64 sm_lds[];
    constexpr int M = 1 << LOGM;
    const int job = blockIdx.x, part = job & ((1 << d) - 1), t = threadIdx.x;
    gcptr src; gptr dst; int m, outer;
    job_pointers(b, job >> d, src, dst, m, outer);
    src += part * M; dst += part * M;
    const int root = (1 << d) + part;
    const Mod md = b.mods[m];
    gcptr psi = (gcptr)(b.psi + ((long)m * M << d));
    if constexpr (LOGM <= 12) {
        constexpr int NBL = LOGM == 12 ? 3 : 2, NL = (SmGeo<LOGM>::E >> NBL) * ((1 << NBL) - 1);
        u64 t0[7], t1[7], t2[7], t3[NL];
        sm_tw_load<0, 3, LOGM>(psi, root, t, t0); sm_tw_load<3, 3, LOGM>(psi, root, t, t1);
        sm_tw_load<6, 3, LOGM>(psi, root, t, t2); sm_tw_load<9, NBL, LOGM>(psi, root, t, t3);
        sm_phase<0, 3, true, false, MODE, 0, LOGM, true>(src, dst, sm_lds, psi, root, t, md, 0, 0, 1, nullptr, t0);
        __syncthreads();
        sm_phase<3, 3, false, false, MODE, 0, LOGM, true>(src, dst, sm_lds, psi, root, t, md, 0, 0, 1, nullptr, t1);
        __syncthreads();
        sm_phase<6, 3, false, false, MODE, 0, LOGM, true>(src, dst, sm_lds, psi, root, t, md, 0, 0, 1, nullptr, t2);
        __syncthreads();
        sm_phase<9, NBL, false, true, MODE, 1, LOGM, true>(src, dst, sm_lds, psi, root, t, md, 0, b.skip_norm, 1, nullptr, t3);
    } else {
        sm_phase<0, 3, true, false, MODE, 0, LOGM>(src, dst, sm_lds, psi, root, t, md, 0, 0);
        __syncthreads();
        sm_phase<3, 3, false, false, MODE, 0, LOGM>(src, dst, sm_lds, psi, root, t, md, 0, 0);
        __syncthreads();
        sm_phase<6, 3, false, false, MODE, 0, LOGM>(src, dst, sm_lds, psi, root, t, md, 0, 0);
        __syncthreads();
        sm_phase<9, 3, false, false, MODE, 0, LOGM>(src, dst, sm_lds, psi, root, t, md, 0, 0);
        __syncthreads();
        sm_phase<12, 1, false, true, MODE, 1, LOGM>(src, dst, sm_lds, psi, root, t, md, 0, b.skip_norm);
    }
}

// the 2^12- and 2^11-point inverse sub-transforms (see SmGeo): four phases, 512 / 256 threads
template <int LOGM>
__global__ void __launch_bounds__(SmGeo<LOGM>::T) ntt_inv_ldsS_kernel(NttBatch b, int d) {
    extern __shared__ __attribute__((aligned(16))) u64 sm_lds[];
    constexpr int M = 1 << LOGM;
    const int job = blockIdx.x, part = job & ((1 << d) - 1), t = threadIdx.x;
    gcptr src; gptr dst; int m, outer;
    int pm = -1;
    const int nsum = job_pointers<true>(b, job >> d, src, dst, m, outer, &pm);
    if (nsum == 0) return;
    long sum_off[VI_SUMS] = {};
#pragma unroll
    for (int k = 1; k < VI_SUMS; ++k) if (k < nsum) sum_off[k] = vi_summand_offset(b, outer, pm, k, m, src);
    src += part * M; dst += part * M;
    const int root = (1 << d) + part;
    const Mod md = b.mods[m];
    gcptr psi = (gcptr)(b.psi + ((long)m * M << d));
    const u64 ninvR = b.aux[6 * m];
    constexpr int NBL = LOGM == 12 ? 3 : 2, NL = (SmGeo<LOGM>::E >> NBL) * ((1 << NBL) - 1);
    u64 t0[7], t1[7], t2[7], t3[NL];
    sm_tw_load<9, NBL, LOGM>(psi, root, t, t3); sm_tw_load<6, 3, LOGM>(psi, root, t, t2);
    sm_tw_load<3, 3, LOGM>(psi, root, t, t1); sm_tw_load<0, 3, LOGM>(psi, root, t, t0);
    sm_phase<9, NBL, true, false, 2, 0, LOGM, true>(src, dst, sm_lds, psi, root, t, md, 0, 0, nsum, sum_off, t3);
    __syncthreads();
    sm_phase<6, 3, false, false, 2, 0, LOGM, true>(src, dst, sm_lds, psi, root, t, md, 0, 0, 1, nullptr, t2);
    __syncthreads();
    sm_phase<3, 3, false, false, 2, 0, LOGM, true>(src, dst, sm_lds, psi, root, t, md, 0, 0, 1, nullptr, t1);
    __syncthreads();
    sm_phase<0, 3, false, true, 2, 2, LOGM, true>(src, dst, sm_lds, psi, root, t, md, ninvR, 0, 1, nullptr, t0);
}

// cross-block radix-2 passes below the first one (level L >= 1: blocks of N >> L coefficients, twiddle index 2^L + block),
// in place on dst; forward: values < 4q in and out; inverse: [0,2q) in and out (N^-1 already folded in)
__global__ void __launch_bounds__(SPLIT_THREADS) ntt_pass_fwd_kernel(NttBatch b, int logN, int L) {
    const int N = 1 << logN, G = N >> (L + 1);
    gcptr src; gptr dst; int m, outer;
    job_pointers(b, blockIdx.y, src, dst, m, outer);
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    for (int j = blockIdx.x * SPLIT_THREADS + threadIdx.x; j < (N >> 1); j += gridDim.x * SPLIT_THREADS) {
        const int blk = j >> (logN - L - 1), off = j - blk * G, i0 = blk * 2 * G + off;         // G = 2^(logN - L - 1)
        const u64 w = b.psi[(long)m * N + (1 << L) + blk];
        const u64 U = csub(dst[i0], q2), V = dst[i0 + G];
        const u64 Tm = mont_mul_sdu(V, w, md.qs, q, ninv);
        dst[i0] = U + Tm;
        dst[i0 + G] = U + (q2 - Tm);
    }
}
__global__ void __launch_bounds__(SPLIT_THREADS) ntt_pass_inv_kernel(NttBatch b, int logN, int L) {
    const int N = 1 << logN, G = N >> (L + 1);
    gcptr src; gptr dst; int m, outer;
    if (!job_pointers<true>(b, blockIdx.y, src, dst, m, outer)) return;
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    for (int j = blockIdx.x * SPLIT_THREADS + threadIdx.x; j < (N >> 1); j += gridDim.x * SPLIT_THREADS) {
        const int blk = j >> (logN - L - 1), off = j - blk * G, i0 = blk * 2 * G + off;         // G = 2^(logN - L - 1)
        const u64 w = b.psi[(long)m * N + (1 << L) + blk];
        const u64 U = dst[i0], V = dst[i0 + G];
        dst[i0] = csub(U + V, q2);
        dst[i0 + G] = mont_mul_sdu(U + q2 - V, w, md.qs, q, ninv);
    }
}

// The two outermost stages in ONE streaming pass (low-latency path at N = 2^15: four 2^13-point sub-transforms per limb):
// every thread owns the coefficients j, j + N/4, j + N/2, j + 3N/4.  Forward: src -> dst (optionally with the Decompose
// reduction), values < 4q out; inverse: in place on dst, [0,2q) in (N^-1 folded in by the sub-transforms), canonical or
// lazy out.
template <bool DEC>
__global__ void __launch_bounds__(SPLIT_THREADS) ntt_pass4_fwd_kernel(NttBatch b, int logN) {
    const int N = 1 << logN, Q = N >> 2;
    gcptr src; gptr dst; int m, outer;
    job_pointers(b, blockIdx.y, src, dst, m, outer);
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const u64 w1 = b.psi[(long)m * N + 1], w2 = b.psi[(long)m * N + 2], w3 = b.psi[(long)m * N + 3];
    bool red = false;
    if constexpr (DEC) {
        int sm = m;
        if (b.reduce_src_mod_is_outer == 1) sm = outer;
        else if (b.reduce_src_mod_is_outer == 2) sm = ((kargptr)__builtin_amdgcn_kernarg_segment_ptr())->outer_mod[outer];
        const u64 qs = b.mods[sm].q << (b.src_lazy ? 2 : 0);
        red = qs > 4 * q;
    }
    for (int j = blockIdx.x * SPLIT_THREADS + threadIdx.x; j < Q; j += gridDim.x * SPLIT_THREADS) {
        u64 a0 = src[j], a1 = src[j + Q], a2 = src[j + 2 * Q], a3 = src[j + 3 * Q];
        if (red) {
            a0 = mont_mul_sdu(a0, md.r1s, md.qs, q, ninv); a1 = mont_mul_sdu(a1, md.r1s, md.qs, q, ninv);
            a2 = mont_mul_sdu(a2, md.r1s, md.qs, q, ninv); a3 = mont_mul_sdu(a3, md.r1s, md.qs, q, ninv);
        } else { a0 = csub(a0, q2); a1 = csub(a1, q2); }
        u64 T = mont_mul_sdu(a2, w1, md.qs, q, ninv);
        const u64 b0 = a0 + T, b2 = a0 + (q2 - T);
        T = mont_mul_sdu(a3, w1, md.qs, q, ninv);
        const u64 b1 = a1 + T, b3 = a1 + (q2 - T);
        u64 u = csub(b0, q2);
        T = mont_mul_sdu(b1, w2, md.qs, q, ninv);
        dst[j] = u + T; dst[j + Q] = u + (q2 - T);
        u = csub(b2, q2);
        T = mont_mul_sdu(b3, w3, md.qs, q, ninv);
        dst[j + 2 * Q] = u + T; dst[j + 3 * Q] = u + (q2 - T);
    }
}
__global__ void __launch_bounds__(SPLIT_THREADS) ntt_pass4_inv_kernel(NttBatch b, int logN) {
    const int N = 1 << logN, Q = N >> 2;
    gcptr src; gptr dst; int m, outer;
    if (!job_pointers<true>(b, blockIdx.y, src, dst, m, outer)) return;
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const u64 w1 = b.psi[(long)m * N + 1], w2 = b.psi[(long)m * N + 2], w3 = b.psi[(long)m * N + 3];
    for (int j = blockIdx.x * SPLIT_THREADS + threadIdx.x; j < Q; j += gridDim.x * SPLIT_THREADS) {
        const u64 a0 = dst[j], a1 = dst[j + Q], a2 = dst[j + 2 * Q], a3 = dst[j + 3 * Q];
        const u64 s0 = csub(a0 + a1, q2), d0 = mont_mul_sdu(a0 + q2 - a1, w2, md.qs, q, ninv);
        const u64 s1 = csub(a2 + a3, q2), d1 = mont_mul_sdu(a2 + q2 - a3, w3, md.qs, q, ninv);
        u64 r0 = csub(s0 + s1, q2), r2 = mont_mul_sdu(s0 + q2 - s1, w1, md.qs, q, ninv);
        u64 r1 = csub(d0 + d1, q2), r3 = mont_mul_sdu(d0 + q2 - d1, w1, md.qs, q, ninv);
        if (!b.lazy_out) { r0 = csub(r0, q); r1 = csub(r1, q); r2 = csub(r2, q); r3 = csub(r3, q); }
        dst[j] = r0; dst[j + Q] = r1; dst[j + 2 * Q] = r2; dst[j + 3 * Q] = r3;
    }
}

// The THREE outermost stages in one streaming pass (round 3; small launches at N = 2^15: eight 2^12-point sub-transforms per limb): every thread owns
// the coefficients j + k N/8, k = 0..7.  Forward: src -> dst (optionally with the Decompose reduction), values < 4q out; inverse: in place on dst,
// [0,2q) in (N^-1 folded in by the sub-transforms), canonical or lazy out.  Sub-transform p occupies words [p N/8, (p + 1) N/8), twiddle root 8 + p.
template <bool DEC>
__global__ void __launch_bounds__(SPLIT_THREADS) ntt_pass8_fwd_kernel(NttBatch b, int logN) {
    const int N = 1 << logN, E = N >> 3;
    gcptr src; gptr dst; int m, outer;
    job_pointers(b, blockIdx.y, src, dst, m, outer);
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    u64 w[8];
#pragma unroll
    for (int i = 1; i < 8; ++i) w[i] = b.psi[(long)m * N + i];
    bool red = false;
    if constexpr (DEC) {
        int sm = m;
        if (b.reduce_src_mod_is_outer == 1) sm = outer;
        else if (b.reduce_src_mod_is_outer == 2) sm = ((kargptr)__builtin_amdgcn_kernarg_segment_ptr())->outer_mod[outer];
        const u64 qs = b.mods[sm].q << (b.src_lazy ? 2 : 0);
        red = qs > 4 * q;
    }
    auto bf = [&](u64& x, u64& y, u64 tw) {
        const u64 U = csub(x, q2), T = mont_mul_sdu(y, tw, md.qs, q, ninv);
        x = U + T; y = U + (q2 - T);
    };
    for (int j = blockIdx.x * SPLIT_THREADS + threadIdx.x; j < E; j += gridDim.x * SPLIT_THREADS) {
        u64 a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = src[j + k * E];
        if (red) {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = mont_mul_sdu(a[k], md.r1s, md.qs, q, ninv);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) bf(a[k], a[k + 4], w[1]);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 2; ++k) bf(a[4 * h + k], a[4 * h + k + 2], w[2 + h]);
#pragma unroll
        for (int c = 0; c < 4; ++c) bf(a[2 * c], a[2 * c + 1], w[4 + c]);
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[j + k * E] = a[k];
    }
}
__global__ void __launch_bounds__(SPLIT_THREADS) ntt_pass8_inv_kernel(NttBatch b, int logN) {
    const int N = 1 << logN, E = N >> 3;
    gcptr src; gptr dst; int m, outer;
    const int nsum = job_pointers<true>(b, blockIdx.y, src, dst, m, outer);
    if (!nsum) return;
    const Mod md = b.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    u64 w[8];
#pragma unroll
    for (int i = 1; i < 8; ++i) w[i] = b.psi[(long)m * N + i];
    // (sum_in_cross: the sub-transforms ran per member, inside ext_fused_lds_kernel -- the members of the destination, all in [0, 2q), meet here)
    long moff[VI_MAX] = {};
    const int ns = b.sum_in_cross ? nsum : 1;
#pragma unroll
    for (int k = 1; k < VI_MAX; ++k) if (k < ns) moff[k] = vi_member_offset(b, outer, k, m, src);
    auto gs = [&](u64& x, u64& y, u64 tw) {
        const u64 s_ = csub(x + y, q2), d_ = mont_mul_sdu(x + q2 - y, tw, md.qs, q, ninv);
        x = s_; y = d_;
    };
    for (int j = blockIdx.x * SPLIT_THREADS + threadIdx.x; j < E; j += gridDim.x * SPLIT_THREADS) {
        u64 a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = dst[j + k * E];
#pragma unroll
        for (int mm = 1; mm < VI_MAX; ++mm)
            if (mm < ns) {
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] = csub(a[k] + dst[moff[mm] + j + k * E], q2);
            }
#pragma unroll
        for (int c = 0; c < 4; ++c) gs(a[2 * c], a[2 * c + 1], w[4 + c]);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 2; ++k) gs(a[4 * h + k], a[4 * h + k + 2], w[2 + h]);
#pragma unroll
        for (int k = 0; k < 4; ++k) gs(a[k], a[k + 4], w[1]);
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[j + k * E] = b.lazy_out ? a[k] : csub(a[k], q);
    }
}

// ---- forward sub-transforms + inner products in one kernel (see ntt_kernels.h, ExtFusedArgs)
template <int NG, int MODE, bool INV>
__device__ __forceinline__ void extf_body(const ExtFusedArgs& a, u64* lds, const Mod& md, int m, int part, int v) {
    constexpr int LOGM = 11, M = 1 << LOGM, TALL = NG * 256, CPT = M / TALL;
    const int g = threadIdx.x >> 8, t = threadIdx.x & 255;
    const int d = a.logN - LOGM, root = (1 << d) + part;
    gcptr psi = (gcptr)(a.psi + (long)m * a.N);
    u64* region = lds + g * SmGeo<LOGM>::LDSW;
    constexpr int NL = (SmGeo<LOGM>::E >> 2) * 3;
    u64 t0[7], t1[7], t2[7], t3[NL];
    sm_tw_load<0, 3, LOGM>(psi, root, t, t0); sm_tw_load<3, 3, LOGM>(psi, root, t, t1);
    sm_tw_load<6, 3, LOGM>(psi, root, t, t2); sm_tw_load<9, 2, LOGM>(psi, root, t, t3);
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const long limb = (long)m * a.N + (long)part * M;
    const int nk = a.nk[v];
    gcptr st = (gcptr)a.stage[v];
    gcptr k0 = (gcptr)a.bg[v][0], k1 = (gcptr)a.bg[v][1];
    u64 acc0[CPT], acc1[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) { acc0[j] = 0; acc1[j] = 0; }
    // (PF) the source of the NEXT round is requested while this one is transformed (first phase: thread t owns the words t + 256 a of its block)
    constexpr bool PF = true, PFS = true;   // the keys of a round and the source of the next one are requested ahead (measured: 2.14 -> 1.99 ms per cnn inference with, 2.10 without)
    static_assert(NG == 1 || NG == 2, "three and four groups (768 / 1024 threads) spill and were measured slower");
    static_assert(!INV || NG == 2, "the inverse sub-transforms are the work of two groups");
    u64 xn[8];
    if constexpr (PFS) {
        if (g < a.nb) {
#pragma unroll
            for (int e = 0; e < 8; ++e) xn[e] = st[(long)g * a.digit_stride + limb + t + 256 * e];
        }
    }
    for (int r = 0; r * NG < a.nb; ++r) {
        const int dg = r * NG + g;
        const bool active = dg < a.nb;
        gcptr src = st + (long)dg * a.digit_stride + limb;
        // (PF) the keys of the round are requested before the transform: the products do not wait for a second global round trip behind the last barrier
        u64 kv0[PF ? NG : 1][CPT], kv1[PF ? NG : 1][CPT];
        if constexpr (PF) {
#pragma unroll
            for (int dd = 0; dd < NG; ++dd) {
                const int dig = r * NG + dd;
                const long ko = (long)(dig < a.nb ? dig : 0) * a.digit_stride + limb;
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    const int pos = (int)threadIdx.x + j * TALL;
                    kv0[dd][j] = k0[ko + pos];
                    kv1[dd][j] = nk > 1 ? k1[ko + pos] : 0;
                }
            }
        }
        if constexpr (PFS) {
            if (active) {
#pragma unroll
                for (int e = 0; e < 8; ++e) region[sm_pad(t + 256 * e)] = xn[e];          // (read back by the same thread: no barrier)
                sm_phase<0, 3, false, false, MODE, 0, LOGM, true>(src, nullptr, region, psi, root, t, md, 0, 0, 1, nullptr, t0);
            }
            if (dg + NG < a.nb) {
#pragma unroll
                for (int e = 0; e < 8; ++e) xn[e] = st[(long)(dg + NG) * a.digit_stride + limb + t + 256 * e];
            }
        } else if (active) sm_phase<0, 3, true, false, MODE, 0, LOGM, true>(src, nullptr, region, psi, root, t, md, 0, 0, 1, nullptr, t0);
        __syncthreads();
        if (active) sm_phase<3, 3, false, false, MODE, 0, LOGM, true>(src, nullptr, region, psi, root, t, md, 0, 0, 1, nullptr, t1);
        __syncthreads();
        if (active) sm_phase<6, 3, false, false, MODE, 0, LOGM, true>(src, nullptr, region, psi, root, t, md, 0, 0, 1, nullptr, t2);
        __syncthreads();
        if (active) sm_phase<9, 2, false, false, MODE, 1, LOGM, true>(src, nullptr, region, psi, root, t, md, 0, 1, 1, nullptr, t3);   // (skip_norm: an engine-internal form)
        __syncthreads();
        // products: this thread's CPT coefficients of every digit of the round
#pragma unroll
        for (int dd = 0; dd < NG; ++dd) {
            const int dig = r * NG + dd;
            if (dig < a.nb) {
                const long ko = (long)dig * a.digit_stride + limb;
#pragma unroll
                for (int j = 0; j < CPT; ++j) {
                    const int pos = (int)threadIdx.x + j * TALL;
                    const u64 h = lds[dd * SmGeo<LOGM>::LDSW + sm_pad(pos)];
                    const u64 g0 = PF ? kv0[PF ? dd : 0][j] : k0[ko + pos];
                    acc0[j] = csub(acc0[j] + mont_mul_lazy(g0, h, q, ninv), q2);
                    if (nk > 1) { const u64 g1 = PF ? kv1[PF ? dd : 0][j] : k1[ko + pos]; acc1[j] = csub(acc1[j] + mont_mul_lazy(g1, h, q, ninv), q2); }
                }
            }
        }
        __syncthreads();
    }
    if constexpr (INV) {
        // ... and the inverse sub-transform of each product by one group: what ntt_inv_ldsS_kernel<11> does with the stored product (members of a
        // merged destination are then summed by the cross pass, ntt_pass8_inv_kernel with NttBatch::sum_in_cross -- the transform is linear)
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int pos = (int)threadIdx.x + j * TALL;
            lds[sm_pad(pos)] = csub(acc0[j], q);
            if (nk > 1) lds[SmGeo<LOGM>::LDSW + sm_pad(pos)] = csub(acc1[j], q);
        }
        __syncthreads();
        const bool work = g < nk;
        gcptr psii = (gcptr)(a.psiinv + (long)m * a.N);
        const u64 ninvR = a.aux[6 * m];
        u64 i0[7], i1[7], i2[7], i3[NL];
        if (work) {
            sm_tw_load<9, 2, LOGM>(psii, root, t, i3); sm_tw_load<6, 3, LOGM>(psii, root, t, i2);
            sm_tw_load<3, 3, LOGM>(psii, root, t, i1); sm_tw_load<0, 3, LOGM>(psii, root, t, i0);
            sm_phase<9, 2, false, false, 2, 0, LOGM, true>(nullptr, nullptr, region, psii, root, t, md, 0, 0, 1, nullptr, i3);
        }
        __syncthreads();
        if (work) sm_phase<6, 3, false, false, 2, 0, LOGM, true>(nullptr, nullptr, region, psii, root, t, md, 0, 0, 1, nullptr, i2);
        __syncthreads();
        if (work) sm_phase<3, 3, false, false, 2, 0, LOGM, true>(nullptr, nullptr, region, psii, root, t, md, 0, 0, 1, nullptr, i1);
        __syncthreads();
        if (work) sm_phase<0, 3, false, true, 2, 2, LOGM, true>(nullptr, (gptr)(a.out[v][g] + limb), region, psii, root, t, md, ninvR, 0, 1, nullptr, i0);
    } else {
    u64* o0 = a.out[v][0] + limb; u64* o1 = a.out[v][1];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int pos = (int)threadIdx.x + j * TALL;
        o0[pos] = csub(acc0[j], q);
        if (nk > 1) o1[limb + pos] = csub(acc1[j], q);
    }
    }
}
template <int NG, bool INV>
__global__ void __launch_bounds__(NG * 256) ext_fused_lds_kernel(ExtFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) u64 sm_lds[];
    const int part = blockIdx.x, m = a.map[blockIdx.y], v = blockIdx.z;
    const Mod md = a.mods[m];
    if ((a.small_mask >> m) & 1) extf_body<NG, 1, INV>(a, sm_lds, md, m, part, v);
    else extf_body<NG, 0, INV>(a, sm_lds, md, m, part, v);
}

static NttBatch in_place_of_dst(const NttBatch& b) {
    NttBatch c = b;
    c.src = b.dst; c.src_outer = b.dst_outer; c.src_inner = b.dst_inner; c.src_mapped = b.dst_mapped;
    for (int i = 0; i < NTT_MAX_ITEMS; ++i) c.src_items[i] = b.dst_items[i];
    c.reduce_in = 0; c.split = 1;
    return c;
}

// ------------------------------------------------------------------ launchers
// Launch state is kept PER DEVICE (a process may hold contexts on several GPUs: the function attribute has to be set and the
// persistent grid sized on each of them) and behind a mutex (contexts may be driven from different host threads).
namespace {
struct LaunchState { std::mutex mu; int resident[64] = {}; bool attr[64] = {}; };
int current_device() { int dev = 0; (void)hipGetDevice(&dev); return dev & 63; }
}
// workgroups that are co-resident on the whole chip for this kernel (persistent grid size)
static int resident_blocks(const void* fn, int threads, size_t lds) {
#ifdef MKHE_SWITCHES
    if (const char* e = getenv("MKHE_NTT_GRID")) { if (e[0] == 'f') return 1 << 30; }     // "full": one workgroup per limb (A/B testing)
#endif
    int dev = 0, cus = 256, per = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, fn, threads, lds) != hipSuccess || per < 1) per = 1;
    return cus * per;
}
template <int LOGN, int MODE, bool DEC> static void launch_fwd_t(const NttBatch& b, hipStream_t st) {
    using G = Geo<LOGN>;
    static LaunchState ls;
    const size_t lds = (size_t)G::LPB * lds_words<LOGN>() * sizeof(u32);
    int resident;
    {
        const int dev = current_device();
        std::lock_guard<std::mutex> g(ls.mu);
        if (!ls.attr[dev]) { (void)hipFuncSetAttribute((const void*)ntt_fwd_kernel<LOGN, MODE, DEC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); ls.attr[dev] = true; }
        if (!ls.resident[dev]) ls.resident[dev] = resident_blocks((const void*)ntt_fwd_kernel<LOGN, MODE, DEC>, G::BT, lds);
        resident = ls.resident[dev];
    }
    const int need = (((b.nslots * b.nouter) << b.split) + G::LPB - 1) / G::LPB;
    const int blocks = need < resident ? need : resident;
    hipLaunchKernelGGL((ntt_fwd_kernel<LOGN, MODE, DEC>), dim3(blocks), dim3(G::BT), lds, st, b);
}
template <int LOGN, bool VI> static void launch_inv_tv(const NttBatch& b, hipStream_t st) {
    using G = Geo<LOGN>;
    static LaunchState ls;
    const size_t lds = (size_t)G::LPB * lds_words<LOGN>() * sizeof(u32);
    int resident;
    {
        const int dev = current_device();
        std::lock_guard<std::mutex> g(ls.mu);
        if (!ls.attr[dev]) { (void)hipFuncSetAttribute((const void*)ntt_inv_kernel<LOGN, VI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); ls.attr[dev] = true; }
        if (!ls.resident[dev]) ls.resident[dev] = resident_blocks((const void*)ntt_inv_kernel<LOGN, VI>, G::BT, lds);
        resident = ls.resident[dev];
    }
    const int need = (((b.nslots * b.nouter) << b.split) + G::LPB - 1) / G::LPB;
    const int blocks = need < resident ? need : resident;
    hipLaunchKernelGGL((ntt_inv_kernel<LOGN, VI>), dim3(blocks), dim3(G::BT), lds, st, b);
}
template <int LOGN> static void launch_inv_t(const NttBatch& b, hipStream_t st) {
    if (b.vi) launch_inv_tv<LOGN, true>(b, st); else launch_inv_tv<LOGN, false>(b, st);
}

template <int MODE, bool DEC> static void launch_fwd_mode(int logN, const NttBatch& b, hipStream_t st) {
    switch (logN) {
        case 10: launch_fwd_t<10, MODE, DEC>(b, st); break;
        case 11: launch_fwd_t<11, MODE, DEC>(b, st); break;
        case 12: launch_fwd_t<12, MODE, DEC>(b, st); break;
        case 13: launch_fwd_t<13, MODE, DEC>(b, st); break;
        case 14: launch_fwd_t<14, MODE, DEC>(b, st); break;
        case 15: launch_fwd_t<15, MODE, DEC>(b, st); break;
        default: break;
    }
}

int split_ntt_fwd(const NttBatch& b, const unsigned char* small_q, NttBatch out[2]) {
    int n = 0;
    for (int cls = 1; cls >= 0; --cls) {
        NttBatch c = b;
        c.nslots = 0;
        for (int s = 0; s < b.nslots; ++s)
            if ((small_q[b.mod[s]] != 0) == (cls == 1)) { c.mod[c.nslots] = b.mod[s]; c.pos[c.nslots] = b.pos[s]; ++c.nslots; }
        c.lazy_out = cls;          // carries the class to launch_ntt_fwd_class (field unused by the forward kernels)
        if (c.nslots) out[n++] = c;
    }
    return n;
}
// A limb is one workgroup's work for 60-70 us whatever the batch size, so a launch with fewer limbs than CUs is latency
// bound at half the chip idle.  Such launches (and every N = 2^16 launch) run split: cross-half radix-2 pass + two
// half-size sub-transforms per limb (2 workgroups per CU fit), which nearly halves the latency of the small launches
// on the critical path (tensor / ExternalProduct inverse NTTs).
static bool use_split(int logN, const NttBatch& b) {
    if (logN == 16) return true;
    if (logN < 13) return false;
    const int limbs = b.vi ? b.vi_jobs : b.nslots * b.nouter;                // merged launches: the jobs that exist
    return limbs <= 128;        // at most one sub-transform workgroup per CU (256 CUs)
}
// depth of the low-latency path for this launch: 0 = register-resident sub-transforms, d >= 1 = 2^d LDS sub-transforms of
// 2^13 coefficients per limb after d streaming passes.  MKHE_NTT_LDS=0 switches it off (A/B tests).
static int lds_depth(int logN, const NttBatch& b) {
    static const int on = MKHE_AB_INT("MKHE_NTT_LDS", 1);
    if (!on || b.prestaged) return 0;
    if (logN != 14 && logN != 15) return 0;
    // N = 2^14: four 2^12-point sub-transforms per limb; N = 2^15: eight of them behind the radix-8 pass (the 2^13-point forms lost in rounds 3 and 4)
    {
        static LaunchState ls12;
        const int dev = current_device();
        std::lock_guard<std::mutex> g(ls12.mu);
        if (!ls12.attr[dev]) {
            const int lds = SmGeo<12>::LDSW * (int)sizeof(u64);
            (void)hipFuncSetAttribute((const void*)ntt_fwd_lds_kernel<0, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute((const void*)ntt_fwd_lds_kernel<1, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            (void)hipFuncSetAttribute((const void*)ntt_inv_ldsS_kernel<12>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            ls12.attr[dev] = true;
        }
        // N = 2^14 launches that would not even give every CU one 2^12-point workgroup: eight 2^11-point sub-transforms per limb (256 threads)
        const int limbs = b.vi ? b.vi_jobs : b.nslots * b.nouter;
        if (logN == 14 && limbs < 64) return 3;
        return logN - 12;
    }
}
bool ntt_fwd_mixed_ok(int logN, const NttBatch& b, const unsigned char* small_q) {
    if (logN != 15 || !b.reduce_in || b.split || b.nslots > 64 || b.nslots * b.nouter <= 512) return false;
    int nsmall = 0;
    for (int s = 0; s < b.nslots; ++s) if (small_q[b.mod[s]]) ++nsmall;
    return nsmall != 0 && nsmall != b.nslots;               // a single class: the specialised kernel
}
void launch_ntt_fwd_mixed(int logN, const NttBatch& b, const unsigned char* small_q, hipStream_t st) {
    (void)logN;
    NttBatch c = b;
    // slots reordered: the big-modulus limbs (the longer jobs) first, so that a ragged last round holds the cheap kind
    c.small_slots = 0; c.nslots = 0;
    for (int cls = 0; cls < 2; ++cls)
        for (int s = 0; s < b.nslots; ++s)
            if ((small_q[b.mod[s]] != 0) == (cls == 1)) {
                c.mod[c.nslots] = b.mod[s]; c.pos[c.nslots] = b.pos[s];
                if (cls) c.small_slots |= 1ull << c.nslots;
                ++c.nslots;
            }
    launch_fwd_t<15, 2, true>(c, st);
}
// would every class part of this prestaged N = 2^16 launch run on the H16 sub-transform kernel?  (only then may the producer stage its
// output in a separate buffer: NttBatch::prestaged_oop)
bool ntt_fwd_prestaged_oop_ok(int logN, const NttBatch& b, const unsigned char* small_q) {
    if (logN != 16 || !b.prestaged) return false;
    NttBatch part[2];
    const int n = split_ntt_fwd(b, small_q, part);
    for (int i = 0; i < n; ++i) {
        NttBatch o = part[i]; o.reduce_in = 0; o.split = 1;
        if (!use_split(logN, part[i]) || lds_depth(logN, part[i]) != 0 || !ntt16_split_ok(o)) return false;
    }
    return n > 0;
}
void launch_ntt_fwd_class(int logN, const NttBatch& b, hipStream_t st) {
    if (b.nslots <= 0 || b.nouter <= 0) return;
    const bool small = b.lazy_out != 0;
    if (b.prestaged == 2) {
        // N = 2^16, both cross stages applied by the producer (decomp_spread_kernel<2>): four one-pass 2^14-point sub-transforms per limb on
        // the H16 kernel, in place on dst or from the producer's staging buffer (the caller has asked ntt_fwd_prestaged_oop_ok)
        NttBatch o = b.prestaged_oop ? b : in_place_of_dst(b);
        o.reduce_in = 0; o.split = 2;
        if (logN != 16 || !ntt16_split_ok(o)) throw std::runtime_error("mkhe: internal: radix-4 prestaged launch outside the H16 path");
        launch_ntt16_fwd_split(o, small, st);
        return;
    }
    if (logN == 16 && !b.prestaged) {
        // N = 2^16 without a fused producer (the tensor-step inputs, plain NTTs): both cross stages as ONE streaming pass, then the four one-pass
        // 2^14-point sub-transforms per limb on the H16 kernel -- instead of a radix-2 pass and two-pass 2^15-point halves (MKHE_NTT16_RADIX4=0)
        static const int r4 = MKHE_AB_INT("MKHE_NTT16_RADIX4", 1);
        NttBatch o = in_place_of_dst(b);
        o.reduce_in = 0; o.split = 2;
        if (r4 && ntt16_split_ok(o)) {
            const dim3 grid(32, b.nslots * b.nouter);
            if (b.reduce_in) hipLaunchKernelGGL(ntt_pass4_fwd_kernel<true>, grid, dim3(SPLIT_THREADS), 0, st, b, logN);
            else hipLaunchKernelGGL(ntt_pass4_fwd_kernel<false>, grid, dim3(SPLIT_THREADS), 0, st, b, logN);
            launch_ntt16_fwd_split(o, small, st);
            return;
        }
    }
    if (use_split(logN, b)) {
        const dim3 grid(32, b.nslots * b.nouter);
        const int d = lds_depth(logN, b);
        if (b.prestaged) { /* first stage done by the producer */ }
        else if (d == 3) {
            if (b.reduce_in) hipLaunchKernelGGL(ntt_pass8_fwd_kernel<true>, grid, dim3(SPLIT_THREADS), 0, st, b, logN);
            else hipLaunchKernelGGL(ntt_pass8_fwd_kernel<false>, grid, dim3(SPLIT_THREADS), 0, st, b, logN);
        }
        else if (d == 2) {
            if (b.reduce_in) hipLaunchKernelGGL(ntt_pass4_fwd_kernel<true>, grid, dim3(SPLIT_THREADS), 0, st, b, logN);
            else hipLaunchKernelGGL(ntt_pass4_fwd_kernel<false>, grid, dim3(SPLIT_THREADS), 0, st, b, logN);
        }
        else if (b.reduce_in) hipLaunchKernelGGL(ntt_split_fwd_kernel<true>, grid, dim3(SPLIT_THREADS), 0, st, b, logN);
        else hipLaunchKernelGGL(ntt_split_fwd_kernel<false>, grid, dim3(SPLIT_THREADS), 0, st, b, logN);
        NttBatch c = in_place_of_dst(b);
        if (b.prestaged && b.prestaged_oop && !d && logN == 16) {
            // the producer staged the data in src: sub-transforms src -> dst (the H16 kernel only; anything else copies nothing and would
            // need the data in dst, so the caller only asks for this when ntt16_split_ok holds)
            NttBatch o = b; o.reduce_in = 0; o.split = 1;
            if (ntt16_split_ok(o)) c = o;
            else throw std::runtime_error("mkhe: internal: out-of-place prestaged launch outside the H16 path");
        }
        if (d) {
            for (int L = 1; L < d && d != 2 && d != 3; ++L) hipLaunchKernelGGL(ntt_pass_fwd_kernel, grid, dim3(SPLIT_THREADS), 0, st, c, logN, L);
            const int jobs = (b.nslots * b.nouter) << d;
            if (logN - d == 12) {
                const size_t lds = SmGeo<12>::LDSW * sizeof(u64);
                if (small) hipLaunchKernelGGL((ntt_fwd_lds_kernel<1, 12>), dim3(jobs), dim3(512), lds, st, c, d);
                else hipLaunchKernelGGL((ntt_fwd_lds_kernel<0, 12>), dim3(jobs), dim3(512), lds, st, c, d);
                return;
            }
            if (logN - d == 11) {
                const size_t lds = SmGeo<11>::LDSW * sizeof(u64);            // (18 KB: below the default dynamic-LDS limit, no attribute needed)
                if (small) hipLaunchKernelGGL((ntt_fwd_lds_kernel<1, 11>), dim3(jobs), dim3(256), lds, st, c, d);
                else hipLaunchKernelGGL((ntt_fwd_lds_kernel<0, 11>), dim3(jobs), dim3(256), lds, st, c, d);
                return;
            }
            throw std::runtime_error("mkhe: internal: LDS sub-transforms of a size no kernel has");
        }
        if (logN == 16 && ntt16_split_ok(c)) launch_ntt16_fwd_split(c, small, st);
        else if (small) launch_fwd_mode<1, false>(logN - 1, c, st); else launch_fwd_mode<0, false>(logN - 1, c, st);
        return;
    }
    if (b.reduce_in) { if (small) launch_fwd_mode<1, true>(logN, b, st); else launch_fwd_mode<0, true>(logN, b, st); }
    else             { if (small) launch_fwd_mode<1, false>(logN, b, st); else launch_fwd_mode<0, false>(logN, b, st); }
}
void launch_ntt_inv(int logN, const NttBatch& b, hipStream_t st) {
    if (b.nslots <= 0 || b.nouter <= 0) return;
    if (ntt16_inv_ok(logN, b)) {
        // launches that fill the chip: one-pass 2^14-point sub-transforms on the H16-class inverse kernel (src -> dst, canonical, N^-1 folded in),
        // then the cross stages of the larger rings as one streaming pass in place on dst
        launch_ntt16_inv(b, st, logN);
        if (logN > 14) {
            NttBatch e = in_place_of_dst(b);                  // only the dst addressing is used
            e.lazy_out = b.lazy_out; e.psi = b.psi; e.split = 0;
            const dim3 grid(32, b.nslots * b.nouter);
            if (logN == 15) hipLaunchKernelGGL(ntt_split_inv_kernel, grid, dim3(SPLIT_THREADS), 0, st, e, logN);
            else hipLaunchKernelGGL(ntt_pass4_inv_kernel, grid, dim3(SPLIT_THREADS), 0, st, e, logN);
        }
        return;
    }
    if (!b.split && use_split(logN, b)) {
        if (const int d = lds_depth(logN, b)) {
            const int jobs = (b.nslots * b.nouter) << d;
            if (logN - d == 12) hipLaunchKernelGGL(ntt_inv_ldsS_kernel<12>, dim3(jobs), dim3(512), SmGeo<12>::LDSW * sizeof(u64), st, b, d);
            else if (logN - d == 11) hipLaunchKernelGGL(ntt_inv_ldsS_kernel<11>, dim3(jobs), dim3(256), SmGeo<11>::LDSW * sizeof(u64), st, b, d);
            else throw std::runtime_error("mkhe: internal: LDS sub-transforms of a size no kernel has");     // (src -> dst, [0,2q), N^-1 folded in)
            const NttBatch ip = in_place_of_dst(b);
            const dim3 grid(32, b.nslots * b.nouter);
            NttBatch e = ip; e.lazy_out = b.lazy_out; e.psi = b.psi;
            if (d == 3) { hipLaunchKernelGGL(ntt_pass8_inv_kernel, grid, dim3(SPLIT_THREADS), 0, st, e, logN); return; }
            if (d == 2) { hipLaunchKernelGGL(ntt_pass4_inv_kernel, grid, dim3(SPLIT_THREADS), 0, st, e, logN); return; }
            for (int L = d - 1; L >= 1; --L) hipLaunchKernelGGL(ntt_pass_inv_kernel, grid, dim3(SPLIT_THREADS), 0, st, ip, logN, L);
            hipLaunchKernelGGL(ntt_split_inv_kernel, grid, dim3(SPLIT_THREADS), 0, st, e, logN);
            return;
        }
        NttBatch c = b;
        c.split = 1;
        launch_ntt_inv(logN - 1, c, st);                      // src halves -> dst halves, lazy, N^-1 folded in
        const NttBatch d = in_place_of_dst(b);                // only the dst addressing is used
        NttBatch e = d; e.lazy_out = b.lazy_out; e.psi = b.psi;
        hipLaunchKernelGGL(ntt_split_inv_kernel, dim3(32, b.nslots * b.nouter), dim3(SPLIT_THREADS), 0, st, e, logN);
        return;
    }
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

void launch_ntt_cross8_dec(const NttBatch& b, int logN, hipStream_t st) {
    if (b.nslots <= 0 || b.nouter <= 0) return;
    const dim3 grid(((1 << logN) / 8 + SPLIT_THREADS - 1) / SPLIT_THREADS, b.nslots * b.nouter);
    hipLaunchKernelGGL(ntt_pass8_fwd_kernel<true>, grid, dim3(SPLIT_THREADS), 0, st, b, logN);
}
void launch_ext_fused_lds(const ExtFusedArgs& a, hipStream_t st) {
    if (a.logN != 14 || a.nv < 1 || a.nv > EXTF_MAX_V || a.nb < 1 || a.N != (1 << a.logN)) throw std::runtime_error("mkhe: internal: ext_fused_lds_kernel outside its shape");
    for (int v = 0; v < a.nv; ++v) if (a.nk[v] < 1 || a.nk[v] > 2) throw std::runtime_error("mkhe: internal: ext_fused_lds_kernel takes one or two keys per vector");
    const int ng = (a.nb >= 2 || a.inv) ? 2 : 1;       // (one digit with the inverse half: the second group idles through the forward round and takes the second product's inverse)
    if (a.inv && (!a.psiinv || !a.aux)) throw std::runtime_error("mkhe: internal: ext_fused_lds_kernel with inverse sub-transforms outside its shape");
    const size_t lds = (size_t)ng * SmGeo<11>::LDSW * sizeof(u64);
    const dim3 grid(1 << (a.logN - 11), a.nslots, a.nv);
    if (a.inv) hipLaunchKernelGGL((ext_fused_lds_kernel<2, true>), grid, dim3(512), lds, st, a);
    else if (ng == 2) hipLaunchKernelGGL((ext_fused_lds_kernel<2, false>), grid, dim3(512), lds, st, a);
    else hipLaunchKernelGGL((ext_fused_lds_kernel<1, false>), grid, dim3(256), lds, st, a);
}
// the cross stages of an inverse launch whose sub-transforms were done by ext_fused_lds_kernel<2, true>: in place on dst, the members of a merged
// destination summed at the load
void launch_ntt_inv_cross8_sum(const NttBatch& b, int logN, hipStream_t st) {
    if (b.nslots <= 0 || b.nouter <= 0) return;
    NttBatch e = in_place_of_dst(b);
    e.lazy_out = b.lazy_out; e.psi = b.psi; e.sum_in_cross = 1;
    const dim3 grid(((1 << logN) / 8 + SPLIT_THREADS - 1) / SPLIT_THREADS, b.nslots * b.nouter);
    hipLaunchKernelGGL(ntt_pass8_inv_kernel, grid, dim3(SPLIT_THREADS), 0, st, e, logN);
}

}  // namespace mkhe
