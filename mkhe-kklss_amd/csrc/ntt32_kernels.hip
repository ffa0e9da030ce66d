// ntt32_kernels.hip -- forward negacyclic NTT, N = 2^15, ONE pass per limb ("H32"; round 4).
//
// The H16 kernel (ntt16_kernels.hip) transforms a limb in two passes of 2^14 points so that two 64-VGPR workgroups share a CU; the price is
// stage 0 computed twice (1/15 of the butterflies), the source limb read twice, two cross-wave exchanges (eight workgroup barriers) per limb,
// eight LDS re-distributions of sixteen registers per limb, and a last phase on the two-round product because a ring of twiddle PAIRS does not
// fit its 64 registers.  It runs at the package power cap (docs/DESIGN_HISTORY.md section 3 "Round 3"), where what a limb costs is the energy of its
// instructions and bytes rather than how well its phases overlap.  H32 is the other trade: ONE workgroup of 1024 threads per CU holds the
// whole limb (32 coefficients per thread, 128 VGPRs, 4 waves per SIMD) and runs the 15 stages as three register phases of five:
//   A: index bits 14..10 in registers, thread = bits 9..0                     31 twiddle pairs, uniform per workgroup (scalar loads)
//   B: bits 9..5, thread = (bits 14..10, bits 4..0): wave = bits 14..11       31 pairs per lane (two distinct addresses per wave)
//   C: bits 4..0, thread = bits 14..5                                         31 pairs per lane
//   E: store layout: registers = bits 10..6 of the wave's 2048 coefficients, lane = bits 5..0 (512 B contiguous per store instruction)
// -- no stage repeated, every source word loaded once, ONE cross-wave exchange (A -> B, four barriers) per limb, three re-distributions of 32
// registers instead of eight of 16, every stage on the one-round product (mm30u / mm31 of h16_arith.h).  B -> C and C -> E stay inside a wave.
//
// LDS image, one 32-bit plane at a time, 16 wave regions of 2080 words (133,120 B of the CU's 160 KiB):
//   A -> B: word of coefficient p = (p >> 10) * 1040 + (p & 1023): written lane-linear (register r, thread t: r * 1040 + t), read by the thread
//           (hi = bits 14..10, lo = bits 4..0) at hi * 1040 + lo + 32 r': rows 2 w, 2 w + 1 are exactly wave w's own region, so that after its
//           last read a wave owns its region again and B -> C, C -> E need no barrier;
//   B -> C, C -> E: region base + 65 * register + lane on the write side (lane-linear, conflict-free), base + 65 * (lane & 31) + 32 * (lane >> 5) +
//           register (B -> C) and base + 65 * (lane & 31) + (lane >> 5) + 2 * register (C -> E) on the read side: 65 = 1 mod 32, conflict-free
//           under the 32-bank rule of ds_read_b32.
// Same arithmetic, ranges and output representatives as H16 (U class: never reduced; 59/60-bit primes: partial reduction after phases A and B;
// canonical or biased engine-internal outputs), so that the kernels are interchangeable bit for bit on everything a consumer can see.
//
// Replaces: lattigo ring.NTTLvl as called from DecomposeSingleNTT (mkrlwe/keyswitch.go:21-31,49-73).
#include "ntt_kernels.h"
#ifndef MKHE_MM_VOPAQUE
#define MKHE_MM_VOPAQUE 0
#endif
#include "h16_arith.h"
#include <cstdlib>
#include <mutex>
#include <stdexcept>

namespace mkhe {
namespace h32 {
using namespace h16;

constexpr int NN = 1 << 15, NT = 1024;
constexpr int ROW = 1040;                  // words per row (index bits 14..10) of the A -> B image
constexpr int WREG = 2 * ROW;              // words per wave region = 32 * 65
constexpr int TWB = 16 * WREG;             // word offset of the waves' phase-B twiddle rows (256 words each)
constexpr int LDS_WORDS = TWB + 16 * 256;
#ifndef MKHE_H32_RING
#define MKHE_H32_RING 5
#endif
constexpr int RING = MKHE_H32_RING;        // twiddle pairs (4 VGPRs each) live per lane in phases B / C
#ifndef MKHE_H32_PRIO
#define MKHE_H32_PRIO 0                    // raised wave priority from the loads to the end of the cross-wave exchange (as in H16)
#endif
#ifndef MKHE_H32_STAGGER
#define MKHE_H32_STAGGER 0
#endif
#ifndef MKHE_H32_SLEEP
#define MKHE_H32_SLEEP 0
#endif
#ifndef MKHE_H32_PHPRIO
#define MKHE_H32_PHPRIO 2
#endif
#ifndef MKHE_H32_PREFETCH
#define MKHE_H32_PREFETCH 0                // (experiment, slower by 3 %: DESIGN.md section 10) the next limb's source loads are issued between the stores of this one (register by register)
#endif
#ifndef MKHE_NTT32_DEFAULT
// MKHE_NTT32: 0 = every launch on the two-pass H16 kernel, 1 = this kernel wherever it applies, 2 (default) = per launch shape the engine times a block
// of launches of each kernel inside the caller's workload, once its clocks have settled, and keeps the faster one (Context::ntt_pick).  Back to back
// this kernel is ahead on every part (230 against 244 us for 1792 limbs, same call); inside the MulRelin it is ahead by 2-5 % on most parts (0.526-0.542
// against 0.506-0.519 of the roofline, eight same-call pairs, MulRelin/s + 0-2 %) and BEHIND by 6 % on some (0.46-0.48 against 0.49-0.515, MulRelin/s
// - 1.6 %: parts that sit at 1255 W and 2.09 GHz under it where the others reach 1400 W and 2.2-2.3 GHz, with the same configured cap -- the two-pass
// kernel's in-context time is the same on both kinds): profiles/README.md.
#define MKHE_NTT32_DEFAULT 2
#endif
template <int... I, class F> __device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }

template <bool SW, bool UC> __device__ __forceinline__ void bf(u64& U, u64& V, const u64* tw, const MC& c) {
#ifdef MKHE_H32_X_NOBFLY        // MKHE_ABLATION: timing experiment only (wrong results)
    U += tw[0] + tw[1]; return;
#endif
    const i64 T = UC ? mm30u<SW>((i64)V, tw[0], tw[1], c) : mm31<SW>((i64)V, tw[0], tw[1], c);
    const i64 u = (i64)U;
    U = (u64)(u + T);
    V = (u64)(u - T);
}
// TWO independent butterflies as ONE asm block (MKHE_H32_BF2): the product temporaries live in pinned registers (v[124:125] + v123, v[126:127] + v122), so
// that the block can name their halves -- the low word that the Montgomery digit multiplies, the words of T in the 32-bit subtraction -- and nothing
// is cut where the compiler would pad an asm boundary with s_nop.  The two wait states gfx940 asks for between a VALU write of a carry register
// and the VALU read of it are filled with the other butterfly's instructions: one s_nop 0 per pair of butterflies instead of three per butterfly.
// A lone wave on a SIMD (the tail of a limb: the other waves wait at the barrier of the cross-wave exchange) issues a butterfly in 53 instead of
// 70 cycles (tools/ubench/bfly_asm_rate.hip); with four waves the nops are hidden either way.
//   T = a w mod q (mm30u / mm31 of h16_arith.h, same columns), Y = U - T (v_sub_co / v_subb on halves), X = U + T (v_lshl_add_u64).
#ifndef MKHE_H32_BF2
#define MKHE_H32_BF2 1
#endif
#define H32_BF2_ASM(OPA, SH, TWC)                                                                                                      \
    asm("v_mad_" OPA " v[124:125], %[k], %[a0A], %[u0A], 0\n\t"                                                                          \
        "v_mad_i64_i32 v[124:125], %[k], %[a1A], %[v0A], v[124:125]\n\t"                                                                 \
        "v_mul_lo_u32 v123, v124, %[ninv]\n\t"                                                                                          \
        "v_bfe_i32 v123, v123, 0, " SH "\n\t"                                                                                           \
        "v_mad_i64_i32 v[124:125], %[k], v123, %[p0], v[124:125]\n\t"                                                                   \
        "v_ashrrev_i64 v[124:125], " SH ", v[124:125]\n\t"                                                                              \
        "v_mad_" OPA " v[124:125], %[k], %[a0A], %[u1A], v[124:125]\n\t"                                                                 \
        "v_mad_i64_i32 v[124:125], %[k], %[a1A], %[v1A], v[124:125]\n\t"                                                                 \
        "v_mad_i64_i32 v[124:125], %[k], v123, %[p1], v[124:125]\n\t"                                                                   \
        "v_sub_co_u32 %[ylA], vcc, %[ulA], v124\n\t"                                                                                    \
        "v_mad_" OPA " v[126:127], %[k], %[a0B], %[u0B], 0\n\t"                                                                          \
        "v_mad_i64_i32 v[126:127], %[k], %[a1B], %[v0B], v[126:127]\n\t"                                                                 \
        "v_subb_co_u32 %[yhA], vcc, %[uhA], v125, vcc\n\t"                                                                              \
        "v_mul_lo_u32 v122, v126, %[ninv]\n\t"                                                                                          \
        "v_bfe_i32 v122, v122, 0, " SH "\n\t"                                                                                           \
        "v_mad_i64_i32 v[126:127], %[k], v122, %[p0], v[126:127]\n\t"                                                                   \
        "v_ashrrev_i64 v[126:127], " SH ", v[126:127]\n\t"                                                                              \
        "v_mad_" OPA " v[126:127], %[k], %[a0B], %[u1B], v[126:127]\n\t"                                                                 \
        "v_mad_i64_i32 v[126:127], %[k], %[a1B], %[v1B], v[126:127]\n\t"                                                                 \
        "v_mad_i64_i32 v[126:127], %[k], v122, %[p1], v[126:127]\n\t"                                                                   \
        "v_sub_co_u32 %[ylB], %[cb], %[ulB], v126\n\t"                                                                                  \
        "v_lshl_add_u64 %[UA], v[124:125], 0, %[UA]\n\t"                                                                                \
        "s_nop 0\n\t"                                                                                                                   \
        "v_subb_co_u32 %[yhB], %[cb], %[uhB], v127, %[cb]\n\t"                                                                          \
        "v_lshl_add_u64 %[UB], v[126:127], 0, %[UB]"                                                                                    \
        : [UA] "+v"(U0), [UB] "+v"(U1), [ylA] "=&v"(ylA), [yhA] "=&v"(yhA), [ylB] "=&v"(ylB), [yhB] "=&v"(yhB), [k] "=&s"(k), [cb] "=&s"(cb)   \
        : [a0A] "v"(a0A), [a1A] "v"(a1A), [a0B] "v"(a0B), [a1B] "v"(a1B), [ulA] "v"(lo32(U0)), [uhA] "v"(hi32(U0)), [ulB] "v"(lo32(U1)),      \
          [uhB] "v"(hi32(U1)), [u0A] TWC(u0A), [u1A] TWC(u1A), [v0A] TWC(v0A), [v1A] TWC(v1A), [u0B] TWC(u0B), [u1B] TWC(u1B),                \
          [v0B] TWC(v0B), [v1B] TWC(v1B), [ninv] "s"(c.ninv), [p0] "s"(c.p0), [p1] "s"(c.p1)                                                   \
        : "v122", "v123", "v124", "v125", "v126", "v127", "vcc")
template <bool SW, bool UC> __device__ __forceinline__ void bf2(u64& U0, u64& V0, const u64* ta, u64& U1, u64& V1, const u64* tb, const MC& c) {
#if defined(MKHE_H32_X_NOBFLY) || !MKHE_H32_BF2
    bf<SW, UC>(U0, V0, ta, c);
    __builtin_amdgcn_sched_barrier(0);
    bf<SW, UC>(U1, V1, tb, c);
#else
    // data digits: the word as it stands (U class: hi signed, lo unsigned) or balanced 32-bit digits (mm31)
    u32 a0A = lo32(V0), a0B = lo32(V1);
    i32 a1A = (i32)hi32(V0), a1B = (i32)hi32(V1);
    if constexpr (!UC) { a1A = (i32)(hi32(V0) + (a0A >> 31)); a1B = (i32)(hi32(V1) + (a0B >> 31)); }
    i32 u0A = (i32)lo32(ta[0]), u1A = (i32)hi32(ta[0]), v0A = (i32)lo32(ta[1]), v1A = (i32)hi32(ta[1]);
    i32 u0B = (i32)lo32(tb[0]), u1B = (i32)hi32(tb[0]), v0B = (i32)lo32(tb[1]), v1B = (i32)hi32(tb[1]);
    if constexpr (SW) asm("" : "+s"(u0A), "+s"(u1A), "+s"(v0A), "+s"(v1A), "+s"(u0B), "+s"(u1B), "+s"(v0B), "+s"(v1B));
    u32 ylA, yhA, ylB, yhB; u64 k, cb;
#define H32_TW_S(x) "s"(x)
#define H32_TW_V(x) "v"(x)
    if constexpr (UC) { if constexpr (SW) H32_BF2_ASM("u64_u32", "30", H32_TW_S); else H32_BF2_ASM("u64_u32", "30", H32_TW_V); }
    else              { if constexpr (SW) H32_BF2_ASM("i64_i32", "31", H32_TW_S); else H32_BF2_ASM("i64_i32", "31", H32_TW_V); }
    V0 = ((u64)yhA << 32) | ylA;
    V1 = ((u64)yhB << 32) | ylB;
#endif
}
// butterflies G0 .. G0 + NG - 1 of the stage on register bit B with scalar twiddle pairs: tw[2 i], tw[2 i + 1] = (u, v) of twiddle (G0 >> B) + i
template <bool UC, int B, int G0, int NG> __device__ __forceinline__ void stage_s(u64 (&x)[32], const u64* tw, const MC& c) {
    static_assert(NG % 2 == 0, "butterflies come in pairs");
#pragma unroll
    for (int g = G0; g < G0 + NG; g += 2) {
        const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1)), g1 = g + 1;
        const int i1 = ((g1 >> B) << (B + 1)) | (g1 & ((1 << B) - 1));
        bf2<true, UC>(x[i0], x[i0 | (1 << B)], tw + 2 * ((g >> B) - (G0 >> B)), x[i1], x[i1 | (1 << B)], tw + 2 * ((g1 >> B) - (G0 >> B)), c);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// butterflies g, g + 1 of the stage on register bit B, per-lane twiddle pairs
template <bool UC, int B> __device__ __forceinline__ void bf1x2(u64 (&x)[32], int g, const u64* ta, const u64* tb, const MC& c) {
    const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1)), g1 = g + 1;
    const int i1 = ((g1 >> B) << (B + 1)) | (g1 & ((1 << B) - 1));
    bf2<false, UC>(x[i0], x[i0 | (1 << B)], ta, x[i1], x[i1 | (1 << B)], tb, c);
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void reduce32(u64 (&x)[32], const MC& c, bool on) {
    if (on) {
#pragma unroll
        for (int r = 0; r < 32; ++r) { x[r] = (u64)pred((i64)x[r], c); __builtin_amdgcn_sched_barrier(0); }
    }
}
template <int K> __device__ __forceinline__ void wait_pair(u64& a, u64& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(K)); }

typedef __attribute__((address_space(3))) u32* lptr;
typedef volatile __attribute__((address_space(3))) u32* vlptr;

// one plane of the 32 registers: written at wr[woff(r)], read back from rd[roff(r)]; CROSS: workgroup barriers around (A -> B)
template <int WS, int RS, bool CROSS, bool HI> __device__ __forceinline__ void plane(u64 (&x)[32], lptr wr, vlptr rd) {
#pragma unroll
    for (int r = 0; r < 32; ++r) wr[WS * r] = HI ? hi32(x[r]) : lo32(x[r]);
    xsync<CROSS>();
    // (volatile: the reads stay single ds_read_b32 -- merged into ds_read2_b32 the two words of one instruction land in a consecutive register
    // pair and every coefficient needs two v_mov to get its own (low, high) pair back)
#pragma unroll
    for (int r = 0; r < 32; ++r) x[r] = HI ? (((u64)rd[RS * r] << 32) | lo32(x[r])) : ((x[r] & 0xffffffff00000000ull) | rd[RS * r]);
}
template <int WS, int RS, bool CROSS> __device__ __forceinline__ void exchange(u64 (&x)[32], lptr wr, vlptr rd) {
#ifdef MKHE_H32_X_NOXCHG        // MKHE_ABLATION: timing experiment only (wrong results)
    return;
#endif
    if constexpr (CROSS) __syncthreads();          // every wave is done with its region (C -> E reads of the previous limb)
    plane<WS, RS, CROSS, false>(x, wr, rd);
    xsync<CROSS>();
    plane<WS, RS, CROSS, true>(x, wr, rd);
    if constexpr (!CROSS) xsync<false>();
}

// The five stages of a per-lane phase (B or C): twiddle pair t = 2^j - 1 + i of the phase (stage j on register bit 4 - j, i < 2^j) is the pair
// number (e << j) + i of the modulus's table, e = 32 + bits 14..10 (phase B) or 1024 + bits 14..5 (phase C) of the thread's coefficients: one
// 16-byte load per lane and pair, RING - 1 pairs requested ahead of the one in use.  `between` runs after the first requests (the re-distribution
// that precedes the phase); `last` just before the butterflies of the last stage.
// TAB: the pairs come from the load-order table (NttBatch::psi31c: base = this wave's [31][64] block, one 1 KiB row per pair ordinal)
typedef __attribute__((address_space(3))) u64x2* lptr2;
template <int T, int TAB> __device__ __forceinline__ void load_pair(u64 (&g)[RING][2], gcptr base, const unsigned e) {
    if constexpr (T < 31) {
        if constexpr (TAB == 2) {
            // phase B: the wave's row in LDS (base = its byte address as an integer in the low word of the pointer slot, e = 16 * 31 * (lane >> 5))
            const u64x2 v = *(lptr2)(unsigned)(e + 16 * T);
            g[T % RING][0] = v.x; g[T % RING][1] = v.y;
        }
        else if constexpr (TAB == 1) ld2(g[T % RING], (gcptr2)sbk(base, 128 * T), e);           // e = lane
        else {
#ifdef MKHE_H32_X_NOTWB        // MKHE_ABLATION: timing experiment only (wrong results): no per-lane twiddle loads in phase B
            g[T % RING][0] = e + T; g[T % RING][1] = e; return;
#endif
            constexpr int j = T < 1 ? 0 : T < 3 ? 1 : T < 7 ? 2 : T < 15 ? 3 : 4, i = T + 1 - (1 << j);
            ld2(g[T % RING], (gcptr2)sbase(base) + i, e << j);
        }
    }
}
template <bool UC, int TAB, class F0, class F1>
__device__ __forceinline__ void phase_lane(u64 (&x)[32], gcptr base, const unsigned e, const MC& c, F0&& between, F1&& last) {
    u64 g[RING][2];
    static_for(std::make_integer_sequence<int, RING - 1>{}, [&](auto tc) { load_pair<decltype(tc)::value, TAB>(g, base, e); });
    between();
    static_for(std::make_integer_sequence<int, 40>{}, [&](auto nc) {
        constexpr int n = 2 * decltype(nc)::value;                 // butterflies n, n + 1
        constexpr int j = n >> 4, gi = n & 15, B = 4 - j;
        constexpr int t = (1 << j) - 1 + (gi >> B), t1 = (1 << j) - 1 + ((gi + 1) >> B);
        constexpr int np = n - 1, jp = np >> 4, tp = n == 0 ? -1 : (1 << jp) - 1 + ((np & 15) >> (4 - jp));
        if constexpr (t != tp) load_pair<t + RING - 1, TAB>(g, base, e);            // first use of pair t: its predecessor's slot is free
        if constexpr (n == 64) last();
        bf1x2<UC, B>(x, gi, g[t % RING], g[t1 % RING], c);
        if constexpr (t1 != t) load_pair<t1 + RING - 1, TAB>(g, base, e);          // (pair t's slot: free only now)
    });
}

// diagnostic build (make trace): shader-clock stamps per wave, 32 words per (job, wave): tools/ntt32_trace.py
#ifdef MKHE_PHASE_TRACE
#define H32_STAMP(k) do { if (jb.trace) jb.trace[(long)wv * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define H32_STAMP(k) do { } while (0)
#endif
// one limb; big: a 59/60-bit modulus (balanced path only)
// asm forms of the source load and the result store: issued in program order (volatile, memory clobber), invisible to the compiler's own
// s_waitcnt bookkeeping -- the counted waits of stage 0 are written out (wait_pair)
__device__ __forceinline__ void ld_into(u64& v, gcptr base, unsigned byte_off) {
#ifdef MKHE_H16_X_NOSRC         // MKHE_ABLATION: timing experiment only (wrong results)
    v = (u64)byte_off * 0x9E3779B97F4A7C15ull; return;
#endif
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(base) : "memory");
}
__device__ __forceinline__ void st_nt(gptr base, unsigned byte_off, u64 v) {
#ifdef MKHE_H32_X_NOSTORE       // MKHE_ABLATION: timing experiment only
    if (v != 0x123456789abcdefull) return;
#endif
    asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(byte_off), "v"(v), "s"(base) : "memory");
}
template <bool DEC, bool UC>
__device__ __forceinline__ void limb(const Job& jb, const bool big_, u32* lds, const int wv, u64 (&x)[32]) {
    const bool big = UC ? false : big_;
    smodptr mp = jb.mp;
    const u64 qs = mp->qs;
    MC c;
    c.q = mp->q; c.ninv = mp->ninv32;
    c.q0 = (i32)lo32(qs); c.q1 = (i32)hi32(qs);
    c.finv = __builtin_bit_cast(float, mp->finv);
    if constexpr (UC) {
        c.p0 = (i32)((u32)c.q << 2) >> 2;                   // q = p1 2^30 + p0, |p0| <= 2^29
        c.p1 = (i32)((c.q - (u64)(i64)c.p0) >> 30);
    } else {
        c.p0 = (i32)((u32)c.q << 1) >> 1;                   // q = p1 2^31 + p0, |p0| <= 2^30
        c.p1 = (i32)((c.q - (u64)(i64)c.p0) >> 31);
    }
    asm("" : "+s"(c.q0), "+s"(c.q1), "+s"(c.ninv), "+s"(c.p0), "+s"(c.p1));
    scptr p31 = (scptr)jb.psi31;                            // (u, v) of twiddle i at words 2 i, 2 i + 1
    gcptr p31v = (gcptr)jb.psi31;
    const gcptr src = jb.src; const gptr dst = jb.dst;
    const bool red = DEC && jb.red;
    H32_STAMP(0);
    // ---- loads (pairs x[g], x[g + 16] in the order stage 0 consumes them) + stage 0 behind counted waits
    if (MKHE_H32_PRIO) __builtin_amdgcn_s_setprio(MKHE_H32_PRIO);
    {
        const unsigned tb = 8u * (unsigned)(wv * 64 + lane_id());
        // this wave's phase-B twiddle row (issued first: it has landed when the last counted wait below returns)
        u64 trow[2];
#if MKHE_H32_PREFETCH
        // the loads are in flight (issued between the previous limb's stores: step g = store r_g, store r_g + 16, load x[g], load x[g + 16]; the
        // first limb's have landed): behind the pair of step g there are 4 (15 - g) memory operations and this load
        (void)tb; (void)src;
        ld2(trow, (gcptr2)sbase((gcptr)jb.psi31n) + 64 * wv, (unsigned)lane_id());
#else
        ld2(trow, (gcptr2)sbase((gcptr)jb.psi31n) + 64 * wv, (unsigned)lane_id());
#pragma unroll
        for (int g = 0; g < 16; ++g) { x[g] = ld_issue(sbk(src, g * NT), tb); x[g + 16] = ld_issue(sbk(src, (g + 16) * NT), tb); }
#endif
        const u64 t0[2] = {p31[2], p31[3]};                 // psi[1]
        static_for(std::make_integer_sequence<int, 8>{}, [&](auto gc) {
            constexpr int g = 2 * decltype(gc)::value;
            wait_pair<MKHE_H32_PREFETCH ? 61 - 4 * g : 30 - 2 * g>(x[g], x[g + 16]);
            wait_pair<MKHE_H32_PREFETCH ? 57 - 4 * g : 28 - 2 * g>(x[g + 1], x[g + 17]);
            // digits of a foreign modulus (Decompose) may be far above q: bring them to (-q, q) first where the class has no headroom for them
            if ((big && (jb.sched & 1)) || red) {
                x[g] = (u64)pred((i64)x[g], c); x[g + 16] = (u64)pred((i64)x[g + 16], c);
                x[g + 1] = (u64)pred((i64)x[g + 1], c); x[g + 17] = (u64)pred((i64)x[g + 17], c);
            }
            bf2<true, UC>(x[g], x[g + 16], t0, x[g + 1], x[g + 17], t0, c);
            __builtin_amdgcn_sched_barrier(0);
        });
        {
            u64x2 v; v.x = trow[0]; v.y = trow[1];
            *((lptr2)((lptr)lds + TWB + 256 * wv) + lane_id()) = v;          // (read back by this wave only: LDS operations of a wave execute in order)
        }
    }
    H32_STAMP(1);
    // ---- phase A, stages 1..4: scalar pairs psi[2^k + i], fetched a stage ahead, the sixteen of the last stage in two halves
    {
        u64 t1[4], t2[8], t3[16], t4[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) t1[i] = p31[4 + i];
#pragma unroll
        for (int i = 0; i < 8; ++i) t2[i] = p31[8 + i];
        stage_s<UC, 3, 0, 16>(x, t1, c);
#pragma unroll
        for (int i = 0; i < 16; ++i) t3[i] = p31[16 + i];
        stage_s<UC, 2, 0, 16>(x, t2, c);
#pragma unroll
        for (int i = 0; i < 16; ++i) t4[i] = p31[32 + i];
        stage_s<UC, 1, 0, 16>(x, t3, c);
        stage_s<UC, 0, 0, 8>(x, t4, c);
#pragma unroll
        for (int i = 0; i < 16; ++i) t4[i] = p31[48 + i];
        stage_s<UC, 0, 8, 8>(x, t4, c);
    }
    // the 59/60-bit primes: |x| <= 2^60 + 5 * 1.03 q < 2^62.9 so far; back to |x| <= 0.51 q after phases A and B
    reduce32(x, c, big);
    H32_STAMP(2);
    // ---- phase B: bits 9..5, behind the cross-wave re-distribution (its first twiddle pairs are requested before it)
    {
        const unsigned e = 4u * (unsigned)(TWB + 256 * wv) + 16u * 31u * (unsigned)(lane_id() >> 5) + (unsigned)(unsigned long)(lptr)lds;
        phase_lane<UC, 2>(x, p31v, e, c,
            [&] {
                const int l2 = lane_id();
                lptr wr = (lptr)lds + wv * 64 + l2;                                          // word r * ROW + t
                vlptr rd = (vlptr)((lptr)lds + (2 * wv + (l2 >> 5)) * ROW + (l2 & 31));      // word hi * ROW + 32 r' + lo
                exchange<ROW, 32, true>(x, wr, rd);
                if (MKHE_H32_PRIO) __builtin_amdgcn_s_setprio(0);
                H32_STAMP(3);
                if (MKHE_H32_PHPRIO == 1) __builtin_amdgcn_s_setprio(3);
                if (MKHE_H32_PHPRIO == 2) { if (wv < 8) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
#if MKHE_H32_SLEEP > 0          // experiment: the upper eight waves (two per SIMD) start the barrier-free part of the limb late, so that their LDS / memory phases
                // fall under the lower eight's butterflies and vice versa (s_setprio does not separate them: profiles/r4_h32_notes.txt)
                if (wv >= 8) { for (int i = 0; i < MKHE_H32_SLEEP; ++i) __builtin_amdgcn_s_sleep(127); }
#endif
            },
            [] {});
    }
    reduce32(x, c, big);
    if (MKHE_H32_PHPRIO == 1) __builtin_amdgcn_s_setprio(2);
    H32_STAMP(4);
    // ---- phase C: bits 4..0, behind the B -> C re-distribution inside the wave's own region
    {
        const unsigned e = (unsigned)lane_id();
        phase_lane<UC, 1>(x, (gcptr)jb.psif + (long)wv * (31 * 64 * 2), e, c,
            [&] {
                const int l2 = lane_id();
                lptr wr = (lptr)lds + wv * WREG + l2;
                vlptr rd = (vlptr)((lptr)lds + wv * WREG + 65 * (l2 & 31) + 32 * (l2 >> 5));
                exchange<65, 1, false>(x, wr, rd);
                H32_STAMP(5);
            },
            [&] {
                if (!(big || !jb.skip_norm)) {
                    // the positive bias of engine-internal digits enters through the sixteen U operands of the LAST stage: X = (U + b) + T, Y = (U + b) - T
                    const i64 bias = UC ? (i64)((c.q << 6) + (c.q << 3) + (c.q << 1) + c.q) : (i64)((c.q << 4) + (c.q << 3));
#pragma unroll
                    for (int r = 0; r < 32; r += 2) x[r] = (u64)((i64)x[r] + bias);
                }
            });
    }
    if (MKHE_H32_PHPRIO == 1) __builtin_amdgcn_s_setprio(1);
    if (MKHE_H32_PHPRIO == 2) { if (wv < 8) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    H32_STAMP(6);
    // ---- output representative
    if (big || !jb.skip_norm) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const i64 y = pred((i64)x[r], c);                                    // (-q, q)
            x[r] = (u64)(y + ((y >> 63) & (i64)c.q));                           // canonical (lattigo: final BRedAdd)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    H32_STAMP(7);
}
// ---- C -> E and the stores: register r = words wave * 2048 + 64 r + lane.  Common to both modulus classes, and with MKHE_H32_PREFETCH the ONE place
// where source loads are issued: register by register the next limb's words follow this limb's stores into the memory pipeline (nsrc: the
// next job's source, or this one's again behind the last job -- loads nobody waits for), so that stage 0 of the next limb starts behind two
// stores and two loads instead of behind thirty-two stores and its own request.
__device__ __forceinline__ void tail(u64 (&x)[32], const Job& jb, gcptr nsrc, u32* lds, const int wv) {
    const gptr dst = jb.dst;
    const int l = lane_id();
    lptr wr = (lptr)lds + wv * WREG + l;
    vlptr rd = (vlptr)((lptr)lds + wv * WREG + 65 * (l & 31) + (l >> 5));
    exchange<65, 2, false>(x, wr, rd);
    H32_STAMP(8);
    const unsigned lb = 8u * (unsigned)l;
#if MKHE_H32_PREFETCH
    const unsigned tb = 8u * (unsigned)(wv * 64 + l);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        st_nt(sbk(dst, wv * 2048 + g * 64), lb, x[g]);
        st_nt(sbk(dst, wv * 2048 + (g + 16) * 64), lb, x[g + 16]);
        ld_into(x[g], sbk(nsrc, g * NT), tb);
        ld_into(x[g + 16], sbk(nsrc, (g + 16) * NT), tb);
    }
#else
    (void)nsrc;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
#ifdef MKHE_H32_X_NOSTORE       // MKHE_ABLATION: timing experiment only
        if (x[r] != 0x123456789abcdefull) continue;
#endif
        __builtin_nontemporal_store(x[r], (u64 __attribute__((address_space(1)))*)at(sbk(dst, wv * 2048 + r * 64), lb));
    }
#endif
    H32_STAMP(9);
    if (MKHE_H32_PHPRIO == 1) __builtin_amdgcn_s_setprio(0);
}

template <bool DEC>
__global__ void __launch_bounds__(NT, 4) __attribute__((amdgpu_num_vgpr(128))) ntt32_fwd_kernel(NttBatch b) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int njobs = b.nslots * b.nouter;
#if MKHE_H32_STAGGER == 2       // static wave priorities: waves w and w + 4 k share a SIMD; two (four) priority groups per SIMD
    if (wv < 8) __builtin_amdgcn_s_setprio(2);
#elif MKHE_H32_STAGGER == 5     // reversed: the waves that the arbiter's oldest-first order leaves behind get the higher priority
    { const int g = wv >> 2; if (g == 3) __builtin_amdgcn_s_setprio(3); else if (g == 2) __builtin_amdgcn_s_setprio(2); else if (g == 1) __builtin_amdgcn_s_setprio(1); }
#elif MKHE_H32_STAGGER == 4
    { const int g = wv >> 2; if (g == 0) __builtin_amdgcn_s_setprio(3); else if (g == 1) __builtin_amdgcn_s_setprio(2); else if (g == 2) __builtin_amdgcn_s_setprio(1); }
#endif
    // job2 (position in this workgroup's walk) -> job (index in the slot-major list): the long jobs (59/60-bit moduli, the head of the list) are
    // placed on the CUs that own one position fewer when the list leaves the last row ragged -- the bijection of ntt16_kernels.hip fwd_body with
    // C = the number of workgroups (one per CU here)
    auto job_of = [&](int job2) {
        int job = job2;
        kargptr kl = (kargptr)__builtin_amdgcn_kernarg_segment_ptr();
        const int B = kl->lpt.B;
        if (B > 0) {
            const int C = kl->lpt.C, r = kl->lpt.r, w = C - r, full = kl->lpt.full, rem = kl->lpt.rem;
            const int row = (int)udiv_magic((unsigned)job2, kl->lpt.magic_C), col = job2 - row * C;
            const int srow = row < full ? w : (row == full ? rem : 0);
            const int before = row <= full ? row * w : B;
            if (col >= r && col - r < srow) job = before + (col - r);
            else job = B + job2 - before - (col > r ? (col - r < srow ? col - r : srow) : 0);
        }
        return job;
    };
    // source limb of a job
    auto src_of = [&](int job) {
        kargptr kb = (kargptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kb));
        const int nouter = kb->nouter;
        const int s = (int)udiv_magic((unsigned)job, kb->magic_nouter);
        int outer = job - s * nouter;
        const int m = kb->mod[s], p = kb->pos[s];
        const u64* sbase_ = kb->src;
        if (kb->nitems > 0) {
            const int item = (int)udiv_magic((unsigned)outer, kb->magic_opi);
            outer -= item * kb->outers_per_item;
            sbase_ = kb->src_items[item];
        }
        return (gcptr)(sbase_ + (long)outer * kb->src_outer + (long)(kb->src_mapped ? m : p) * kb->src_inner);
    };
    u64 x[32];
    // all registers of x behind every memory operation in flight (s_waitcnt vmcnt(0)): where the compiler may move them (between the two loops below)
    auto landed = [&] {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),
                     "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]) : : "memory");
        asm volatile("" : "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]), "+v"(x[24]), "+v"(x[25]),
                     "+v"(x[26]), "+v"(x[27]), "+v"(x[28]), "+v"(x[29]), "+v"(x[30]), "+v"(x[31]) : : "memory");
    };
#if MKHE_H32_PREFETCH
    if ((int)blockIdx.x < njobs) {
        // the first limb's words (every later limb's are requested by the limb before it); landed before the loop: one exposed round trip per launch
        const gcptr src0 = src_of(job_of(blockIdx.x));
        const unsigned tb = 8u * (unsigned)((int)threadIdx.x);
#pragma unroll
        for (int g = 0; g < 32; ++g) ld_into(x[g], sbk(src0, g * NT), tb);
        landed();
    }
#endif
    // one limb: the job's constants, the transform (limb<DEC, UC>), the stores with the next job's loads between them (tail)
    auto body = [&](const int job2, auto ucc) {
        constexpr bool UC = decltype(ucc)::value;
        const int job = job_of(job2);
        kargptr kb = (kargptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kb));
        const int nouter = kb->nouter;
        const int s = (int)udiv_magic((unsigned)job, kb->magic_nouter);
        int outer = job - s * nouter;
        const int m = kb->mod[s], p = kb->pos[s];
        const u64* sbase_ = kb->src; u64* dbase_ = kb->dst;
        if (kb->nitems > 0) {
            const int opi = kb->outers_per_item;
            const int item = (int)udiv_magic((unsigned)outer, kb->magic_opi);
            outer -= item * opi;
            sbase_ = kb->src_items[item]; dbase_ = kb->dst_items[item];
        }
        Job jb;
        jb.src = (gcptr)(sbase_ + (long)outer * kb->src_outer + (long)(kb->src_mapped ? m : p) * kb->src_inner);
        jb.dst = (gptr)(dbase_ + (long)outer * kb->dst_outer + (long)(kb->dst_mapped ? m : p) * kb->dst_inner);
        jb.psi = kb->psi + (long)m * NN;
        jb.psi31 = kb->psi31 + 2 * (long)m * NN;
        jb.root = 1;
        jb.psi31n = kb->psi31b + (long)m * (16 * 64 * 2);         // (this kernel's use of the slot: the rows of the middle phase)
        jb.trace = kb->trace ? kb->trace + (long)job2 * 16 * 32 : nullptr;
        jb.psif = kb->psi31c + (long)m * (16 * 31 * 64 * 2);      // (this kernel's use of the slot: the load-order pairs of the last phase)
        // bit 0 of the H16 schedule (reduce at the load) carries over: inputs below 2^60 leave a 59/60-bit modulus below 2^62.9 through the five
        // stages of phase A (2^60 + 5 * 1.03 q); lazy inputs (BFV digits, ring-R polynomials) are reduced at the load
        // (the byte through a scalar dword load: a byte load is a VECTOR memory instruction, and the s_waitcnt vmcnt(0) the compiler puts behind it
        // would wait for every store and prefetched word of the previous limb right here)
        jb.sched = kb->src_lazy ? 15 : (int)((((const __attribute__((address_space(4))) unsigned*)kb->sched)[m >> 2] >> (8 * (m & 3))) & 0xffu);
#ifdef MKHE_H16_X_SCHEDBYTE     // MKHE_ABLATION: round 3's form of the line above (same value; the vector byte load and its vmcnt(0)), for the A/B in one call
        jb.sched = kb->src_lazy ? 15 : kb->sched[m];
#endif
        jb.mp = (smodptr)kb->mods + m;
        jb.skip_norm = kb->skip_norm != 0;
        jb.red = false;
        if constexpr (DEC) {
            int sm = m;
            const int rs = kb->reduce_src_mod_is_outer;
            if (rs == 1) sm = outer; else if (rs == 2) sm = kb->outer_mod[outer];
            const u64 qsb = ((smodptr)kb->mods)[sm].q << (kb->src_lazy ? 2 : 0);     // bound of the digit values (< 2^63)
            // U class: raw canonical digits of any modulus (< 2^60) fit its range budget (input + 75 q of growth + the 75 q bias < 2^62); the balanced
            // path reduces what exceeds 4q -- the rules of ntt16_kernels.hip
            if constexpr (UC) jb.red = qsb >= (1ull << 62) - 150 * jb.mp->q;
            else jb.red = qsb > 4 * jb.mp->q;
        }
#ifdef MKHE_PHASE_TRACE
        if (jb.trace && ((int)threadIdx.x & 63) == 0) { u64* tw = jb.trace + (long)wv * 32; tw[12] = __builtin_amdgcn_s_memrealtime(); tw[14] = blockIdx.x; tw[15] = m; }
#endif
        limb<DEC, UC>(jb, UC ? false : ((kb->small_slots >> s) & 1) == 0, lds, wv, x);
        // (the next job's source, this one's again behind the last: see tail)
        gcptr nsrc = jb.src;
        if (MKHE_H32_PREFETCH && job2 + (int)gridDim.x < njobs) nsrc = src_of(job_of(job2 + (int)gridDim.x));
        tail(x, jb, nsrc, lds, wv);
#ifdef MKHE_PHASE_TRACE
        if (jb.trace && ((int)threadIdx.x & 63) == 0) jb.trace[(long)wv * 32 + 13] = __builtin_amdgcn_s_memrealtime();
#endif
    };
    // the launcher lists the slots of the balanced path first (59/60-bit moduli, then the ones between the classes) and the U class last, and the
    // placement above keeps that order inside every workgroup's walk: two loops with ONE instantiation of the limb each -- in one loop with both,
    // the 32 registers in flight across the back edge meet a two-way merge, and the allocator moves (spills) them there
    auto is_u = [&](const int job2) {
        kargptr kb = (kargptr)__builtin_amdgcn_kernarg_segment_ptr();
        const int s = (int)udiv_magic((unsigned)job_of(job2), kb->magic_nouter);
        return ((kb->u_mods >> kb->mod[s]) & 1) != 0;
    };
    int job2 = blockIdx.x;
#pragma unroll 1
    for (; job2 < njobs && !is_u(job2); job2 += gridDim.x) body(job2, std::false_type{});
#if MKHE_H32_PREFETCH
    landed();
#endif
#pragma unroll 1
    for (; job2 < njobs; job2 += gridDim.x) body(job2, std::true_type{});
#if MKHE_H32_PREFETCH
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the loads behind the last job's stores)
#endif
}

}  // namespace h32

// ------------------------------------------------------------------ launcher
namespace {
struct LaunchState32 { std::mutex mu; int resident[64] = {}; };
unsigned magic_of32(int d) { return d > 1 ? (unsigned)((1ull << 32) / (unsigned)d + 1) : 0u; }
// CU count of the CURRENT device, cached per device behind a mutex (a process may hold contexts on several GPUs, driven by several threads)
int device_cus32() {
    static std::mutex mu; static int cus_of[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    int& c = cus_of[dev & 63];
    if (!c) { int v = 256; (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev); c = v > 0 ? v : 256; }
    return c;
}
}
int ntt32_mode() { static const int on = MKHE_CFG_INT("MKHE_NTT32", MKHE_NTT32_DEFAULT); return on; }
bool ntt32_ok(int logN, const NttBatch& b) {
    static const int on = MKHE_CFG_INT("MKHE_NTT32", MKHE_NTT32_DEFAULT), minl = MKHE_AB_INT("MKHE_NTT32_MIN", 512);
    if (!on || logN != 15 || b.no_h16 || !b.psi31 || !b.psi31c || !b.psi31b || b.split || b.prestaged || b.nslots > 64) return false;
    if ((long)b.nslots * b.nouter >= 65536 || b.nouter >= 65536) return false;      // (job-walk reciprocals: exact below 2^16)
    // (one workgroup per CU, whole limbs: a launch that does not deal the limbs evenly leaves CUs idle in its last round, where the H16 kernel deals
    // half-limb jobs -- which kernel a shape takes is measured, Context::ntt_pick)
    return b.nslots * b.nouter >= minl;
}
void launch_ntt32_fwd(const NttBatch& b, const unsigned char* small_q, hipStream_t st) {
    using namespace h32;
    NttBatch c = b;
    // big-modulus limbs (the longer jobs) first, as in launch_ntt16_fwd
    // ... then the moduli between the classes, the U class last (the kernel runs the two instantiations of its limb in two loops, in this order)
    c.small_slots = 0; c.nslots = 0;
    for (int cls = 0; cls < 3; ++cls)
        for (int s = 0; s < b.nslots; ++s) {
            const bool small = small_q[b.mod[s]] != 0, u = small && ((b.u_mods >> b.mod[s]) & 1);
            if ((cls == 0 && !small) || (cls == 1 && small && !u) || (cls == 2 && u)) {
                c.mod[c.nslots] = b.mod[s]; c.pos[c.nslots] = b.pos[s];
                if (cls) c.small_slots |= 1ull << c.nslots;
                ++c.nslots;
            }
        }
    if (c.nslots != b.nslots) throw std::runtime_error("mkhe: internal: a U-class modulus outside the small class");
    const size_t lds = (size_t)LDS_WORDS * sizeof(u32);
    static LaunchState32 ls;
    int dev = 0;
    (void)hipGetDevice(&dev);
    int resident = 0;
    {
        std::lock_guard<std::mutex> g(ls.mu);
        if (!ls.resident[dev & 63]) {
            (void)hipFuncSetAttribute((const void*)ntt32_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)ntt32_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            ls.resident[dev & 63] = cus;                      // one workgroup per CU (133 KB of LDS, 1024 threads x 128 VGPRs)
        }
        resident = ls.resident[dev & 63];
    }
    const int need = c.nslots * c.nouter;
    const int blocks = need < resident ? need : resident;
    int nbig = 0;
    for (int s2 = 0; s2 < c.nslots; ++s2) if (!((c.small_slots >> s2) & 1)) ++nbig;
    const int lpt_long = nbig < c.nslots ? nbig * c.nouter : 0;
    c.magic_nouter = magic_of32(c.nouter);
    c.magic_opi = magic_of32(c.nitems > 0 ? c.outers_per_item : 1);
    c.lpt = NttBatch::Lpt{};
    c.half_jobs = 0; c.lazy_out = 0;
    const int C = blocks;
    if (lpt_long > 0 && C > 0 && need > blocks && need % C != 0) {
        const int r = need % C, w = C - r, q = need / C;
        const int B = lpt_long < q * w ? lpt_long : q * w;
        c.lpt.B = B; c.lpt.C = C; c.lpt.r = r; c.lpt.full = B / w; c.lpt.rem = B - (B / w) * w; c.lpt.magic_C = magic_of32(C);
    }
    if (c.reduce_in) hipLaunchKernelGGL(ntt32_fwd_kernel<true>, dim3(blocks), dim3(NT), lds, st, c);
    else hipLaunchKernelGGL(ntt32_fwd_kernel<false>, dim3(blocks), dim3(NT), lds, st, c);
}

}  // namespace mkhe
