// ntt16_f2_kernels.hip -- step F2 of MulAndRelin inside the Decompose NTT of the t_i, N = 2^15, alpha = 1 (round 6).
//
// keyswitch_hoisted.go:161-178: for every party i of op0, t_i = <h(c0_i), y>_P is decomposed again and its digits meet the party's v_i (into out_0) and
// the CRS u (into out_i).  Those digits -- beta x (l + p) limbs per party, 470 MB for four parties on PN15QP880 -- were written by the Decompose NTT
// and read back exactly once by a streaming inner-product launch.  Here they stay in registers: a workgroup (1024 threads, 16 coefficients per thread, as
// a pass of the H16 kernel of ntt16_kernels.hip) owns a run of digits of ONE (party, modulus, half limb), runs the forward pass of digit after digit --
// stage 0 recomputed from the source limb, the four register phases A..D with the re-distributions of h16_core.h, same twiddles for the whole run -- and
// multiplies what it holds with the two key words of the coefficient (mm of h16_arith.h: the two-round signed product, keys are plain Montgomery
// residues) into two signed 64-bit accumulators per coefficient.  128 VGPRs: 32 data + 64 accumulators + the twiddle / key rings; one workgroup per CU.
// At the end of a run the accumulators are brought to [0, q) and stored as one PART of the two products (F2FusedArgs); the inverse NTT adds the parts.
//
// Integers: a part is the canonical residue of sum_d NTT(digit d) (.) key[d] over its digits d -- the sum ext_inner_kernel forms from the stored digits
// with canonical modular additions -- and the parts of a product add up canonically at the load of the inverse transform: bit for bit the same input
// to InvNTTLazy + ModDown as the unfused launches (the digits' own representatives never leave the kernel).
//
// Ranges: the digits leave phase D un-normalised and UNBIASED (the bias of the stored engine-internal digits exists for the unsigned products of their
// consumers): U class |x| < 79 q < 2^61, moduli between the classes |x| < 20 q, 59/60-bit primes |x| < 2^62.9 by their reduction schedule -- all inside
// mm's domain (|a| < 2^63 - 2^31).  A product is below q in magnitude; the accumulators of a 59/60-bit modulus are partially reduced every third digit
// (|acc| < 3.5 q < 2^62), the others every 32nd.
//
// Replaces: mkrlwe/keyswitch.go:21-31,49-73 (Decompose of t_i) + keyswitch_hoisted.go:10-35 (the two inner products of step F2).
#include "ntt_kernels.h"
#include "h16_arith.h"
#include "h16_core.h"
#include <mutex>
#include <stdexcept>

namespace mkhe {
namespace h16 {

typedef const __attribute__((address_space(4))) F2FusedArgs* f2argptr;

#ifndef MKHE_F2_KRING
#define MKHE_F2_KRING 3            // key-word pairs (v, u) in flight per thread in the product phase (4 with SP = 12: the allocator spills inside the digit loop)
#endif
constexpr int KRING = MKHE_F2_KRING;
// One workgroup owns the CU: its sixteen waves leave the cross-wave exchange together and, four to a SIMD in round-robin, reach every per-lane
// twiddle load of phases C / D and every key load at the same time (tools/f2_trace.py: phase C takes 11 300 cycles per wave for 6 600 of issue).
// MKHE_F2_PHPRIO = 2: behind that exchange the lower eight waves (two per SIMD: waves w and w + 4 k share one) run at a higher priority than the
// upper eight, so that one pair's memory waits fall under the other pair's butterflies (the H32 kernel's remedy, ntt32_kernels.hip)
// The per-lane twiddle pairs of phase C (fifteen 16-byte pairs per group of four lanes) are the same for every digit of a run -- same modulus, same
// half, same wave: fetched once per run into LDS (61 KB beside the 68 KB of the re-distribution image: one workgroup owns the CU's 160 KB) and read
// from there by ds_read_b128, instead of fifteen L2 round trips per lane and pass with three of them in flight.  Wave-local: no barrier.
#ifndef MKHE_F2_TWLDS
#define MKHE_F2_TWLDS 1
#endif
constexpr int TWC_WORDS = MKHE_F2_TWLDS ? 16 * 16 * 15 * 4 : 0;      // u32 words: waves x lane groups x pairs x 16 B
#ifndef MKHE_F2_PHPRIO
#define MKHE_F2_PHPRIO 2
#endif

// The source limb of a pass -- 32 words per thread, L2 / Infinity Cache hits mostly: 32 workgroups read every t_i limb -- is what the one resident
// workgroup waited for longest (tools/ab_libs.sh: 192 -> 168 us with the loads compiled out).  MKHE_F2_SPIPE: the (x[j], x[j + N/2]) pairs of the NEXT
// digit are requested from inside the product phase of this one, as its registers come free: a consumed coefficient returns two, a consumed key pair
// four -- twelve of the sixteen pairs are in flight when the pass begins, the last four follow behind every second stage-0 product.  Every wait is
// counted (memory operations of a wave complete in order); the last digit of a run requests its own limb again (nobody reads them: drained at the flush).
#ifndef MKHE_F2_SPIPE
#define MKHE_F2_SPIPE 1
#endif
#ifndef MKHE_F2_SP
#define MKHE_F2_SP 10
#endif
constexpr int SP = MKHE_F2_SP;                         // source pairs in flight at the top of a pass (8 .. 12)
static_assert(SP >= 8 && SP <= 12, "eight pairs ride on the registers of the consumed coefficients, up to four more on those of the key ring");
// stage 0: one more pair behind every second product; the rest when the first eight results have gone to LDS (behind pair 7)
constexpr int s0_issued(int j) { return j >= 8 ? 16 : (SP + j / 2 > 16 ? 16 : SP + j / 2); }      // pairs requested by the time stage 0 waits for pair j
constexpr int mac_src_cnt(int i) { return ((i & 1) ? 1 : 0) + (i >= 16 - (SP - 8) ? 1 : 0); }      // source pairs requested behind the products of coefficient i
constexpr int mac_src_first(int i) { int n = 0; for (int k = 0; k < i; ++k) n += mac_src_cnt(k); return n; }      // ... the first of them: pairs are requested in ascending order
static_assert(mac_src_first(16) == SP, "the product phase requests exactly the pairs a pass finds in flight");
constexpr int mac_younger_src(int r, int kring) { int n = 0; for (int i = (r - kring > 0 ? r - kring : 0); i < r; ++i) n += mac_src_cnt(i); return n; }
#if !defined(MKHE_ABLATION) && (defined(MKHE_F2_X_NOKEYS) || defined(MKHE_F2_X_NOMAC) || defined(MKHE_F2_X_NOPASS))
#error "MKHE_F2_X_* switches give wrong results on purpose (timing experiments): build them with -DMKHE_ABLATION"
#endif
__device__ __forceinline__ u64 ld_issue_nt(gcptr base, unsigned byte_off) {
#ifdef MKHE_F2_X_NOKEYS         // MKHE_ABLATION: timing experiment only (wrong results): no key loads
    return (u64)byte_off * 0x9E3779B97F4A7C15ull + (u64)base;
#endif
    u64 v; asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(v) : "v"(byte_off), "s"(base)); return v;
}
__device__ __forceinline__ u64 ld_issue_key(gcptr base, unsigned byte_off) {
#ifdef MKHE_F2_X_NOKEYS
    return (u64)byte_off * 0x9E3779B97F4A7C15ull + (u64)base;
#endif
    return ld_issue(base, byte_off);
}
// the key word in the signed-split form mm() takes (modarith.h sd_split): the high word absorbs the carry of reading the low word as signed
__device__ __forceinline__ u64 key_split(u64 k) { return ((u64)(hi32(k) + (lo32(k) >> 31)) << 32) | lo32(k); }

// One forward pass of 2^14 points: half h of the limb `src` (a digit of a foreign modulus) under the modulus of c -- stage 0 through phase D of
// limb<true, false, UC, 15> (ntt16_kernels.hip), out of place, results left in x in the layout of phase D, un-normalised.
// diagnostic build (make trace): shader-clock stamps per (workgroup, wave, pass), 16 words each: tools/f2_trace.py
#ifdef MKHE_PHASE_TRACE
#define F2_STAMP(k) do { if (tr) tr[(k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define F2_STAMP(k) do { } while (0)
#endif
template <bool UC, bool PIPE>
__device__ __forceinline__ void f2_pass(u64* tr, const gcptr src, const MC& c, const bool big_, const bool red, const int sched, scptr p31, scptr pn, gcptr psi_v, gcptr p31v,
                                        u32* lds, const int wv, const int h, u64 (&x)[16], u64 (&PU)[SP], u64 (&PV)[SP]) {
    const bool big = UC ? false : big_;
    (void)tr;
    F2_STAMP(0);
    if (MKHE_H16_PRIO && MKHE_F2_PHPRIO < 2) __builtin_amdgcn_s_setprio(MKHE_H16_PRIO);
    // ---- stage 0: the cross-half butterflies, this pass's output only (x + w y for h = 0, x - w y = x + (-w) y for h = 1)
    {
        u64 w1[2] = {p31[2], p31[3]};
        if (h != 0) {
            if constexpr (UC) { w1[0] = pn[2]; w1[1] = pn[3]; }
            else {
                w1[0] = ((u64)(u32)(0 - (i32)hi32(w1[0])) << 32) | (u32)(0 - (i32)lo32(w1[0]));
                w1[1] = ((u64)(u32)(0 - (i32)hi32(w1[1])) << 32) | (u32)(0 - (i32)lo32(w1[1]));
            }
        }
        const unsigned tb = 8u * (unsigned)(wv * 64 + lane_id());
        typedef __attribute__((address_space(3))) u64* lptr64;
        typedef volatile __attribute__((address_space(3))) u64* vlptr64;
        if constexpr (PIPE) {
        // pairs 0 .. SP-1 are in flight (f2_run); pairs SP .. 15 are requested behind every second product
        u64 QU[16 - SP], QV[16 - SP];
        static_for(std::make_integer_sequence<int, 16>{}, [&](auto jc) {
            constexpr int j = decltype(jc)::value;
            u64& U = j < SP ? PU[j < SP ? j : 0] : QU[j >= SP ? j - SP : 0];
            u64& V = j < SP ? PV[j < SP ? j : 0] : QV[j >= SP ? j - SP : 0];
            ld_wait_pair<2 * (s0_issued(j) - j - 1)>(U, V);
            if ((big && (sched & 1)) || red) { U = (u64)pred((i64)U, c); V = (u64)pred((i64)V, c); }
            const i64 T = UC ? mm30u<true>((i64)V, w1[0], w1[1], c) : mm31<true>((i64)V, w1[0], w1[1], c);
            x[j] = (u64)((i64)U + T);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (j == 7) {
                // (the first eight results wait in the wave's own LDS region while the other pairs land: see limb())
                lptr64 st = (lptr64)((__attribute__((address_space(3))) u32*)lds + wv * WSTR) + lane_id();
#pragma unroll
                for (int k = 0; k < 8; ++k) { st[k * 64] = x[k]; }
                asm volatile("" ::: "memory");
            }
            if constexpr (j < 15 && s0_issued(j + 1) > s0_issued(j)) {
                // the next pairs, requested now that products (or the stash) have returned their registers
                static_for(std::make_integer_sequence<int, s0_issued(j < 15 ? j + 1 : j) - s0_issued(j)>{}, [&](auto kc) {
                    constexpr int n = s0_issued(j) + decltype(kc)::value;
                    QU[n - SP] = ld_issue(sbk(src, n * NT), tb); QV[n - SP] = ld_issue(sbk(src, HH + n * NT), tb);
                });
            }
        });
        {
            vlptr64 st = (vlptr64)((__attribute__((address_space(3))) u32*)lds + wv * WSTR) + lane_id();
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = st[r * 64];
        }
        } else {
        (void)PU; (void)PV;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += SG) {
            u64 U[SG], V[SG];
#pragma unroll
            for (int r = 0; r < SG; ++r) { U[r] = ld_issue(sbk(src, (r0 + r) * NT), tb); V[r] = ld_issue(sbk(src, HH + (r0 + r) * NT), tb); }
            ld_wait(U, V);
            if ((big && (sched & 1)) || red) {
#pragma unroll
                for (int r = 0; r < SG; ++r) { U[r] = (u64)pred((i64)U[r], c); V[r] = (u64)pred((i64)V[r], c); __builtin_amdgcn_sched_barrier(0); }
            }
#pragma unroll
            for (int r = 0; r < SG; ++r) {
                const i64 T = UC ? mm30u<true>((i64)V[r], w1[0], w1[1], c) : mm31<true>((i64)V[r], w1[0], w1[1], c);
                x[r0 + r] = (u64)((i64)U[r] + T);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("" ::: "memory");
            // (the first group's results wait in the wave's own LDS region while the second group's loads are in flight: see limb())
            if (r0 == 0) {
                lptr64 st = (lptr64)((__attribute__((address_space(3))) u32*)lds + wv * WSTR) + lane_id();
#pragma unroll
                for (int r = 0; r < SG; ++r) { st[r * 64] = x[r]; }
                asm volatile("" ::: "memory");
            }
        }
        {
            vlptr64 st = (vlptr64)((__attribute__((address_space(3))) u32*)lds + wv * WSTR) + lane_id();
#pragma unroll
            for (int r = 0; r < SG; ++r) x[r] = st[r * 64];
        }
        }
    }
    F2_STAMP(1);
    // ---- phase A: bits 13..10, twiddles psi[2^k + (h << (k-1)) + i], k = 1..4 (root 2 + h)
    {
        u64 tw[6], tm4[8], ta[8], tb[8];
#pragma unroll
        for (int i = 0; i < 2; ++i) tw[i] = p31[2 * (2 + h) + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) tw[2 + i] = p31[2 * (4 + 2 * h) + i];
        stage31<UC, 3>(x, tw, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) tm4[i] = p31[2 * (8 + 4 * h) + i];
        stage31<UC, 2>(x, tw + 2, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) ta[i] = p31[2 * (16 + 8 * h) + i];
        stage31<UC, 1>(x, tm4, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) tb[i] = p31[2 * (16 + 8 * h + 4) + i];
        stage31<UC, 0, 0, 4>(x, ta, c);
        stage31<UC, 0, 4, 4>(x, tb, c);
    }
    reduce_all(x, c, big && (sched & 2));
    F2_STAMP(2);
    exchange<X_AB>(x, lds, wv);
    if (MKHE_H16_PRIO) __builtin_amdgcn_s_setprio(0);
    if (MKHE_F2_PHPRIO == 2) { if (wv < 8) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
    if (MKHE_F2_PHPRIO == 3) { if ((wv >> 2) & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); }
    if (MKHE_F2_PHPRIO == 4) { const int g = wv >> 2; if (g == 0) __builtin_amdgcn_s_setprio(3); else if (g == 1) __builtin_amdgcn_s_setprio(2); else if (g == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    F2_STAMP(3);
    // ---- phase B: bits 9..6
    {
        const int cb = 16 * h + wv;
        u64 tw[6], tm4[8], ta[8], tb[8];
#pragma unroll
        for (int i = 0; i < 2; ++i) tw[i] = p31[2 * (32 + cb) + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) tw[2 + i] = p31[2 * (64 + 2 * cb) + i];
        stage31<UC, 3>(x, tw, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) tm4[i] = p31[2 * (128 + 4 * cb) + i];
        stage31<UC, 2>(x, tw + 2, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) ta[i] = p31[2 * (256 + 8 * cb) + i];
        stage31<UC, 1>(x, tm4, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) tb[i] = p31[2 * (256 + 8 * cb + 4) + i];
        stage31<UC, 0, 0, 4>(x, ta, c);
        stage31<UC, 0, 4, 4>(x, tb, c);
    }
    reduce_all(x, c, big && (sched & 4));
    // ---- phase C: bits 5..2, per-lane pairs, RING of them in flight
    {
        const int lc = lane_id();
        const unsigned cu = (unsigned)((16 * h + wv) * 16 + (lc >> 2));
        __builtin_assume(cu < 512);
        u64 g[RING][2];
        typedef volatile __attribute__((address_space(3))) u64x2* ltw;
        ltw twl = (ltw)((__attribute__((address_space(3))) u32*)lds + LDS_WORDS) + (wv * 16 + (lc >> 2)) * 15;
        auto loadt = [&](int t) {
            if (MKHE_F2_TWLDS) { if (t < 15) { const u64x2 v = twl[t]; g[t % RING][0] = v.x; g[t % RING][1] = v.y; } return; }
            if (t == 0) ld2(g[0], (gcptr2)sbk(p31v, 2 * 512), cu);
            else if (t < 3) ld2(g[t % RING], (gcptr2)sbk(p31v, 2 * 1024) + (t - 1), 2 * cu);
            else if (t < 7) ld2(g[t % RING], (gcptr2)sbk(p31v, 2 * 2048) + (t - 3), 4 * cu);
            else if (t < 15) ld2(g[t % RING], (gcptr2)sbk(p31v, 2 * 4096) + (t - 7), 8 * cu);
        };
        loadt(0); loadt(1); if (RING > 3) loadt(2);
        F2_STAMP(4);
        exchange<X_BC>(x, lds, wv);
        F2_STAMP(5);
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            const int gi = n & 7;
            const int t = n < 8 ? 0 : n < 16 ? 1 + (gi >> 2) : n < 24 ? 3 + (gi >> 1) : 7 + gi;
            const int tp = n == 0 ? -1 : (n - 1 < 8 ? 0 : n - 1 < 16 ? 1 + (((n - 1) & 7) >> 2) : n - 1 < 24 ? 3 + (((n - 1) & 7) >> 1) : 7 + ((n - 1) & 7));
            if (t != tp) loadt(t + RING - 1);
            if (n < 8) bfly1_31<UC, 3>(x, gi, g[t % RING], c);
            else if (n < 16) bfly1_31<UC, 2>(x, gi, g[t % RING], c);
            else if (n < 24) bfly1_31<UC, 1>(x, gi, g[t % RING], c);
            else bfly1_31<UC, 0>(x, gi, g[t % RING], c);
        }
    }
    reduce_all(x, c, big && (sched & 8));
    // ---- phase D: bits 1..0 on the two-round product (8-byte twiddles), as limb()
    {
        const int ld = lane_id();
        const unsigned du = (unsigned)((16 * h + wv) * 64 + ld);
        __builtin_assume(du < 2048);
        // (a ring of three 16-byte groups: group G = twiddles 2 G, 2 G + 1 of the phase; with all six named the allocator has 24 registers to place)
        u64 g[3][2];
        auto loadg = [&](int k) {
            if (k < 2) ld2(g[k % 3], (gcptr2)sbk(psi_v, 8192) + k, 2 * du);
            else if (k < 6) ld2(g[k % 3], (gcptr2)sbk(psi_v, 16384) + (k - 2), 4 * du);
        };
        loadg(0); loadg(1); loadg(2);
        F2_STAMP(6);
        exchange<X_CD>(x, lds, wv);
        F2_STAMP(7);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int gi = n & 7;
            const int G = n < 8 ? (gi >> 2) : 2 + (gi >> 1);
            const int Gp = n == 0 ? -1 : (n - 1 < 8 ? (((n - 1) & 7) >> 2) : 2 + (((n - 1) & 7) >> 1));
            if (G != Gp && G >= 1) loadg(G + 2);             // group G - 1 is done: its slot takes group G + 2
            if (n < 8) bfly1<1>(x, gi, g[G % 3][(gi >> 1) & 1], c);
            else bfly1<0>(x, gi, g[G % 3][gi & 1], c);
        }
    }
}

template <int K> __device__ __forceinline__ void key_wait(u64& a, u64& b) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(K)); }

// a run of digits of one (party, modulus m, half h): nd passes, two accumulators per coefficient, one part of each product out
template <bool UC, bool PIPE>
__device__ __forceinline__ void f2_run(const F2Seg sg, u32* lds, const int wv) {
    f2argptr ka = (f2argptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const int m = ka->mod[sg.slot], h = sg.half, party = sg.party;
    smodptr mp = (smodptr)ka->mods + m;
    const u64 qs = mp->qs;
    MC c;
    c.q = mp->q; c.ninv = mp->ninv32;
    c.q0 = (i32)lo32(qs); c.q1 = (i32)hi32(qs);
    c.finv = __builtin_bit_cast(float, mp->finv);
    if constexpr (UC) {
        c.p0 = (i32)((u32)c.q << 2) >> 2;
        c.p1 = (i32)((c.q - (u64)(i64)c.p0) >> 30);
    } else {
        c.p0 = (i32)((u32)c.q << 1) >> 1;
        c.p1 = (i32)((c.q - (u64)(i64)c.p0) >> 31);
    }
    asm("" : "+s"(c.q0), "+s"(c.q1), "+s"(c.ninv), "+s"(c.p0), "+s"(c.p1));
    const bool big = UC ? false : ((ka->small_mask >> m) & 1) == 0;
    const int sched = (int)((((const __attribute__((address_space(4))) unsigned*)ka->sched)[m >> 2] >> (8 * (m & 3))) & 0xffu);
    if (MKHE_F2_TWLDS) {
        // this wave's phase-C pairs: lane l fetches pairs t = (l & 3), (l & 3) + 4, .. of its group of four lanes (pair t of the phase = pair
        // (512 << j) + (cu << j) + t - (2^j - 1) of the modulus's table, j = the stage: f2_pass)
        typedef __attribute__((address_space(3))) u64x2* ltw;
        const int l = lane_id();
        const unsigned cu = (unsigned)((16 * h + wv) * 16 + (l >> 2));
        ltw twl = (ltw)((__attribute__((address_space(3))) u32*)lds + LDS_WORDS) + (wv * 16 + (l >> 2)) * 15;
        gcptr2 tab = (gcptr2)(ka->psi31 + 2 * (long)m * NN);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = 4 * k + (l & 3);
            if (t < 15) {
                const int j = t == 0 ? 0 : t < 3 ? 1 : t < 7 ? 2 : 3;
                const unsigned idx = (512u << j) + (cu << j) + (unsigned)(t - ((1 << j) - 1));
                u64 g2[2];
                ld2(g2, tab, idx);
                u64x2 v; v.x = g2[0]; v.y = g2[1];
                twl[t] = v;
            }
        }
    }
    i64 av[16], au[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { av[r] = 0; au[r] = 0; }
    u64 PU[SP], PV[SP];
    if constexpr (PIPE) {
        // the first digit's first SP pairs
        const gcptr src0 = (gcptr)(ka->src[party] + (long)sg.d0 * NN);
        const unsigned tb = 8u * (unsigned)(wv * 64 + lane_id());
#pragma unroll
        for (int j = 0; j < SP; ++j) { PU[j] = ld_issue(sbk(src0, j * NT), tb); PV[j] = ld_issue(sbk(src0, HH + j * NT), tb); }
    }
#pragma unroll 1
    for (int dd = 0; dd < sg.nd; ++dd) {
        const int d = __builtin_amdgcn_readfirstlane(sg.d0 + dd);
        f2argptr kb = (f2argptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kb));
        // the digit's own modulus is its index (alpha = 1, Q limbs first): the rules of fwd_body for what is reduced at the load
        const u64 qsb = ((smodptr)kb->mods)[d].q;
        bool red;
        if (UC) red = qsb >= (1ull << 62) - 150 * c.q; else red = qsb > 4 * c.q;
        const gcptr src = (gcptr)(kb->src[party] + (long)d * NN);
        u64 x[16];
#ifdef MKHE_PHASE_TRACE
        u64* tr = kb->trace ? kb->trace + (((long)blockIdx.x * 16 + wv) * 16 + (sg.pad1 * 8 + dd)) * 16 : nullptr;
        if (tr && dd + sg.pad1 * 8 >= 16) tr = nullptr;
#else
        u64* tr = nullptr;
#endif
        f2_pass<UC, PIPE>(tr, src, c, big, red, sched, (scptr)(kb->psi31 + 2 * (long)m * NN), (scptr)(kb->psi31n + 8 * (long)m), (gcptr)(kb->psi + (long)m * NN),
                    (gcptr)(kb->psi31 + 2 * (long)m * NN), lds, wv, h, x, PU, PV);
        // ---- the two products of the sixteen coefficients; key words in the store layout of the H16 pass (register r = words wave * 1024 + 64 r + lane)
        const long koff = (long)d * kb->digit_stride + (long)m * NN + h * HH + wv * 1024;
        const gcptr bv = (gcptr)(kb->kv[party] + koff), bu = (gcptr)(kb->ku + koff);
        const unsigned lb = 8u * (unsigned)lane_id();
        u64 kvr[16], kur[16];
        static_for(std::make_integer_sequence<int, KRING>{}, [&](auto rc) {
            constexpr int r = decltype(rc)::value;
            kvr[r] = ld_issue_nt(sbk(bv, r * 64), lb); kur[r] = ld_issue_key(sbk(bu, r * 64), lb);
        });
        F2_STAMP(8);
        exchange<X_DE>(x, lds, wv);
        F2_STAMP(9);
        // (the next digit's source limb: the digits of a party are consecutive limbs of its t; behind the last digit of the run this one's again)
        const gcptr nsrc = (gcptr)(kb->src[party] + (long)(dd + 1 < (int)sg.nd ? d + 1 : d) * NN);
        const unsigned tbn = 8u * (unsigned)(wv * 64 + lane_id());
        static_for(std::make_integer_sequence<int, 16>{}, [&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int younger = (r + KRING < 16 ? r + KRING : 16) - (r + 1) + (PIPE ? mac_younger_src(r, KRING) : 0);
            key_wait<2 * younger>(kvr[r], kur[r]);
            if constexpr (r + KRING < 16) { kvr[r + KRING] = ld_issue_nt(sbk(bv, (r + KRING) * 64), lb); kur[r + KRING] = ld_issue_key(sbk(bu, (r + KRING) * 64), lb); }
            const i64 a = (i64)x[r];
#ifdef MKHE_F2_X_NOMAC          // MKHE_ABLATION: timing experiment only (wrong results): the products replaced by one addition each
            av[r] += a + (i64)kvr[r]; au[r] += a + (i64)kur[r];
#else
            av[r] += mm<false>(a, key_split(kvr[r]), c);
            __builtin_amdgcn_sched_barrier(0);
            au[r] += mm<false>(a, key_split(kur[r]), c);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PIPE && mac_src_cnt(r) >= 1) { constexpr int j = mac_src_first(r); PU[j] = ld_issue(sbk(nsrc, j * NT), tbn); PV[j] = ld_issue(sbk(nsrc, HH + j * NT), tbn); }
            if constexpr (PIPE && mac_src_cnt(r) >= 2) { constexpr int j = mac_src_first(r) + 1; PU[j] = ld_issue(sbk(nsrc, j * NT), tbn); PV[j] = ld_issue(sbk(nsrc, HH + j * NT), tbn); }
        });
        F2_STAMP(10);
#ifdef MKHE_PHASE_TRACE
        if (tr) { tr[11] = __builtin_amdgcn_s_memrealtime(); tr[12] = (u64)m; tr[13] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
#endif
        // accumulator ranges (see the head of the file)
        const bool fold = big ? (dd % 3 == 2) : ((dd & 31) == 31);
        if (fold) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { av[r] = pred(av[r], c); au[r] = pred(au[r], c); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    // (the pairs requested behind the last digit: nobody reads them, but their registers are theirs until they have landed)
    if constexpr (PIPE) asm volatile("s_waitcnt vmcnt(0)" : "+v"(PU[0]), "+v"(PU[1]), "+v"(PU[2]), "+v"(PU[3]), "+v"(PU[4]), "+v"(PU[5]), "+v"(PU[6]), "+v"(PU[7]), "+v"(PU[8]), "+v"(PU[9]), "+v"(PU[10]), "+v"(PU[11]),
                 "+v"(PV[0]), "+v"(PV[1]), "+v"(PV[2]), "+v"(PV[3]), "+v"(PV[4]), "+v"(PV[5]), "+v"(PV[6]), "+v"(PV[7]), "+v"(PV[8]), "+v"(PV[9]), "+v"(PV[10]), "+v"(PV[11]));
    // ---- the part: canonical residues, stored where a Decompose pass stores its digits
    {
        f2argptr kb = (f2argptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kb));
        const int iv = sg.part == 0 ? kb->item_v[party] : kb->extra_v[party] + (sg.part - 1);
        const int iu = sg.part == 0 ? kb->item_u[party] : kb->extra_u[party] + (sg.part - 1);
        const long ooff = (long)m * NN + h * HH + wv * 1024;
        const gptr ov = (gptr)(kb->c1 + (long)iv * kb->item_words + ooff), ou = (gptr)(kb->c1 + (long)iu * kb->item_words + ooff);
        const unsigned lb = 8u * (unsigned)lane_id();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            i64 y = pred(av[r], c);
            y += (y >> 63) & (i64)c.q;
            *at(sbk(ov, r * 64), lb) = (u64)y;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            i64 y = pred(au[r], c);
            y += (y >> 63) & (i64)c.q;
            *at(sbk(ou, r * 64), lb) = (u64)y;
            __builtin_amdgcn_sched_barrier(0);
        }
        // a group that is cut into fewer parts than the launch's products have: its last run zeroes the rest (F2Seg::pad0)
#pragma unroll 1
        for (int z = 1; z <= (int)sg.pad0; ++z) {
            const gptr zv = (gptr)(kb->c1 + (long)(kb->extra_v[party] + (sg.part + z - 1)) * kb->item_words + ooff);
            const gptr zu = (gptr)(kb->c1 + (long)(kb->extra_u[party] + (sg.part + z - 1)) * kb->item_words + ooff);
#pragma unroll
            for (int r = 0; r < 16; ++r) { *at(sbk(zv, r * 64), lb) = 0; *at(sbk(zu, r * 64), lb) = 0; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

__global__ void __launch_bounds__(NT, 4) ntt16_f2_kernel(F2FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#pragma unroll 1
    for (int si = 0; si < F2_SEGS; ++si) {
        f2argptr ka = (f2argptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const unsigned long long raw = ((const __attribute__((address_space(4))) unsigned long long*)ka->segs)[(long)blockIdx.x * F2_SEGS + si];
        F2Seg sg = __builtin_bit_cast(F2Seg, raw);
        if (sg.nd == 0) continue;
        sg.pad1 = (unsigned char)si;                 // (diagnostic builds: which run of the workgroup)
        const int m = ka->mod[sg.slot];
        // (the pipelined source loads hold twelve pairs in flight across the loop's back edge: the U class has the registers for that -- tools/check_inflight.py
        // finds no spill of a register in flight --, the balanced path with its digit fix-ups and partial reductions does not)
        if ((ka->u_mods >> m) & 1) f2_run<true, MKHE_F2_SPIPE != 0>(sg, lds, wv);
        else f2_run<false, false>(sg, lds, wv);
        __syncthreads();               // (every wave is done with the LDS regions before the next run's stage 0 stashes into them)
    }
}

}  // namespace h16

// ------------------------------------------------------------------ launcher
namespace {
struct LaunchStateF2 { std::mutex mu; int cus[64] = {}; };
LaunchStateF2& f2_state() { static LaunchStateF2 s; return s; }
int f2_device_cus() {
    LaunchStateF2& ls = f2_state();
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(ls.mu);
    int& c = ls.cus[dev & 63];
    if (!c) {
        (void)hipFuncSetAttribute((const void*)h16::ntt16_f2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)(h16::LDS_WORDS + h16::TWC_WORDS) * sizeof(u32)));
        int v = 256;
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        c = v > 0 ? v : 256;
    }
    return c;
}
}
int ntt16_f2_grid() { return f2_device_cus(); }
int ntt16_f2_mode() { static const int v = MKHE_AB_INT("MKHE_F2_FUSED", 1); return v; }      // 0: off; 1: the grid is planned (f2_plan_schedule); 2: one workgroup per CU or nothing (the first version)
bool ntt16_f2_ok(int logN, int nparties, int nb, int nslots) {
    if (!ntt16_f2_mode() || logN != 15 || nparties < 1 || nparties > F2_MAX_P || nb < 1 || nb > 255 || nslots < 1 || nslots > NTT_MAX_SLOTS) return false;
    return true;
}
void launch_ntt16_f2(const F2FusedArgs& a, hipStream_t st) {
    using namespace h16;
    (void)f2_device_cus();
    const size_t lds = (size_t)(LDS_WORDS + TWC_WORDS) * sizeof(u32);
    if (a.nwg < 1) return;
    hipLaunchKernelGGL(ntt16_f2_kernel, dim3(a.nwg), dim3(NT), lds, st, a);
}

}  // namespace mkhe
