// ntt16_kernels.hip -- forward negacyclic NTT, N = 2^15, for the large Decompose launches of a MulRelin ("H16" form).
//
// Why a second forward kernel (docs/DESIGN_HISTORY.md section 4, round 2): the register-resident kernel of ntt_kernels.hip
// holds a whole limb in ONE workgroup per CU (1024 threads x 32 coefficients x 128 VGPRs).  Its three resources --
// VALU (butterflies), LDS (re-distribution) and the memory pipe (loads / stores) -- are then used one after the
// other, because the 16 waves of the only resident workgroup move through the same phases together: 36 us + 9 us +
// 10 us per limb and CU = the 55 us that kernel measures.  Here a thread holds 16 coefficients (64 VGPRs), so TWO
// workgroups are resident per CU (8 waves per SIMD) and one workgroup's LDS / memory phases run under the other
// one's butterflies.
//
// A limb is transformed in two passes by the same workgroup (1024 threads):
//   pass 0: load x[j], x[j + N/2]; stage 0 (the cross-half butterflies, twiddle psi[1]); keep the lower outputs; then the
//           14 remaining stages of the lower half;
//   pass 1: out of place (every Decompose launch) the same loads and stage 0 again, keeping the upper outputs (16 products
//           per thread repeated instead of half a limb written to HBM and read back); in place (src == dst) pass 0 parks the
//           upper outputs in the upper half of the destination limb and pass 1 reloads them; then the same 14 stages.
// The 14 stages of a half (2^14 points, index bits 13..0) run as four register phases with LDS re-distributions:
//   A: bits 13..10 in registers, thread = bits 9..0          twiddles uniform per workgroup  (scalar loads)
//   B: bits  9..6,  wave = bits 13..10, lane = bits 5..0      twiddles uniform per wave       (scalar loads)
//   C: bits  5..2,  lane = (bits 9..6, bits 1..0)             twiddles per lane (vector loads, 16-B)
//   D: bits  1..0 (registers hold bits 3..0), lane = (bits 9..6, bits 5..4)
//   E: store layout: registers = bits 9..6, lane = bits 5..0 (512 B contiguous per store instruction)
// Only A -> B crosses waves (4 workgroup barriers per pass); B -> C, C -> D, D -> E stay inside a wave's own 1024
// coefficients and need no barrier at all.  LDS image: one 32-bit plane of the half at a time (68 KiB per workgroup,
// 136 KiB per CU), every layout addressed as base(thread) + immediate(register), conflict-free under the 32-bank rule
// of ds_read_b32 / ds_write_b32 (padding constants below).
//
// Arithmetic: signed never-reduced butterflies for BOTH modulus classes on the one-round product mm31 (stage 0, phases A, B, C: the
// twiddle as the pair w 2^31, w 2^63 mod q, one Montgomery round of radix 2^31, 8 + 1 instructions) and on mont_mul_sd of
// modarith.h (phase D: two rounds, 12 + 2); the 59/60-bit primes get a 7-instruction float-estimated partial reduction after every
// phase (pred() below) where ntt_kernels.hip runs Harvey butterflies.  Same DEC digit reduction, same skip_norm / canonical outputs,
// so the two kernels are interchangeable bit for bit on canonical outputs (internal lazy representatives differ).
//
// Round 3: two instantiations of the limb body -- the U class (moduli with 160 q < 2^62: unsigned low data digit, one-round product of radix
// 2^30 "mm30u", no reductions) and the balanced path above for the 59/60-bit primes, whose partial reductions follow a per-modulus schedule
// (NttBatch::sched); lane-linear LDS writes as ds_write_addtid_b32; the same body as ntt14_fwd_kernel for N = 2^14 (one pass per limb).  The
// kernel runs at the package power cap (tools/power_probe.sh); docs/DESIGN_HISTORY.md section 3 "Round 3" has the steady-state ablation of its two sides.
//
// Replaces: lattigo ring.NTTLvl as called from DecomposeSingleNTT (mkrlwe/keyswitch.go:21-31,49-73).
#include "ntt_kernels.h"
#include "h16_arith.h"
#include "h16_core.h"
#include <cstdlib>
#include <mutex>
#include <utility>
#include <stdexcept>

namespace mkhe {
namespace h16 {

// ------------------------------------------------------------------ one limb
// diagnostic build (make trace): shader-clock stamps per wave and pass, 32 words per (job, wave): [16 * pass + k], see tools/ntt16_trace.py
// (every lane stores the same word: a lane-0 branch here makes the compiler lose the uniformity of the scalar twiddle loads)
#ifdef MKHE_PHASE_TRACE
#define H16_STAMP(k) do { if (jb.trace) jb.trace[(long)wv * 32 + 16 * h + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define H16_STAMP(k) do { } while (0)
#endif

// big: modulus class 0 (q up to 2^60, partial reductions interposed); one instantiation serves both classes (wave-uniform
// branches around the reductions), which halves the code the two workgroups of a CU -- and the neighbouring CU that shares the
// instruction cache -- stream through
// SPLIT: the 2^15 points are one half of a 2^16-point limb whose cross-half stage has already been applied (NttBatch::split): the
// group i' of local stage k then uses the twiddle psi16[root * 2^k + i'], root = 2 + half, where a whole limb uses psi[2^k + i'].
// UC: the modulus is of the U class (160 q < 2^62): mm30u, twiddle pairs in the unsigned radix-2^30 format, no reductions anywhere, every
// stage (phase D included) on the one-round product.  The other instantiation serves the 59/60-bit primes (`big`) and the moduli in between.
// LOGN = 14 (round 3): a 2^14-point limb IS one pass of this kernel -- 1024 threads x 16 coefficients, no cross-half stage, the four register
// phases with the twiddles of root 1 (psi[2^k + i]): in the index formulas below (root 2 tm + h of a 2^15-point limb's half h) that is
// tm = 0 with the twiddle half ht = 1.
template <bool DEC, bool SPLIT, bool UC, int LOGN = 15>
__device__ __forceinline__ void limb(const Job& jb, const bool big_, u32* lds, const int wv, const int h_first = 0, const int h_last = 1) {
    static_assert(LOGN == 15 || LOGN == 14, "H16 covers N = 2^15 and N = 2^14, whole limbs or the halves / quarters of a split N = 2^16 limb");
    const bool big = UC ? false : big_;
    // LOGN = 14 + SPLIT (round 3): the 2^14 points are one QUARTER of a 2^16-point limb whose two cross stages have been applied by the producer
    // (decomp_spread_kernel<2>): sub-transform root 4 + quarter = 2 tm + ht in the index formulas below
    const int tm = LOGN == 14 ? (SPLIT ? (jb.root >> 1) : 0) : (SPLIT ? jb.root : 1);
    smodptr mp = jb.mp;                                 // scalar loads: the constants live in SGPRs
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
    asm("" : "+s"(c.q0), "+s"(c.q1), "+s"(c.ninv), "+s"(c.p0), "+s"(c.p1));        // opaque wave-uniform 32-bit values (see modarith.h mont_mul_sd)
    scptr psi_s = (scptr)jb.psi;
    scptr p31 = (scptr)jb.psi31;                            // (u, v) of twiddle i at words 2i, 2i + 1
    gcptr psi_v = (gcptr)jb.psi;
    const gcptr src = jb.src; const gptr dst = jb.dst;
    const bool red = (DEC || (SPLIT && LOGN == 14)) && jb.red;
    u64 x[16];
#pragma unroll 1
    for (int hh = (LOGN == 15 ? h_first : 0); hh <= (LOGN == 15 ? h_last : 0); ++hh) {
        const int h = __builtin_amdgcn_readfirstlane(hh);
        const int ht = LOGN == 14 ? (SPLIT ? (jb.root & 1) : 1) : h;              // the half as the twiddle indices see it
        H16_STAMP(0);
        // The part of a pass that ends in the four workgroup barriers of the A -> B exchange runs at raised wave priority, so that the sixteen
        // waves of a workgroup reach those barriers together while the co-resident workgroup's waves, if they are in the barrier-free phases
        // B .. E, yield (profiles/r2_ntt16_phase_trace.txt: a wave spends a quarter of a pass waiting at those barriers): 264.9 -> 262.4 us
        // for the 1792-limb launch, 143.5 -> 140.6 us for the 896-limb one (MKHE_H16_PRIO=0 switches it off)
        if (MKHE_H16_PRIO) __builtin_amdgcn_s_setprio(MKHE_H16_PRIO);
        // ---- stage 0: the cross-half butterflies.  Out of place (every Decompose launch: the source is a ciphertext limb) BOTH passes
        // load x[j], x[j + N/2] and keep their own output of the butterfly -- the second pass repeats 16 products per thread (1/15 of
        // the butterfly work) and re-reads the source, which 15 other workgroups read as well (L2 / Infinity Cache), instead of
        // parking the upper outputs in the destination limb in pass 0 and reloading them: 128 KiB written + 128 KiB read per limb,
        // a third of what this kernel moved through HBM, and memory is what bounds it once the butterflies are cheap
        // (tools/ubench + the MKHE_H16_X_* ablations: 245 us of a 367 us launch remain with every butterfly removed).
        // In place (src == dst) pass 0's results overwrite the source: the upper outputs are parked as before.
#ifdef MKHE_H16_FORCE_PARK      // experiment: park the upper stage-0 outputs in the destination limb also out of place (no recompute, no source re-read)
        const bool park = LOGN == 15;
#else
        const bool park = LOGN == 15 && (const void*)src == (const void*)dst;
#endif
        if constexpr (LOGN == 14) {
            // the whole limb: registers = bits 13..10, thread = bits 9..0
            const unsigned tb = 8u * (unsigned)(wv * 64 + lane_id());
            u64 lo8[8], hi8[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) { lo8[r] = ld_issue(sbk(src, r * NT), tb); hi8[r] = ld_issue(sbk(src, (8 + r) * NT), tb); }
            ld_wait16(lo8, hi8);
#pragma unroll
            for (int r = 0; r < 8; ++r) { x[r] = lo8[r]; x[8 + r] = hi8[r]; }
            if ((big && (jb.sched & 1)) || red) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { x[r] = (u64)pred((i64)x[r], c); __builtin_amdgcn_sched_barrier(0); }
            }
        } else
        if (h == 0 || !park) {
            u64 w1[2] = {p31[2 * tm], p31[2 * tm + 1]};
            if (h != 0) {
                // second pass: x - w y = x + (-w) y, the butterfly keeps its `+` output
                if constexpr (UC) {
                    // (the unsigned digits of u cannot be negated in place: the pair of -w mod q comes from its own small table, NttBatch::psi31n)
                    scptr pn = (scptr)jb.psi31n;
                    w1[0] = pn[2 * tm]; w1[1] = pn[2 * tm + 1];
                } else {
                    // both digits of both constants negated (scalar)
                    w1[0] = ((u64)(u32)(0 - (i32)hi32(w1[0])) << 32) | (u32)(0 - (i32)lo32(w1[0]));
                    w1[1] = ((u64)(u32)(0 - (i32)hi32(w1[1])) << 32) | (u32)(0 - (i32)lo32(w1[1]));
                }
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // (keeps the store queue of pass 0 from growing under the loads)
            }
            const unsigned tb = 8u * (unsigned)(wv * 64 + lane_id());
#ifdef MKHE_H16_PIPE
            if (!park) {
                // EXPERIMENT (make XFLAGS=-DMKHE_H16_PIPE, not the default): one memory round trip per pass instead of two.  Eleven of the sixteen
                // (x[j], x[j + N/2]) pairs are requested up front, the other five as registers come free (a consumed pair turns four VGPRs into
                // two; the first eight results wait in the wave's LDS region), and every pair is consumed behind a counted wait -- memory
                // operations of a wave complete in order, so `vmcnt(loads issued after the pair)` is exact (tools/check_inflight.py verifies on
                // the ISA that nothing touches a register of a load in flight: with twelve pairs up front the allocator spills one of them).
                // Measured in the steady state (2000 launches back to back): 276 us against 271 us for the two groups of eight below -- the
                // kernel runs at the 1400 W package power cap (tools/power_probe.sh: 1360 W, 2.06 GHz), where hiding latency buys nothing
                // and the 23 additional spilled SGPRs cost instructions.
                u64 U[16], V[16];
#pragma unroll
                for (int r = 0; r < PIPE_P; ++r) { U[r] = ld_issue(sbk(src, r * NT), tb); V[r] = ld_issue(sbk(src, HH + r * NT), tb); }
                static_for(std::make_integer_sequence<int, 16>{}, [&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    ld_wait_pair<2 * (pipe_issued(r) - r - 1)>(U[r], V[r]);
                    // digits of a foreign modulus (Decompose) may be far above q: bring them to (-q, q) first.  MODE 0 (q up to 2^60)
                    // has no headroom for five stages on raw inputs and always starts from reduced values.
                    if ((big && (jb.sched & 1)) || red) { U[r] = (u64)pred((i64)U[r], c); V[r] = (u64)pred((i64)V[r], c); }
                    const i64 T = UC ? mm30u<true>((i64)V[r], w1[0], w1[1], c) : mm31<true>((i64)V[r], w1[0], w1[1], c);
                    x[r] = (u64)((i64)U[r] + T);
                    __builtin_amdgcn_sched_barrier(0);       // one butterfly at a time
#ifndef MKHE_H16_NO_STASH
                    if constexpr (r == 7) {
                        typedef __attribute__((address_space(3))) u64* lptr64;
                        lptr64 st = (lptr64)((__attribute__((address_space(3))) u32*)lds + wv * WSTR) + lane_id();
#pragma unroll
                        for (int k = 0; k < 8; ++k) { st[k * 64] = x[k]; }
                        asm volatile("" ::: "memory");
                    }
#endif
                    if constexpr (pipe_issued(r + 1) > pipe_issued(r)) {
                        constexpr int n = pipe_issued(r);    // the next pair, requested now that its registers are free
                        U[n] = ld_issue(sbk(src, n * NT), tb); V[n] = ld_issue(sbk(src, HH + n * NT), tb);
                    }
                });
            } else
#endif
#pragma unroll
            for (int r0 = 0; r0 < 16; r0 += SG) {
                u64 U[SG], V[SG];
#pragma unroll
                for (int r = 0; r < SG; ++r) { U[r] = ld_issue(sbk(src, (r0 + r) * NT), tb); V[r] = ld_issue(sbk(src, HH + (r0 + r) * NT), tb); }
                ld_wait(U, V);
                if ((big && (jb.sched & 1)) || red) {
#pragma unroll
                    for (int r = 0; r < SG; ++r) { U[r] = (u64)pred((i64)U[r], c); V[r] = (u64)pred((i64)V[r], c); __builtin_amdgcn_sched_barrier(0); }
                }
#pragma unroll
                for (int r = 0; r < SG; ++r) {
                    const i64 T = UC ? mm30u<true>((i64)V[r], w1[0], w1[1], c) : mm31<true>((i64)V[r], w1[0], w1[1], c);
                    x[r0 + r] = (u64)((i64)U[r] + T);
#ifndef MKHE_H16_X_NOPARK      // timing experiment only (wrong results): no parking store / reload of the upper half
                    if (park) st_issue(sbk(dst, HH + (r0 + r) * NT), tb, (u64)((i64)U[r] - T));
#endif
                    __builtin_amdgcn_sched_barrier(0);       // one butterfly at a time: interleaved, their temporaries do not fit beside 16 loads in flight
                }
                asm volatile("" ::: "memory");
#ifndef MKHE_H16_NO_STASH
                // The first group's results wait in the wave's own LDS region (idle until the A -> B exchange) while the second group's
                // 16 loads are in flight: left in registers, the allocator sent six of them to scratch and back -- 140 KB of scratch
                // writes per limb through HBM (the scratch of 512 resident workgroups is 30 MB, more than the L2s hold).
                if (r0 == 0) {
                    typedef __attribute__((address_space(3))) u64* lptr64;
                    lptr64 st = (lptr64)((__attribute__((address_space(3))) u32*)lds + wv * WSTR) + lane_id();
#pragma unroll
                    for (int r = 0; r < SG; ++r) { st[r * 64] = x[r]; }
                    asm volatile("" ::: "memory");
                }
#endif
            }
#ifndef MKHE_H16_NO_STASH
            {
                typedef volatile __attribute__((address_space(3))) u64* lptr64;
                lptr64 st = (lptr64)((__attribute__((address_space(3))) u32*)lds + wv * WSTR) + lane_id();
#pragma unroll
                for (int r = 0; r < SG; ++r) x[r] = st[r * 64];
            }
#endif
        } else {
            // the parked half: written by this same thread in pass 0; all but the 16 youngest memory operations (the final
            // stores of pass 0) are complete before the reload is issued
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            const unsigned tb = 8u * (unsigned)(wv * 64 + lane_id());
#pragma unroll
            for (int r = 0; r < 16; ++r)
#ifndef MKHE_H16_X_NOPARK
                x[r] = __hip_atomic_load(at(sbk(dst, HH + r * NT), tb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                x[r] = x[r] * 3 + tb;
#endif
        }
        H16_STAMP(1);                            // loads landed, stage 0 done
        // ---- phase A: bits 13..10, twiddles psi[2^k + (h << (k-1)) + i], k = 1..4
        {
            // (the 8 twiddles of the last stage are fetched after the first two stages: 30 twiddle SGPRs at once do not fit the
            // 80-SGPR budget of 8 waves per SIMD beside the job state, and every spilled SGPR costs VALU lane moves)
            // (the twiddle pairs of a stage are fetched one stage ahead, the eight of the last stage in two halves: all 30 pairs at once
            // = 120 SGPRs do not fit the SGPR budget of 8 waves per SIMD beside the job state, and every spilled SGPR costs VALU lane moves)
            u64 tw[6], tm4[8], ta[8], tb[8];
#pragma unroll
            for (int i = 0; i < 2; ++i) tw[i] = p31[2 * (2 * tm + ht) + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) tw[2 + i] = p31[2 * (4 * tm + 2 * ht) + i];
            stage31<UC, 3>(x, tw, c);
#pragma unroll
            for (int i = 0; i < 8; ++i) tm4[i] = p31[2 * (8 * tm + 4 * ht) + i];
            stage31<UC, 2>(x, tw + 2, c);
#pragma unroll
            for (int i = 0; i < 8; ++i) ta[i] = p31[2 * (16 * tm + 8 * ht) + i];
            stage31<UC, 1>(x, tm4, c);
#pragma unroll
            for (int i = 0; i < 8; ++i) tb[i] = p31[2 * (16 * tm + 8 * ht + 4) + i];
            stage31<UC, 0, 0, 4>(x, ta, c);
            stage31<UC, 0, 4, 4>(x, tb, c);
        }
        // The 59/60-bit primes: partial reductions where the schedule of the modulus asks for them (NttBatch::sched, computed by the Context from
        // the headroom 2^62.9 / q: bit 0 at the load, bits 1..3 after phases A, B, C).  Round 2 reduced after every phase; the 60-bit head prime
        // needs it after A and B only (raw digits < 2^60 = q: 1 + 5 stages of 1.03q = 6.2q < 7.46q; then 0.5 + 4.1; C and D add 4.1 + 1.6), the
        // 59-bit special primes after A only.
        reduce_all(x, c, big && (jb.sched & 2));
        H16_STAMP(2);
        exchange<X_AB>(x, lds, wv);
        if (MKHE_H16_PRIO) __builtin_amdgcn_s_setprio(0);
        H16_STAMP(3);
        // ---- phase B: bits 9..6, twiddles psi[2^k + ((16h + wave) << (k-5)) + i], k = 5..8
        {
            const int cb = 16 * ht + wv;
            u64 tw[6], tm4[8], ta[8], tb[8];
#pragma unroll
            for (int i = 0; i < 2; ++i) tw[i] = p31[2 * (32 * tm + cb) + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) tw[2 + i] = p31[2 * (64 * tm + 2 * cb) + i];
            stage31<UC, 3>(x, tw, c);
#pragma unroll
            for (int i = 0; i < 8; ++i) tm4[i] = p31[2 * (128 * tm + 4 * cb) + i];
            stage31<UC, 2>(x, tw + 2, c);
#pragma unroll
            for (int i = 0; i < 8; ++i) ta[i] = p31[2 * (256 * tm + 8 * cb) + i];
            stage31<UC, 1>(x, tm4, c);
#pragma unroll
            for (int i = 0; i < 8; ++i) tb[i] = p31[2 * (256 * tm + 8 * cb + 4) + i];
            stage31<UC, 0, 0, 4>(x, ta, c);
            stage31<UC, 0, 4, 4>(x, tb, c);
        }
        reduce_all(x, c, big && (jb.sched & 4));
        // Phases C and D: per-lane twiddles, fetched in 16-byte groups a few butterflies ahead of their use (at most three groups =
        // 12 VGPRs live; issuing all 15 + 12 at once would not fit beside the 32 data registers).  The first groups of a phase are
        // requested before the re-distribution that precedes it.
        // ---- phase C: bits 5..2, twiddles psi[2^k + (cc << (k-9)) + i], k = 9..12, cc = (16h + wave) * 16 + bits 9..6 (lane >> 2)
        // Twiddle t of the phase (t = 0: k = 9; 1, 2: k = 10; 3..6: k = 11; 7..14: k = 12) is one 16-byte load of its (u, v) pair; the
        // butterflies use them in that order (8, 4, 4, 2, 2, 2, 2, 1 x 8 times), a ring of four pairs (16 VGPRs) keeps the next three in flight.
        {
            const int lc = lane_id();
            const unsigned cu = (unsigned)((16 * ht + wv) * 16 + (lc >> 2));
            __builtin_assume(cu < 512);
            u64 g[RING][2];
            gcptr p31v = (gcptr)jb.psi31;
            auto loadt = [&](int t) {
                if (t == 0) ld2(g[0], (gcptr2)sbk(p31v, 2 * 512 * tm), cu);
                else if (t < 3) ld2(g[t % RING], (gcptr2)sbk(p31v, 2 * 1024 * tm) + (t - 1), 2 * cu);
                else if (t < 7) ld2(g[t % RING], (gcptr2)sbk(p31v, 2 * 2048 * tm) + (t - 3), 4 * cu);
                else if (t < 15) ld2(g[t % RING], (gcptr2)sbk(p31v, 2 * 4096 * tm) + (t - 7), 8 * cu);
            };
            loadt(0); loadt(1); if (RING > 3) loadt(2);
            H16_STAMP(4);
            exchange<X_BC>(x, lds, wv);
            H16_STAMP(5);
#pragma unroll
            for (int n = 0; n < 32; ++n) {
                const int gi = n & 7;
                const int t = n < 8 ? 0 : n < 16 ? 1 + (gi >> 2) : n < 24 ? 3 + (gi >> 1) : 7 + gi;
                const int tp = n == 0 ? -1 : (n - 1 < 8 ? 0 : n - 1 < 16 ? 1 + (((n - 1) & 7) >> 2) : n - 1 < 24 ? 3 + (((n - 1) & 7) >> 1) : 7 + ((n - 1) & 7));
                if (t != tp) loadt(t + RING - 1);            // first use of pair t: its predecessor's slot is free
                if (n < 8) bfly1_31<UC, 3>(x, gi, g[t % RING], c);
                else if (n < 16) bfly1_31<UC, 2>(x, gi, g[t % RING], c);
                else if (n < 24) bfly1_31<UC, 1>(x, gi, g[t % RING], c);
                else bfly1_31<UC, 0>(x, gi, g[t % RING], c);
            }
        }
        reduce_all(x, c, big && (jb.sched & 8));
#ifdef MKHE_H16_D31
        constexpr bool D31 = true;
#else
        constexpr bool D31 = UC && MKHE_H16_UD31;
#endif
        // ---- phase D: bits 1..0, twiddles psi[2^13 + 4d + i], psi[2^14 + 8d + i], d = (16h + wave) * 64 + lane: pairs 0..3 (k = 13,
        // two butterflies each) and 4..11 (k = 14)
        if constexpr (D31) {
            const int ld = lane_id();
            const unsigned du = (unsigned)((16 * ht + wv) * 64 + ld);
            __builtin_assume(du < 2048);
            u64 g[RING][2];
            gcptr p31v = (gcptr)jb.psi31;
            auto loadt = [&](int t) {
                if (t < 4) ld2(g[t % RING], (gcptr2)sbk(p31v, 2 * 8192 * tm) + t, 4 * du);
                else if (t < 12) ld2(g[t % RING], (gcptr2)sbk(p31v, 2 * 16384 * tm) + (t - 4), 8 * du);
            };
            loadt(0); loadt(1); if (RING > 3) loadt(2);
            H16_STAMP(6);
            exchange<X_CD>(x, lds, wv);
            H16_STAMP(7);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int gi = n & 7;
                const int t = n < 8 ? (gi >> 1) : 4 + gi;
                const int tp = n == 0 ? -1 : (n - 1 < 8 ? (((n - 1) & 7) >> 1) : 4 + ((n - 1) & 7));
                if (t != tp) loadt(t + RING - 1);
                if (n < 8) bfly1_31<UC, 1>(x, gi, g[t % RING], c);
                else bfly1_31<UC, 0>(x, gi, g[t % RING], c);
            }
        } else {
        // (phase D keeps the two-round product on the balanced path: with pairs of constants per twiddle the register allocator spills around it)
        // ---- phase D: bits 1..0, twiddles psi[2^13 + 4d + i], psi[2^14 + 8d + i], d = (16h + wave) * 64 + lane
            const int ld = lane_id();
            const unsigned du = (unsigned)((16 * ht + wv) * 64 + ld);
            __builtin_assume(du < 2048);
            u64 g[6][2];
            auto loadg = [&](int k) {
                if (k < 2) ld2(g[k], (gcptr2)sbk(psi_v, 8192 * tm) + k, 2 * du);
                else ld2(g[k], (gcptr2)sbk(psi_v, 16384 * tm) + (k - 2), 4 * du);
            };
            loadg(0); loadg(1); loadg(2);
            H16_STAMP(6);
            exchange<X_CD>(x, lds, wv);
            H16_STAMP(7);
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                if (n == 4) loadg(3);
                if (n == 8) loadg(4);
                if (n == 10) loadg(5);
                if (n == 8 && !(big || !jb.skip_norm)) {
                    // the positive bias of engine-internal digits (see "output representative" below) enters through the eight U operands of the
                    // LAST stage -- X = (U + b) + T, Y = (U + b) - T -- instead of being added to the sixteen results
                    const i64 bias = UC ? (i64)((c.q << 6) + (c.q << 3) + (c.q << 1) + c.q) : (i64)((c.q << 4) + (c.q << 3));
#pragma unroll
                    for (int r = 0; r < 16; r += 2) x[r] = (u64)((i64)x[r] + bias);
                }
                const int gi = n & 7;
                if (n < 8) bfly1<1>(x, gi, g[gi >> 2][(gi >> 1) & 1], c);
                else bfly1<0>(x, gi, g[2 + (gi >> 1)][gi & 1], c);
            }
        }
        H16_STAMP(8);
        // ---- output representative
        if (big || !jb.skip_norm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const i64 y = pred((i64)x[r], c);                                    // (-q, q)
                x[r] = (u64)(y + ((y >> 63) & (i64)c.q));                           // canonical (lattigo: final BRedAdd)
            }
        } else {
            // engine-internal digits: same residue, positive.  Balanced path: |x| < 20q -> + 24q -> (4q, 44q); U class: x in (-75q, 79q) -> + 75q -> (0, 154q) < 2^62
            if constexpr (D31) {           // (the two-round phase D above has added it already, through the U operands of its last stage)
                const i64 bias = UC ? (i64)((c.q << 6) + (c.q << 3) + (c.q << 1) + c.q) : (i64)((c.q << 4) + (c.q << 3));
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = (u64)((i64)x[r] + bias);
            }
        }
        H16_STAMP(9);
        exchange<X_DE>(x, lds, wv);
        H16_STAMP(10);
        {
            const int obase = h * HH + wv * 1024;
            const unsigned lb = 8u * (unsigned)lane_id();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#ifdef MKHE_H16_X_NOSTORE
                if (x[r] != 0x123456789abcdefull) continue;      // timing experiment only: (almost) no result stores
#endif
#ifndef MKHE_X_NO_NTSTORE      // results are written once and read by later kernels: stream them past the caches (twiddles, sources and the CRS stay resident)
                __builtin_nontemporal_store(x[r], (u64 __attribute__((address_space(1)))*)at(sbk(dst, obase + r * 64), lb));
#else
                *at(sbk(dst, obase + r * 64), lb) = x[r];
#endif
            }
        }
        H16_STAMP(11);
    }
}

// ------------------------------------------------------------------ the F class (round 3): moduli below 2^45.67 in double precision
// FP64 FMA issues at full rate on CDNA4, the 32 x 32 -> 64 multiply-adds of the integer products at a quarter of it, and a modular product of exact
// integers held in doubles is an error-free transformation:
//   h = a w (rounded), l = fma(a, w, -h) (the exact rest), k = rint(h (1/q)) -- three roundings of relative size 2^-53 on a quotient below 2^51:
//   within 0.75 of a w / q before the rint --, r = fma(-k, q, h) (exact), T = r + l = a w - k q exactly, |T| <= 1.25 q (1.02 q observed)
//   --  6 full-rate instructions + add and subtract.
// tools/ubench/bflyf64_rate.hip: 38.9 cycles per wave-butterfly at 2.37 GHz = 16.4 ns against 53.8 cycles at 2.11 GHz = 25.5 ns for mm30u (the clock
// RISES: the kernel runs at the package power cap, and the FP64 pipe draws less than the integer multiplier).  Everything must stay an exact integer
// below 2^53 (and data words below 2^51 for the bound above): inputs in [0, 22.2 q) (decomp_spread4_kernel; < 4 q behind ntt_pass4_fwd_kernel), 14
// stages of growth by at most 1.25 q: x in (-17.5 q, 39.7 q), and the 40 q bias of internal digits: (22.5 q, 79.7 q) -- 80 q < 2^52 is the class,
// the 45-bit primes of PN16QP1761 (33 of its 38 moduli).  Twiddles are PLAIN residues as doubles (NttBatch::psif), values travel
// through the LDS re-distributions as their bit patterns, u64 <-> double by the 2^52 trick (one OR / AND on the high word, one add).
__device__ __forceinline__ double u2d(u64 v) { return __builtin_bit_cast(double, v | 0x4330000000000000ull) - 4503599627370496.0; }      // v < 2^52
__device__ __forceinline__ u64 d2u(double y) { return __builtin_bit_cast(u64, y + 4503599627370496.0) & 0x000fffffffffffffull; }         // 0 <= y < 2^52
struct FC { double q, qinv; };
__device__ __forceinline__ void bflyF(u64& U, u64& V, double w, const FC& c) {
    const double a = __builtin_bit_cast(double, V), u = __builtin_bit_cast(double, U);
    const double h = a * w;
    const double l = __builtin_fma(a, w, -h);
    const double k = __builtin_rint(h * c.qinv);
    const double r = __builtin_fma(-k, c.q, h);
    const double T = r + l;
    U = __builtin_bit_cast(u64, u + T);
    V = __builtin_bit_cast(u64, u - T);
}
template <int B, int G0 = 0, int NG = 8> __device__ __forceinline__ void stageF(u64 (&x)[16], const u64* tw, const FC& c) {
#pragma unroll
    for (int g = G0; g < G0 + NG; ++g) {
        const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
        bflyF(x[i0], x[i0 | (1 << B)], __builtin_bit_cast(double, tw[(g >> B) - (G0 >> B)]), c);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int B> __device__ __forceinline__ void bflyF1(u64 (&x)[16], int g, u64 w, const FC& c) {
    const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
    bflyF(x[i0], x[i0 | (1 << B)], __builtin_bit_cast(double, w), c);
    __builtin_amdgcn_sched_barrier(0);
}
// one quarter (2^14 points, twiddle root jb.root) of an N = 2^16 limb: the forward pass of limb<.., 14> on the products above
__device__ __forceinline__ void limb_f(const Job& jb, u32* lds, const int wv) {
    const int tm = jb.root >> 1, ht = jb.root & 1;
    FC c;
    {
        const u64 q = jb.mp->q;
        c.q = (double)q; c.qinv = 1.0 / (double)q;
        // (computed on the vector unit from a wave-uniform q: back into scalar registers, so that they are SGPR operands of every product)
        const u64 qb = __builtin_bit_cast(u64, c.q), ib = __builtin_bit_cast(u64, c.qinv);
        const u64 qs_ = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)hi32(qb)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)lo32(qb));
        const u64 is_ = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)hi32(ib)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)lo32(ib));
        c.q = __builtin_bit_cast(double, qs_); c.qinv = __builtin_bit_cast(double, is_);
    }
    scptr pf = (scptr)jb.psif;                                // plain twiddles as doubles, bit-reversed order, rows of the whole limb
    gcptr pfv = (gcptr)jb.psif;
    const gcptr src = jb.src; const gptr dst = jb.dst;
    u64 x[16];
    {
        const unsigned tb = 8u * (unsigned)(wv * 64 + lane_id());
        u64 lo8[8], hi8[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { lo8[r] = ld_issue(sbk(src, r * NT), tb); hi8[r] = ld_issue(sbk(src, (8 + r) * NT), tb); }
        ld_wait16(lo8, hi8);
#pragma unroll
        for (int r = 0; r < 8; ++r) { x[r] = __builtin_bit_cast(u64, u2d(lo8[r])); x[8 + r] = __builtin_bit_cast(u64, u2d(hi8[r])); __builtin_amdgcn_sched_barrier(0); }
    }
    if (MKHE_H16_PRIO) __builtin_amdgcn_s_setprio(MKHE_H16_PRIO);
    {
        u64 tw[8];
        tw[0] = pf[2 * tm + ht];
        stageF<3>(x, tw, c);
#pragma unroll
        for (int i = 0; i < 2; ++i) tw[i] = pf[4 * tm + 2 * ht + i];
        stageF<2>(x, tw, c);
#pragma unroll
        for (int i = 0; i < 4; ++i) tw[i] = pf[8 * tm + 4 * ht + i];
        stageF<1>(x, tw, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) tw[i] = pf[16 * tm + 8 * ht + i];
        stageF<0>(x, tw, c);
    }
    exchange<X_AB>(x, lds, wv);
    if (MKHE_H16_PRIO) __builtin_amdgcn_s_setprio(0);
    {
        const int cb = 16 * ht + wv;
        u64 tw[8];
        tw[0] = pf[32 * tm + cb];
        stageF<3>(x, tw, c);
#pragma unroll
        for (int i = 0; i < 2; ++i) tw[i] = pf[64 * tm + 2 * cb + i];
        stageF<2>(x, tw, c);
#pragma unroll
        for (int i = 0; i < 4; ++i) tw[i] = pf[128 * tm + 4 * cb + i];
        stageF<1>(x, tw, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) tw[i] = pf[256 * tm + 8 * cb + i];
        stageF<0>(x, tw, c);
    }
    // phase C: the 15 per-lane twiddles as eight 16-byte groups, in the order {t0}, {1,2}, {3,4}, {5,6}, {7,8} .. {13,14}; three groups in flight
    {
        const int lc = lane_id();
        const unsigned cu = (unsigned)((16 * ht + wv) * 16 + (lc >> 2));
        __builtin_assume(cu < 512);
        u64 g[3][2];
        auto loadt = [&](int G, int slot) {
            if (G == 0) { g[slot][0] = *at(sbk(pfv, 512 * tm), 8u * cu); g[slot][1] = 0; }
            else if (G == 1) ld2(g[slot], (gcptr2)sbk(pfv, 1024 * tm), cu);
            else if (G < 4) ld2(g[slot], (gcptr2)sbk(pfv, 2048 * tm) + (G - 2), 2 * cu);
            else ld2(g[slot], (gcptr2)sbk(pfv, 4096 * tm) + (G - 4), 4 * cu);
        };
        loadt(0, 0); loadt(1, 1);
        exchange<X_BC>(x, lds, wv);
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            const int gi = n & 7;
            const int G = n < 8 ? 0 : n < 16 ? 1 : n < 24 ? 2 + (gi >> 2) : 4 + (gi >> 1);
            const int Gp = n == 0 ? -1 : (n - 1 < 8 ? 0 : n - 1 < 16 ? 1 : n - 1 < 24 ? 2 + (((n - 1) & 7) >> 2) : 4 + (((n - 1) & 7) >> 1));
            if (G != Gp && G + 2 < 8) loadt(G + 2, (G + 2) % 3);
            const u64 w = n < 8 ? g[G % 3][0] : n < 16 ? g[G % 3][gi >> 2] : n < 24 ? g[G % 3][(gi >> 1) & 1] : g[G % 3][gi & 1];
            if (n < 8) bflyF1<3>(x, gi, w, c);
            else if (n < 16) bflyF1<2>(x, gi, w, c);
            else if (n < 24) bflyF1<1>(x, gi, w, c);
            else bflyF1<0>(x, gi, w, c);
        }
    }
    // phase D: psi[2^13 root' + 4 d + i], psi[2^14 root' + 8 d + i] as in the integer pass
    {
        const int ld = lane_id();
        const unsigned du = (unsigned)((16 * ht + wv) * 64 + ld);
        __builtin_assume(du < 2048);
        // (a ring of three 16-byte groups, as in phase C: with all six groups named the allocator spilled ten registers around this phase)
        u64 g[3][2];
        auto loadg = [&](int G, int slot) {
            if (G < 2) ld2(g[slot], (gcptr2)sbk(pfv, 8192 * tm) + G, 2 * du);
            else ld2(g[slot], (gcptr2)sbk(pfv, 16384 * tm) + (G - 2), 4 * du);
        };
        loadg(0, 0);
        exchange<X_CD>(x, lds, wv);
        loadg(1, 1);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int gi = n & 7;
            const int G = n < 8 ? (gi >> 2) : 2 + (gi >> 1);
            const int Gp = n == 0 ? -1 : (n - 1 < 8 ? (((n - 1) & 7) >> 2) : 2 + (((n - 1) & 7) >> 1));
            if (G != Gp && G + 2 < 6) loadg(G + 2, (G + 2) % 3);
            if (n < 8) bflyF1<1>(x, gi, g[G % 3][(gi >> 1) & 1], c);
            else bflyF1<0>(x, gi, g[G % 3][gi & 1], c);
        }
    }
    // output representative: always canonical (a valid positive representative for the engine-internal digits too: in double precision the
    // reduction is four instructions more than adding a bias, and one code path instead of two keeps the pass within its 64 registers)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const double v = __builtin_bit_cast(double, x[r]);
        double y = __builtin_fma(-__builtin_rint(v * c.qinv), c.q, v);          // |y| <= (q + 1) / 2
        y = y < 0.0 ? y + c.q : y;
        x[r] = d2u(y);
        __builtin_amdgcn_sched_barrier(0);                                        // (one value at a time: interleaved, their temporaries spill)
    }
    exchange<X_DE>(x, lds, wv);
    {
        const int obase = wv * 1024;
        const unsigned lb = 8u * (unsigned)lane_id();
#pragma unroll
        for (int r = 0; r < 16; ++r)
            __builtin_nontemporal_store(x[r], (u64 __attribute__((address_space(1)))*)at(sbk(dst, obase + r * 64), lb));
    }
}

template <bool DEC, bool SPLIT, int LOGN = 15>
__device__ __forceinline__ void fwd_body(const NttBatch& b, u32* lds) {
    constexpr int NL = 1 << LOGN;                 // words per limb = twiddle words per modulus
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    // half_jobs (out-of-place N = 2^15 launches, round 3): the two passes of a limb do not depend on each other out of place (each recomputes
    // stage 0 from the source), so they are two jobs of the walk -- twice as many, half as long: a launch of 896 limbs deals 7 half-limbs to
    // every CU instead of 4 limbs to one half of them and 3 to the other
    const bool halves = !SPLIT && LOGN == 15 && b.half_jobs != 0;
    const int njobs = (b.nslots * b.nouter) << (SPLIT ? (LOGN == 14 ? 2 : 1) : (halves ? 1 : 0));
    // optional start delay of the second half of the persistent grid (the co-resident workgroup of every CU, as far as the
    // dispatcher deals workgroups b and b + gridDim/2 to the same CU): the two workgroups of a CU then sit in different phases
    if (b.lazy_out > 0 && blockIdx.x >= (gridDim.x >> 1)) { for (int i = 0; i < b.lazy_out; ++i) __builtin_amdgcn_s_sleep(127); }
#pragma unroll 1
    for (int job2 = blockIdx.x; job2 < njobs; job2 += gridDim.x) {
        // Which limb position job2 of the walk gets (round 3).  Workgroups b and b + C share a CU (C = gridDim / 2), so CU c owns the positions
        // = c mod C of the job list, and when the list does not fill the last row (njobs mod C = r != 0) the CUs r .. C-1 own one position fewer.
        // The first `lpt_long` jobs of the list are the long ones (59/60-bit moduli: +35 % instructions); they go to those lighter CUs, row by
        // row -- 896 limbs with 168 long ones on 256 CUs: no CU carries more than 4.0 limb-units where the list order gives half of them 4.35
        // (141.8 -> 138.3 us for that launch).  A bijection of [0, njobs): every limb is still transformed exactly once.
        // (every quotient of the mapping is a launch constant computed by launch_ntt16_fwd -- NttBatch::lpt -- or a multiply-high by a
        // precomputed reciprocal on the scalar unit: an integer division here runs on the VALU, in all sixteen waves, for every limb)
        int job = job2, half_pass = 0;
        if constexpr (!SPLIT) {
            kargptr kl = (kargptr)__builtin_amdgcn_kernarg_segment_ptr();
            const int B = kl->lpt.B;
            if (B > 0) {
                const int C = kl->lpt.C, r = kl->lpt.r, w = C - r, full = kl->lpt.full, rem = kl->lpt.rem;
                const int row = (int)udiv_magic((unsigned)job2, kl->lpt.magic_C), col = job2 - row * C;
                const int srow = row < full ? w : (row == full ? rem : 0);
                const int before = row <= full ? row * w : B;         // special positions in the rows above
                if (col >= r && col - r < srow) job = before + (col - r);
                else job = B + job2 - before - (col > r ? (col - r < srow ? col - r : srow) : 0);
            }
            if (halves) {
                // (the mapping above ran on half-limb positions.)  XCD = workgroup index mod 8 and the job list is slot-major with the source limbs
                // innermost, so whole-limb jobs give every XCD one eighth of the source limbs to keep in its L2; half-limb positions keep that
                // property when blocks of 16 of them hold 8 limbs x 2 passes (limb mod 8 = position mod 8) -- plain (limb, pass) pairs put a limb's
                // passes on two XCDs and double every L2's source set (the 1792-limb launch: 256 -> 270 us)
                if (kl->half_jobs == 2) { half_pass = (job >> 3) & 1; job = ((job >> 4) << 3) | (job & 7); }
                else { half_pass = job & 1; job >>= 1; }
            }
        } else job = job2 >> (LOGN == 14 ? 2 : 1);           // SPLIT: the two halves (four quarters) of a limb are consecutive jobs
        // The launch description is re-read from the kernel-argument segment for every limb (a handful of scalar loads) instead
        // of being kept in SGPRs across the limb: kept live it overflows the SGPR file into VGPR lanes, and those VGPRs are
        // what the 64-register budget of this kernel does not have.
        kargptr kb = (kargptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kb));
        // slot-major job order (ntt_kernels.hip job_pointers): consecutive workgroups share a modulus
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
        constexpr int NLIMB = SPLIT ? (1 << 16) : NL;         // twiddle words per modulus: a split launch uses the rows of the whole 2^16-point limb
        jb.psi = kb->psi + (long)m * NLIMB;
        jb.psi31 = kb->psi31 + 2 * (long)m * NLIMB;
        jb.psi31n = kb->psi31n + 8 * (long)m;
        // the per-modulus schedule assumes inputs below 2^60; lazy inputs (BFV digits and ring-R polynomials: src_lazy) and the halves of a split
        // N = 2^16 limb (values in [0, 4q) behind the streaming cross-half stage) keep the round-2 schedule: reduce at the load and after every phase
        // (the byte through a scalar dword load: a byte load of a kernel argument is a VECTOR memory instruction, and the s_waitcnt vmcnt(0) the
        // compiler puts behind it waits for every store of the previous job before this one has requested a word -- round 3 shipped that; 0.4 %)
        jb.sched = (kb->src_lazy || (SPLIT && LOGN == 15)) ? 15 : (int)((((const __attribute__((address_space(4))) unsigned*)kb->sched)[m >> 2] >> (8 * (m & 3))) & 0xffu);
#ifdef MKHE_H16_X_SCHEDBYTE     // MKHE_ABLATION: round 3's form of the line above (same value; the vector byte load and its vmcnt(0)), for the A/B in one call
        jb.sched = (kb->src_lazy || (SPLIT && LOGN == 15)) ? 15 : kb->sched[m];
#endif
        jb.root = 1;
        if constexpr (SPLIT && LOGN == 15) { const int half = job2 & 1; jb.src += half * NN; jb.dst += half * NN; jb.root = 2 + half; }
        if constexpr (SPLIT && LOGN == 14) { const int quarter = job2 & 3; jb.src += quarter * NL; jb.dst += quarter * NL; jb.root = 4 + quarter; }
        jb.mp = (smodptr)kb->mods + m;
        jb.skip_norm = kb->skip_norm != 0;
        jb.trace = kb->trace ? kb->trace + (long)job2 * 16 * 32 : nullptr;
        jb.red = false;
        // quarters: the radix-4 spread hands over values below 22.2 q (poly_kernels.hip decomp_spread4_kernel), which the U class takes as they are
        // (Context::decompose_batch checks its range budget) and the balanced path reduces at the load
        if constexpr (SPLIT && LOGN == 14) jb.red = ((kb->u_mods >> m) & 1) == 0;
        if constexpr (DEC) {
            int sm = m;
            const int rs = kb->reduce_src_mod_is_outer;
            if (rs == 1) sm = outer; else if (rs == 2) sm = kb->outer_mod[outer];
            const u64 qsb = ((smodptr)kb->mods)[sm].q << (kb->src_lazy ? 2 : 0);     // bound of the digit values (< 2^63)
            // U class: raw canonical digits of ANY modulus (< 2^60) fit its range budget -- input + 75q of growth + the 75q bias stay below 2^62 for
            // the 2^54 - delta primes -- so that only lazy (BFV) digits are reduced at the load; the balanced path reduces what exceeds 4q
            if (MKHE_H16_URED && ((kb->u_mods >> m) & 1)) jb.red = qsb >= (1ull << 62) - 150 * jb.mp->q;
            else jb.red = qsb > 4 * jb.mp->q;
        }
#ifdef MKHE_PHASE_TRACE
        if (jb.trace && ((int)threadIdx.x & 63) == 0) {
            u64* tw = jb.trace + (long)wv * 32;
            tw[12] = __builtin_amdgcn_s_memrealtime();
            tw[13] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
            tw[14] = blockIdx.x;
        }
#endif
        const int hf = halves ? half_pass : 0, hl = halves ? half_pass : 1;
        if constexpr (SPLIT && LOGN == 14) {
            // the quarters of an N = 2^16 limb under a modulus of the F class: double-precision butterflies (limb_f)
            if (kb->psif && ((kb->f_mods >> m) & 1)) { jb.psif = kb->psif + (long)m * (1 << 16); limb_f(jb, lds, wv); continue; }
        }
        if ((kb->u_mods >> m) & 1) limb<DEC, SPLIT, true, LOGN>(jb, false, lds, wv, hf, hl);
        else limb<DEC, SPLIT, false, LOGN>(jb, ((kb->small_slots >> s) & 1) == 0, lds, wv, hf, hl);
#ifdef MKHE_PHASE_TRACE
        if (jb.trace && ((int)threadIdx.x & 63) == 0) jb.trace[(long)wv * 32 + 28] = __builtin_amdgcn_s_memrealtime();
#endif
    }
}

template <bool DEC>
__global__ void __launch_bounds__(NT, 8) ntt16_fwd_kernel(NttBatch b) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    fwd_body<DEC, false>(b, lds);
}
// N = 2^14: one pass per limb (the reference's first benchmark set PN14QP439, mkckks/mkckks_benchmark_test.go:13, and the cnn ring)
template <bool DEC>
__global__ void __launch_bounds__(NT, 8) ntt14_fwd_kernel(NttBatch b) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    fwd_body<DEC, false, 14>(b, lds);
}
// the two 2^15-point sub-transforms of every 2^16-point limb, in place, after the cross-half stage (NttBatch::split)
__global__ void __launch_bounds__(NT, 8) ntt16_fwd_split_kernel(NttBatch b) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    fwd_body<false, true>(b, lds);
}

// the four 2^14-point sub-transforms of every 2^16-point limb, one pass each (in place or not), after the two cross stages (NttBatch::split = 2)
__global__ void __launch_bounds__(NT, 8) ntt14_fwd_split_kernel(NttBatch b) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    fwd_body<false, true, 14>(b, lds);
}

// ------------------------------------------------------------------ inverse (round 3)
// The mirror image of one forward pass on 2^14 points: Gentleman-Sande butterflies X = U + V, Y = (U - V) w on the same one-round products, the
// phases in the order D, C, B, A with the four re-distributions run backwards (written where the forward pass reads, read where it writes: the same
// conflict-free address sets), inverse twiddles at the forward indices (NttBatch::psi31 / psi = the INVERSE tables for these launches).  A job is
//   split = 0: a whole N = 2^14 limb (root 1);   split = 1: one half of an N = 2^15 limb (root 2 + half), the cross-half stage follows as the
//   streaming pass ntt_split_inv_kernel;   split = 2: one quarter of an N = 2^16 limb (root 4 + quarter), the two cross stages follow as
//   ntt_pass4_inv_kernel -- one pass per job either way, in place or not.  N^-1 of the WHOLE limb is folded into the last stage (both outputs are
// products there: NttBatch::inv31c holds the pairs of N^-1 and psiinv[root] N^-1), results are canonical.
// Ranges: a sum doubles per stage where a forward stage adds, so the never-reduced values are brought back by the cheap partial reduction after
// phase D and after phase B (U class: 0.51 q x 2^8 = 131 q < 2^62 / q = 160), the moduli in between also after phase C, the 59/60-bit primes at
// the load and after every second stage (2^62 = 4 q there).  Merged launches (NttBatch::vi) add up the group's members at the load.
// (one partial reduction at a time: interleaved, the temporaries of sixteen of them do not fit beside a ring of twiddle pairs in flight)
__device__ __forceinline__ void reduce_seq(u64 (&x)[16], const MC& c, bool on) {
    if (on) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { x[r] = (u64)pred((i64)x[r], c); __builtin_amdgcn_sched_barrier(0); }
    }
}
template <bool SW, bool UC> __device__ __forceinline__ void gs31(u64& U, u64& V, const u64* tw, const MC& c) {
    const i64 u = (i64)U, v = (i64)V;
    U = (u64)(u + v);
    V = (u64)(UC ? mm30u<SW>(u - v, tw[0], tw[1], c) : mm31<SW>(u - v, tw[0], tw[1], c));
}
template <bool UC, int B, int G0 = 0, int NG = 8> __device__ __forceinline__ void stage_gs31(u64 (&x)[16], const u64* tw, const MC& c) {
#pragma unroll
    for (int g = G0; g < G0 + NG; ++g) {
        const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
        gs31<true, UC>(x[i0], x[i0 | (1 << B)], tw + 2 * ((g >> B) - (G0 >> B)), c);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <bool UC, int B> __device__ __forceinline__ void gs1_31(u64 (&x)[16], int g, const u64* tw, const MC& c) {
    const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
    gs31<false, UC>(x[i0], x[i0 | (1 << B)], tw, c);
    __builtin_amdgcn_sched_barrier(0);
}
template <int B> __device__ __forceinline__ void gs1(u64 (&x)[16], int g, u64 w, const MC& c) {
    const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
    const i64 u = (i64)x[i0], v = (i64)x[i0 | (1 << B)];
    x[i0] = (u64)(u + v);
    x[i0 | (1 << B)] = (u64)mm<false>(u - v, w, c);
    __builtin_amdgcn_sched_barrier(0);
}
template <int X> __device__ __forceinline__ void exchange_inv(u64 (&x)[16], u32* lds, int wv) {
    constexpr bool CROSS = X == X_AB;
    const int l = lane_id();
    typedef __attribute__((address_space(3))) u32* lptr;
    typedef volatile __attribute__((address_space(3))) u32* vlptr;
    lptr wr = (lptr)lds + rbase<X>(wv, l);
    vlptr rd = (vlptr)((lptr)lds + wbase<X>(wv, l));
#pragma unroll
    for (int r = 0; r < 16; ++r) wr[roff<X>(r)] = lo32(x[r]);
    xsync<CROSS>();
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = (x[r] & 0xffffffff00000000ull) | rd[woff<X>(r)];
    xsync<CROSS>();
#pragma unroll
    for (int r = 0; r < 16; ++r) wr[roff<X>(r)] = hi32(x[r]);
    xsync<CROSS>();
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = ((u64)rd[woff<X>(r)] << 32) | lo32(x[r]);
    xsync<CROSS>();                  // (cross-wave: the regions are free again before anybody's next write)
}
struct JobInv { gcptr src; gptr dst; const u64* psi; const u64* psi31; const u64* fin; smodptr mp; int root; int nsum; int g, pm; };
// NttBatch::vi_parts[item] through a scalar dword load (see ntt_kernels.hip)
__device__ __forceinline__ unsigned vi_parts_of(kargptr kb, int item) {
    return (((const __attribute__((address_space(4))) unsigned*)kb->vi_parts)[item >> 1] >> (16 * (item & 1))) & 0xffffu;
}
// word offset of summand k >= 1 of a merged job relative to jb.src (ntt_kernels.hip vi_summand_offset: members, their further parts, the Q-only extra)
__device__ __forceinline__ long inv_summand_offset(const JobInv& jb, int k) {
    kargptr kb = (kargptr)__builtin_amdgcn_kernarg_segment_ptr();
    const unsigned mem = kb->vi_mem[jb.g];
    if (jb.pm >= 0) {
        const int item = (int)((mem >> (8 * jb.pm)) & 255u);
        return ((long)(vi_parts_of(kb, item) >> 8) + (k - 1) - (long)item) * kb->src_outer;
    }
    const int cnt = kb->vi_cnt[jb.g], item0 = (int)(mem & 255u);
    int kk = k;
    for (int j = 0; j < cnt; ++j) {
        const int item = (int)((mem >> (8 * j)) & 255u);
        const unsigned parts = vi_parts_of(kb, item);
        const int np_ = 1 + (int)(parts & 255u);
        if (kk < np_) return ((long)(kk == 0 ? item : (int)(parts >> 8) + kk - 1) - (long)item0) * kb->src_outer;
        kk -= np_;
    }
    // (the Q-only extra is a plain polynomial: its limb m sits where the first member's does inside its item)
    return (long)(((u64)kb->vi_extra[jb.g] - (u64)kb->src) >> 3) - (long)item0 * kb->src_outer;
}
template <bool UC>
__device__ __forceinline__ void limb_inv(const JobInv& jb, const bool big_, u32* lds, const int wv) {
    // (the balanced instantiation serves the 59/60-bit primes and the few moduli between the classes alike, on the schedule of the former: no
    // wave-uniform branches around the reductions -- with them the register allocator spilled 140 VGPRs)
    constexpr bool big = !UC;
    (void)big_;
    const int tm = jb.root >> 1, ht = jb.root & 1;
    smodptr mp = jb.mp;
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
    scptr p31 = (scptr)jb.psi31;
    gcptr psi_v = (gcptr)jb.psi;
    const gcptr src = jb.src; const gptr dst = jb.dst;
    u64 x[16];
    {
        // the forward pass's output layout: register r = words wave * 1024 + r * 64 + lane of the 2^14 block
        const unsigned lb = 8u * (unsigned)(wv * 1024 + lane_id());
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = *at(sbk(src, r * 64), lb);
#pragma unroll 1
        for (int k = 1; k < jb.nsum; ++k) {
            {
                const gcptr sk = src + inv_summand_offset(jb, k);
                // (eight summands in flight at a time: sixteen more loads beside the sixteen values do not fit the 64-VGPR budget)
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    u64 t8[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) t8[r] = *at(sbk(sk, (r0 + r) * 64), lb);
#pragma unroll
                    for (int r = 0; r < 8; ++r) x[r0 + r] += t8[r];                     // canonical summands: < 13q together (VI_SUMS)
                    asm volatile("" ::: "memory");
                }
            }
        }
    }
    reduce_seq(x, c, big);
    exchange_inv<X_DE>(x, lds, wv);
    // ---- phase D backwards: bits 0, 1 (two-round product on the 8-byte inverse twiddles, as in the forward pass)
    {
        const int ld = lane_id();
        const unsigned du = (unsigned)((16 * ht + wv) * 64 + ld);
        __builtin_assume(du < 2048);
        u64 g[6][2];
        auto loadg = [&](int k) {
            if (k < 2) ld2(g[k], (gcptr2)sbk(psi_v, 8192 * tm) + k, 2 * du);
            else ld2(g[k], (gcptr2)sbk(psi_v, 16384 * tm) + (k - 2), 4 * du);
        };
        loadg(2); loadg(3); loadg(4);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            if (n == 2) loadg(5);
            if (n == 4) loadg(0);
            if (n == 6) loadg(1);
            const int gi = n & 7;
            if (n < 8) gs1<0>(x, gi, g[2 + (gi >> 1)][gi & 1], c);
            else gs1<1>(x, gi, g[gi >> 2][(gi >> 1) & 1], c);
        }
    }
    reduce_seq(x, c, true);
    // ---- phase C backwards: bits 2..5, per-lane pairs; pair t as in the forward pass (0: k = 9; 1, 2: k = 10; 3..6: k = 11; 7..14: k = 12), used
    // in the order 7..14, 3..6, 1, 2, 0
    {
        const int lc = lane_id();
        const unsigned cu = (unsigned)((16 * ht + wv) * 16 + (lc >> 2));
        __builtin_assume(cu < 512);
        u64 g[RING][2];
        gcptr p31v = (gcptr)jb.psi31;
        auto loadt = [&](int t, int slot) {
            if (t == 0) ld2(g[slot], (gcptr2)sbk(p31v, 2 * 512 * tm), cu);
            else if (t < 3) ld2(g[slot], (gcptr2)sbk(p31v, 2 * 1024 * tm) + (t - 1), 2 * cu);
            else if (t < 7) ld2(g[slot], (gcptr2)sbk(p31v, 2 * 2048 * tm) + (t - 3), 4 * cu);
            else ld2(g[slot], (gcptr2)sbk(p31v, 2 * 4096 * tm) + (t - 7), 8 * cu);
        };
        constexpr int seq[15] = {7, 8, 9, 10, 11, 12, 13, 14, 3, 4, 5, 6, 1, 2, 0};
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) loadt(seq[i], i);
        exchange_inv<X_CD>(x, lds, wv);
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            const int gi = n & 7;
            const int i = n < 8 ? gi : n < 16 ? 8 + (gi >> 1) : n < 24 ? 12 + (gi >> 2) : 14;          // position in seq of this butterfly's pair
            const int ip = n == 0 ? -1 : (n - 1 < 8 ? ((n - 1) & 7) : n - 1 < 16 ? 8 + (((n - 1) & 7) >> 1) : n - 1 < 24 ? 12 + (((n - 1) & 7) >> 2) : 14);
            if (i != ip && i + RING - 1 < 15) loadt(seq[i + RING - 1], (i + RING - 1) % RING);
            if (n < 8) gs1_31<UC, 0>(x, gi, g[i % RING], c);
            else if (n < 16) gs1_31<UC, 1>(x, gi, g[i % RING], c);
            else if (n < 24) gs1_31<UC, 2>(x, gi, g[i % RING], c);
            else gs1_31<UC, 3>(x, gi, g[i % RING], c);
            if (n == 15) reduce_seq(x, c, big);
        }
    }
    reduce_seq(x, c, !UC);
    exchange_inv<X_BC>(x, lds, wv);
    // ---- phase B backwards: bits 6..9, scalar pairs
    {
        const int cb = 16 * ht + wv;
        u64 tw[6], tm4[8], ta[8], tb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ta[i] = p31[2 * (256 * tm + 8 * cb) + i];
#pragma unroll
        for (int i = 0; i < 8; ++i) tb[i] = p31[2 * (256 * tm + 8 * cb + 4) + i];
        stage_gs31<UC, 0, 0, 4>(x, ta, c);
        stage_gs31<UC, 0, 4, 4>(x, tb, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) tm4[i] = p31[2 * (128 * tm + 4 * cb) + i];
        stage_gs31<UC, 1>(x, tm4, c);
        reduce_seq(x, c, big);
#pragma unroll
        for (int i = 0; i < 4; ++i) tw[2 + i] = p31[2 * (64 * tm + 2 * cb) + i];
        stage_gs31<UC, 2>(x, tw + 2, c);
#pragma unroll
        for (int i = 0; i < 2; ++i) tw[i] = p31[2 * (32 * tm + cb) + i];
        stage_gs31<UC, 3>(x, tw, c);
    }
    reduce_seq(x, c, true);
    exchange_inv<X_AB>(x, lds, wv);
    // ---- phase A backwards: bits 10..13; the last stage multiplies both outputs (N^-1 folded in)
    {
        u64 tw[6], tm4[8], ta[8], tb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ta[i] = p31[2 * (16 * tm + 8 * ht) + i];
#pragma unroll
        for (int i = 0; i < 8; ++i) tb[i] = p31[2 * (16 * tm + 8 * ht + 4) + i];
        stage_gs31<UC, 0, 0, 4>(x, ta, c);
        stage_gs31<UC, 0, 4, 4>(x, tb, c);
#pragma unroll
        for (int i = 0; i < 8; ++i) tm4[i] = p31[2 * (8 * tm + 4 * ht) + i];
        stage_gs31<UC, 1>(x, tm4, c);
        reduce_seq(x, c, big);
#pragma unroll
        for (int i = 0; i < 4; ++i) tw[2 + i] = p31[2 * (4 * tm + 2 * ht) + i];
        stage_gs31<UC, 2>(x, tw + 2, c);
        scptr fin = (scptr)jb.fin;
        const u64 f[4] = {fin[0], fin[1], fin[2], fin[3]};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const i64 u = (i64)x[g], v = (i64)x[g + 8];
            const i64 s = UC ? mm30u<true>(u + v, f[0], f[1], c) : mm31<true>(u + v, f[0], f[1], c);
            const i64 d = UC ? mm30u<true>(u - v, f[2], f[3], c) : mm31<true>(u - v, f[2], f[3], c);
            x[g] = (u64)s; x[g + 8] = (u64)d;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const i64 y = pred((i64)x[r], c);
        x[r] = (u64)(y + ((y >> 63) & (i64)c.q));                           // canonical
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        const unsigned tb = 8u * (unsigned)(wv * 64 + lane_id());
#pragma unroll
        for (int r = 0; r < 16; ++r) *at(sbk(dst, r * NT), tb) = x[r];
    }
}
// The same pass for the moduli outside the U class (the 59/60-bit primes and the few between the classes): Harvey-style Gentleman-Sande
// butterflies on the two-round product with the 8-byte inverse twiddles, values in [0, 2q) throughout -- X = (U + V) - [2q], Y = (U + 2q - V) w --
// so that no partial reduction is needed anywhere (on the one-round product with reductions after every second stage this instantiation spilled
// 738 VGPRs).  15 + 6 instructions per butterfly against 9 + 3: the minority of the limbs.
template <bool SW> __device__ __forceinline__ void gsH(u64& U, u64& V, u64 w, const MC& c, u64 q2) {
    const u64 s = csub(U + V, q2), d = U + q2 - V;
    U = s;
    V = (u64)(mm<SW>((i64)d, w, c) + (i64)c.q);
}
template <int B, int G0 = 0, int NG = 8> __device__ __forceinline__ void stage_gsH(u64 (&x)[16], const u64* tw, const MC& c, u64 q2) {
#pragma unroll
    for (int g = G0; g < G0 + NG; ++g) {
        const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
        gsH<true>(x[i0], x[i0 | (1 << B)], tw[(g >> B) - (G0 >> B)], c, q2);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int B> __device__ __forceinline__ void gsH1(u64 (&x)[16], int g, u64 w, const MC& c, u64 q2) {
    const int i0 = ((g >> B) << (B + 1)) | (g & ((1 << B) - 1));
    gsH<false>(x[i0], x[i0 | (1 << B)], w, c, q2);
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void limb_inv_cs(const JobInv& jb, u32* lds, const int wv) {
    const int tm = jb.root >> 1, ht = jb.root & 1;
    smodptr mp = jb.mp;
    const u64 qs = mp->qs;
    MC c;
    c.q = mp->q; c.ninv = mp->ninv32;
    c.q0 = (i32)lo32(qs); c.q1 = (i32)hi32(qs);
    c.finv = 0.0f; c.p0 = 0; c.p1 = 0;
    asm("" : "+s"(c.q0), "+s"(c.q1), "+s"(c.ninv));
    const u64 q = c.q, q2 = 2 * c.q;
    scptr psi_s = (scptr)jb.psi;
    gcptr psi_v = (gcptr)jb.psi;
    const gcptr src = jb.src; const gptr dst = jb.dst;
    u64 x[16];
    {
        const unsigned lb = 8u * (unsigned)(wv * 1024 + lane_id());
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = *at(sbk(src, r * 64), lb);
#pragma unroll 1
        for (int k = 1; k < jb.nsum; ++k) {
            {
                const gcptr sk = src + inv_summand_offset(jb, k);
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    u64 t8[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) t8[r] = *at(sbk(sk, (r0 + r) * 64), lb);
#pragma unroll
                    for (int r = 0; r < 8; ++r) x[r0 + r] = csub(x[r0 + r] + t8[r], q);      // canonical summands, canonical sum
                    asm volatile("" ::: "memory");
                }
            }
        }
    }
    exchange_inv<X_DE>(x, lds, wv);
    {
        const int ld = lane_id();
        const unsigned du = (unsigned)((16 * ht + wv) * 64 + ld);
        __builtin_assume(du < 2048);
        u64 g[6][2];
        auto loadg = [&](int k) {
            if (k < 2) ld2(g[k], (gcptr2)sbk(psi_v, 8192 * tm) + k, 2 * du);
            else ld2(g[k], (gcptr2)sbk(psi_v, 16384 * tm) + (k - 2), 4 * du);
        };
        loadg(2); loadg(3); loadg(4);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            if (n == 2) loadg(5);
            if (n == 4) loadg(0);
            if (n == 6) loadg(1);
            const int gi = n & 7;
            if (n < 8) gsH1<0>(x, gi, g[2 + (gi >> 1)][gi & 1], c, q2);
            else gsH1<1>(x, gi, g[gi >> 2][(gi >> 1) & 1], c, q2);
        }
    }
    // phase C: the 15 twiddles psi[512 tm + cu] | psi[1024 tm + 2 cu + {0,1}] | psi[2048 tm + 4 cu + {0..3}] | psi[4096 tm + 8 cu + {0..7}] as eight
    // 16-byte groups, used in the order G0..G3 (bit 2: 8 twiddles), G4, G5 (bit 3: 4), G6 (bit 4: 2), G7 (bit 5: 1); three groups in flight
    {
        const int lc = lane_id();
        const unsigned cu = (unsigned)((16 * ht + wv) * 16 + (lc >> 2));
        __builtin_assume(cu < 512);
        u64 g[3][2];
        auto loadt = [&](int G, int slot) {
            if (G < 4) ld2(g[slot], (gcptr2)sbk(psi_v, 4096 * tm) + G, 4 * cu);
            else if (G < 6) ld2(g[slot], (gcptr2)sbk(psi_v, 2048 * tm) + (G - 4), 2 * cu);
            else if (G == 6) ld2(g[slot], (gcptr2)sbk(psi_v, 1024 * tm), cu);
            else { g[slot][0] = *at(sbk(psi_v, 512 * tm), 8u * cu); g[slot][1] = 0; }
        };
        loadt(0, 0); loadt(1, 1);
        exchange_inv<X_CD>(x, lds, wv);
#pragma unroll
        for (int n = 0; n < 32; ++n) {
            const int gi = n & 7;
            const int G = n < 8 ? (gi >> 1) : n < 16 ? 4 + (gi >> 2) : n < 24 ? 6 : 7;
            const int Gp = n == 0 ? -1 : (n - 1 < 8 ? (((n - 1) & 7) >> 1) : n - 1 < 16 ? 4 + (((n - 1) & 7) >> 2) : n - 1 < 24 ? 6 : 7);
            if (G != Gp && G + 2 < 8) loadt(G + 2, (G + 2) % 3);
            const u64 w = n < 8 ? g[G % 3][gi & 1] : n < 16 ? g[G % 3][(gi >> 1) & 1] : n < 24 ? g[G % 3][gi >> 2] : g[G % 3][0];
            if (n < 8) gsH1<0>(x, gi, w, c, q2);
            else if (n < 16) gsH1<1>(x, gi, w, c, q2);
            else if (n < 24) gsH1<2>(x, gi, w, c, q2);
            else gsH1<3>(x, gi, w, c, q2);
        }
    }
    exchange_inv<X_BC>(x, lds, wv);
    {
        const int cb = 16 * ht + wv;
        u64 tw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) tw[i] = psi_s[256 * tm + 8 * cb + i];
        stage_gsH<0>(x, tw, c, q2);
#pragma unroll
        for (int i = 0; i < 4; ++i) tw[i] = psi_s[128 * tm + 4 * cb + i];
        stage_gsH<1>(x, tw, c, q2);
#pragma unroll
        for (int i = 0; i < 2; ++i) tw[i] = psi_s[64 * tm + 2 * cb + i];
        stage_gsH<2>(x, tw, c, q2);
        tw[0] = psi_s[32 * tm + cb];
        stage_gsH<3>(x, tw, c, q2);
    }
    exchange_inv<X_AB>(x, lds, wv);
    {
        u64 tw[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) tw[i] = psi_s[16 * tm + 8 * ht + i];
        stage_gsH<0>(x, tw, c, q2);
#pragma unroll
        for (int i = 0; i < 4; ++i) tw[i] = psi_s[8 * tm + 4 * ht + i];
        stage_gsH<1>(x, tw, c, q2);
#pragma unroll
        for (int i = 0; i < 2; ++i) tw[i] = psi_s[4 * tm + 2 * ht + i];
        stage_gsH<2>(x, tw, c, q2);
        scptr fin = (scptr)jb.fin;
        const u64 f0 = fin[4], f1 = fin[5];                   // N^-1 R, psiinv[root] N^-1 R (signed-split form)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const u64 U = x[g], V = x[g + 8];
            const u64 sv = (u64)(mm<true>((i64)(U + V), f0, c) + (i64)q);
            const u64 dv = (u64)(mm<true>((i64)(U + q2 - V), f1, c) + (i64)q);
            x[g] = csub(sv, q); x[g + 8] = csub(dv, q);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    {
        const unsigned tb = 8u * (unsigned)(wv * 64 + lane_id());
#pragma unroll
        for (int r = 0; r < 16; ++r) *at(sbk(dst, r * NT), tb) = x[r];
    }
}
__global__ void __launch_bounds__(NT, 8) ntt14_inv_kernel(NttBatch b) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int njobs = (b.nslots * b.nouter) << b.split;
#pragma unroll 1
    for (int job2 = blockIdx.x; job2 < njobs; job2 += gridDim.x) {
        kargptr kb = (kargptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kb));
        const int split = kb->split, job = job2 >> split, part = job2 & ((1 << split) - 1);
        const int nouter = kb->nouter;
        const int s = (int)udiv_magic((unsigned)job, kb->magic_nouter);
        int outer = job - s * nouter;
        const int m = kb->mod[s];
        JobInv jb;
        jb.nsum = 1; jb.g = outer; jb.pm = -1;
        if (kb->vi) {
            // as ntt_kernels.hip job_pointers<true> / vi_summand_offset
            const int cnt = kb->vi_cnt[outer];
            const unsigned mem = kb->vi_mem[outer];
            const int k = s < kb->vi_q ? 0 : (int)udiv_magic((unsigned)(s - kb->vi_q), kb->magic_vi_np);
            if (k >= cnt) continue;                          // a P slot of a member this group does not have (the whole workgroup skips it)
            const int item = (int)((mem >> (8 * k)) & 255u);
            const long off = (long)item * kb->src_outer + (long)m * kb->src_inner;
            jb.src = (gcptr)(kb->src + off); jb.dst = (gptr)(kb->dst + off);
            if (s < kb->vi_q) {
                int n = kb->vi_extra[outer] != nullptr ? 1 : 0;
                for (int j = 0; j < cnt; ++j) n += 1 + (int)(vi_parts_of(kb, (mem >> (8 * j)) & 255u) & 255u);
                jb.nsum = n;
            } else { jb.pm = k; jb.nsum = 1 + (int)(vi_parts_of(kb, item) & 255u); }
        } else {
            const int p = kb->pos[s];
            const u64* sbase_ = kb->src; u64* dbase_ = kb->dst;
            if (kb->nitems > 0) {
                const int item = (int)udiv_magic((unsigned)outer, kb->magic_opi);
                outer -= item * kb->outers_per_item;
                sbase_ = kb->src_items[item]; dbase_ = kb->dst_items[item];
            }
            jb.src = (gcptr)(sbase_ + (long)outer * kb->src_outer + (long)(kb->src_mapped ? m : p) * kb->src_inner);
            jb.dst = (gptr)(dbase_ + (long)outer * kb->dst_outer + (long)(kb->dst_mapped ? m : p) * kb->dst_inner);
        }
        jb.src += part * HH; jb.dst += part * HH;
        jb.root = (1 << split) + part;
        const long nlimb = (long)HH << split;                     // twiddle words per modulus: the rows of the whole limb
        jb.psi = kb->psi + (long)m * nlimb;
        jb.psi31 = kb->psi31 + 2 * (long)m * nlimb;
        jb.fin = kb->inv31c + ((long)m * 8 + jb.root) * 6;
        jb.mp = (smodptr)kb->mods + m;
#if defined(MKHE_INV_ONLY) && MKHE_INV_ONLY == 1
        limb_inv<true>(jb, false, lds, wv);
#elif defined(MKHE_INV_ONLY) && MKHE_INV_ONLY == 0
        limb_inv_cs(jb, lds, wv);
#else
        if ((kb->u_mods >> m) & 1) limb_inv<true>(jb, false, lds, wv);
        else limb_inv_cs(jb, lds, wv);
#endif
    }
}

}  // namespace h16

// ------------------------------------------------------------------ launcher
namespace {
struct LaunchState16 { std::mutex mu; int resident[64] = {}; };      // per device, see ntt_kernels.hip
}
namespace {
int resident16(size_t lds) {
    using namespace h16;
    static LaunchState16 ls;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(ls.mu);
    if (!ls.resident[dev & 63]) {
        (void)hipFuncSetAttribute((const void*)ntt16_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)ntt16_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)ntt16_fwd_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)ntt14_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)ntt14_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)ntt14_fwd_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int cus = 256, per = 1;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)ntt16_fwd_kernel<true>, NT, lds) != hipSuccess || per < 1) per = 1;
        // (grid size, LPT placement and half-limb jobs all rest on this figure: take the minimum over the kernels that share it)
        const void* others[] = {(const void*)ntt16_fwd_kernel<false>, (const void*)ntt16_fwd_split_kernel, (const void*)ntt14_fwd_kernel<true>,
                                (const void*)ntt14_fwd_kernel<false>, (const void*)ntt14_fwd_split_kernel};
        for (const void* f : others) { int p2 = per; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&p2, f, NT, lds) == hipSuccess && p2 >= 1 && p2 < per) per = p2; }
        ls.resident[dev & 63] = cus * per;
    }
    return ls.resident[dev & 63];
}
}
// launch constants of the job walk (fwd_body): reciprocals for the scalar multiply-high divisions and the placement of the long jobs
static int ab_ntt16() { static const int v = MKHE_AB_INT("MKHE_NTT16", 1); return v; }
static int ab_ntt16_min() { static const int v = MKHE_AB_INT("MKHE_NTT16_MIN", 128); return v; }
static unsigned magic_of(int d) { return d > 1 ? (unsigned)((1ull << 32) / (unsigned)d + 1) : 0u; }      // (0: divisor 1, see udiv_magic)
static void fill_job_constants(NttBatch& c, int njobs, int blocks, int lpt_long) {
    if (njobs >= 65536 || c.nouter >= 65536) throw std::runtime_error("mkhe: internal: an H16 launch of 2^16 limbs or more");
    c.magic_nouter = magic_of(c.nouter);
    c.magic_opi = magic_of(c.nitems > 0 ? c.outers_per_item : 1);
    c.lpt = NttBatch::Lpt{};
    const int C = blocks >> 1;
    if (lpt_long > 0 && C > 0 && blocks == 2 * C && njobs > blocks && njobs % C != 0) {
        const int r = njobs % C, w = C - r, q = njobs / C;             // CUs 0 .. r-1 own one position more; w light CUs; q complete rows
        const int B = lpt_long < q * w ? lpt_long : q * w;             // long jobs that find a light position
        c.lpt.B = B; c.lpt.C = C; c.lpt.r = r; c.lpt.full = B / w; c.lpt.rem = B - (B / w) * w; c.lpt.magic_C = magic_of(C);
    }
}
// sub-transforms of a split N = 2^16 launch (one modulus class per launch: `small` = 31 q < 2^62 for every slot)
bool ntt16_split_ok(const NttBatch& c) {
    const int on = ab_ntt16(), minl = ab_ntt16_min();
    // (the job walk divides by multiply-high reciprocals that are exact below 2^16: larger launches keep the round-1 kernels)
    return on && !c.no_h16 && c.psi31 && (c.split == 1 || c.split == 2) && !c.reduce_in && c.nslots <= 64 && 2 * c.nslots * c.nouter >= minl &&
           (long)c.nslots * c.nouter < 65536 && c.nouter < 65536;
}
void launch_ntt16_fwd_split(const NttBatch& b, bool small, hipStream_t st) {
    using namespace h16;
    NttBatch c = b;
    c.small_slots = small ? ~0ull : 0ull;
    c.lazy_out = 0;
    const size_t lds = (size_t)LDS_WORDS * sizeof(u32);
    const int resident = resident16(lds);
    const int need = (c.nslots * c.nouter) << c.split;
    fill_job_constants(c, c.nslots * c.nouter, 0, 0);
    if (c.split == 2) hipLaunchKernelGGL(ntt14_fwd_split_kernel, dim3(need < resident ? need : resident), dim3(NT), lds, st, c);
    else hipLaunchKernelGGL(ntt16_fwd_split_kernel, dim3(need < resident ? need : resident), dim3(NT), lds, st, c);
}
// inverse: launches of at least MKHE_NTT16_INV_MIN one-pass jobs (2^14 points each); the caller (launch_ntt_inv) runs the cross stages of
// N = 2^15 / 2^16 behind it.  b.psi31 / b.psi are the INVERSE tables here, b.inv31c the pairs of the last stage.
bool ntt16_inv_ok(int logN, const NttBatch& b) {
    static const int on = MKHE_AB_INT("MKHE_NTT16_INV", 1), minj = MKHE_AB_INT("MKHE_NTT16_INV_MIN", 256), minj14 = MKHE_AB_INT("MKHE_NTT14_INV_MIN", 129);
    if (!on || b.no_h16 || !b.psi31 || !b.inv31c || b.split || b.nslots > 64 || logN < 14 || logN > 16) return false;
    const int limbs = b.vi ? b.vi_jobs : b.nslots * b.nouter;
    // N = 2^14 (round 4): up to 128 limbs run as LDS sub-transforms (launch_ntt_inv), everything above comes here -- between 129 and 255 limbs
    // the launch used to fall to the round-1 register kernel, 69 us per launch whatever its size: the batched evaluation of the small rings
    // (batch.hip) lives in exactly that range
    if (logN == 14) return limbs >= minj14 && b.nslots * b.nouter < 65536;
    return (limbs << (logN - 14)) >= minj && b.nslots * b.nouter < 65536;
}
void launch_ntt16_inv(const NttBatch& b, hipStream_t st, int logN) {
    using namespace h16;
    NttBatch c = b;
    c.split = logN - 14;
    const size_t lds = (size_t)LDS_WORDS * sizeof(u32);
    static LaunchState16 ls;
    int dev = 0;
    (void)hipGetDevice(&dev);
    int resident = 0;
    {
        std::lock_guard<std::mutex> g(ls.mu);
        if (!ls.resident[dev & 63]) {
            (void)hipFuncSetAttribute((const void*)ntt14_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            int cus = 256, per = 1;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)ntt14_inv_kernel, NT, lds) != hipSuccess || per < 1) per = 1;
            ls.resident[dev & 63] = cus * per;
        }
        resident = ls.resident[dev & 63];
    }
    const int need = (c.nslots * c.nouter) << c.split;
    fill_job_constants(c, c.nslots * c.nouter, 0, 0);
    c.magic_vi_np = magic_of(c.vi ? c.vi_np : 1);
    hipLaunchKernelGGL(ntt14_inv_kernel, dim3(need < resident ? need : resident), dim3(NT), lds, st, c);
}
bool ntt16_ok(int logN, const NttBatch& b) {
    static const int minl14 = MKHE_AB_INT("MKHE_NTT14_MIN", 128);
    const int on = ab_ntt16(), minl = ab_ntt16_min();
    if (!on || b.no_h16 || !b.psi31 || b.split || b.prestaged || b.nslots > 64) return false;
    if ((long)b.nslots * b.nouter >= 65536 || b.nouter >= 65536) return false;      // (fill_job_constants: reciprocals exact below 2^16)
    if (logN == 14) return b.nslots * b.nouter >= minl14;      // (one pass: a workgroup has loaded its whole limb before it stores, in place included)
    return logN == 15 && b.nslots * b.nouter >= minl;
}
void launch_ntt16_fwd(const NttBatch& b, const unsigned char* small_q, hipStream_t st, int logN) {
    using namespace h16;
    NttBatch c = b;
    // big-modulus limbs (the longer jobs) first, as in launch_ntt_fwd_mixed
    c.small_slots = 0; c.nslots = 0;
    for (int cls = 0; cls < 2; ++cls)
        for (int s = 0; s < b.nslots; ++s)
            if ((small_q[b.mod[s]] != 0) == (cls == 1)) {
                c.mod[c.nslots] = b.mod[s]; c.pos[c.nslots] = b.pos[s];
                if (cls) c.small_slots |= 1ull << c.nslots;
                ++c.nslots;
            }
    const size_t lds = (size_t)LDS_WORDS * sizeof(u32);
    c.lazy_out = 0;                                    // (no start delay of the co-resident workgroups: measured, not kept -- docs/DESIGN_HISTORY.md)
    int nbig = 0;                                      // the long jobs lead the slot-major list: every slot of a 59/60-bit modulus, nouter limbs each
    for (int s2 = 0; s2 < c.nslots; ++s2) if (!((c.small_slots >> s2) & 1)) ++nbig;
    const int lpt_long = nbig < c.nslots ? nbig * c.nouter : 0;
    const int resident = resident16(lds);
    const int need = c.nslots * c.nouter;
    const int blocks = need < resident ? need : resident;
    // half-limb jobs: Decompose launches only (source = ciphertext limbs, destination = hoisted digits: never in place) whose whole limbs would
    // leave the last row of positions ragged (896 limbs on 256 CUs: 134.0 -> 128.7 us); a launch that deals whole limbs evenly keeps them -- as
    // half-limb jobs the 1792-limb launch (7 limbs per CU either way) is 5 % SLOWER (256 -> 270 us, measured twice on one box; MKHE_NTT16_HALVES=2
    // forces them for every launch)
    static const int halfj = MKHE_AB_INT("MKHE_NTT16_HALVES", 1);
    const int cus = resident >> 1;
    c.half_jobs = (halfj && logN == 15 && c.reduce_in && need > resident && 2 * need < 65536 && (halfj == 2 || (cus > 0 && need % cus != 0))) ? 1 : 0;
    if (c.half_jobs && need % 8 == 0 && lpt_long % 8 == 0) c.half_jobs = 2;      // 8 limbs x 2 passes per block of 16 positions (see fwd_body)
    fill_job_constants(c, c.half_jobs ? 2 * need : need, blocks, c.half_jobs ? 2 * lpt_long : lpt_long);
    if (logN == 14) {
        if (c.reduce_in) hipLaunchKernelGGL(ntt14_fwd_kernel<true>, dim3(blocks), dim3(NT), lds, st, c);
        else hipLaunchKernelGGL(ntt14_fwd_kernel<false>, dim3(blocks), dim3(NT), lds, st, c);
        return;
    }
    if (c.reduce_in) hipLaunchKernelGGL(ntt16_fwd_kernel<true>, dim3(blocks), dim3(NT), lds, st, c);
    else hipLaunchKernelGGL(ntt16_fwd_kernel<false>, dim3(blocks), dim3(NT), lds, st, c);
}

}  // namespace mkhe
