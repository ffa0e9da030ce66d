// h16_arith.h -- device arithmetic and addressing shared by the H16 / H32 forward and inverse NTT kernels (ntt16_kernels.hip, ntt32_kernels.hip):
// the saddr-form global loads / stores, the signed-digit Montgomery product and the one-round products mm31 / mm30u, the float-estimated
// partial reduction, the per-limb job description.  Device code only; see ntt16_kernels.hip for the derivations and DESIGN.md section 3.
#pragma once
#include "ntt_kernels.h"
#include <utility>

// per-lane twiddle words made opaque in front of the one-round products (0: left to the register coalescer -- ntt32_kernels.hip)
#ifndef MKHE_MM_VOPAQUE
#define MKHE_MM_VOPAQUE 1
#endif

namespace mkhe {
namespace h16 {

typedef const __attribute__((address_space(1))) u64* gcptr;
typedef __attribute__((address_space(1))) u64* gptr;
typedef const __attribute__((address_space(4))) u64* scptr;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) u64x2* gcptr2;
typedef const __attribute__((address_space(4))) NttBatch* kargptr;
typedef const __attribute__((address_space(4))) Mod* smodptr;

// Uniform (SGPR) base + 32-bit per-lane index: the `saddr` form of global_load / global_store.  The base is made opaque per access so
// that the compiler neither folds it into 64-bit per-lane addresses nor hoists 16 of those out of the pass loop (spills); the
// constant part of an address is added AFTER the opaque copy for the same reason (base + constant is loop invariant too).
template <class P> __device__ __forceinline__ P sbase(P p) { asm volatile("" : "+s"(p)); return p; }
// base + constant as ONE scalar value: opaque before the addition (so that it is not hoisted out of the pass loop and spilled) and
// after it (so that it is not re-associated into a 64-bit per-lane address)
template <class P> __device__ __forceinline__ P sbk(P p, long k) { return sbase(sbase(p) + k); }
// element `byte_off / 8` of a scalar base: the per-lane part of the address is a 32-bit BYTE offset, so that base + zext(offset) selects
// the SGPR-base (saddr) form of the global instructions whatever the compiler knows about the index range
// The pass-0 source loads and the parking stores are spelled out: left to the compiler, the second group of 16 loads is addressed with
// 64-bit per-lane additions (one multiplier-class VALU instruction per load) and some loaded values are spilled the moment they land
// (s_waitcnt vmcnt(0) + scratch_store between two loads).  ld_issue only issues; ld_wait16 is the one wait, and it takes the sixteen
// destinations as read-write operands so that no use can be scheduled above it.
__device__ __forceinline__ u64 ld_issue(gcptr base, unsigned byte_off) {
#ifdef MKHE_H16_X_NOSRC         // timing experiment only (wrong results): no source loads
    return (u64)byte_off * 0x9E3779B97F4A7C15ull;
#endif
    u64 v; asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(v) : "v"(byte_off), "s"(base)); return v;
}
__device__ __forceinline__ void ld_wait16(u64 (&a)[8], u64 (&b)[8]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                                         "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
}
__device__ __forceinline__ void st_issue(gptr base, unsigned byte_off, u64 v) {
    asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"(byte_off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ gcptr at(gcptr p, unsigned byte_off) { return (gcptr)((const __attribute__((address_space(1))) char*)p + byte_off); }
__device__ __forceinline__ gptr at(gptr p, unsigned byte_off) { return (gptr)((__attribute__((address_space(1))) char*)p + byte_off); }

// two consecutive twiddles in one 16-byte load
__device__ __forceinline__ void ld2(u64* out, gcptr2 base, unsigned idx) {
#ifdef MKHE_H16_X_NOTWLOAD      // timing experiment only (wrong results): no per-lane twiddle loads
    out[0] = idx; out[1] = idx + 1; return;
#endif
    // the byte offset is formed in 32 bits, so that base + zext(offset) selects the SGPR-base addressing form
    const u64x2 v = *(gcptr2)((const __attribute__((address_space(1))) char*)base + (unsigned)(idx * 16u));
    out[0] = v.x; out[1] = v.y;
}

// per-job constants, wave-uniform (SGPRs)
struct MC { i32 q0, q1; u32 ninv; u64 q; float finv; i32 p0, p1; /* radix-2^31 digits of q, balanced (mm31) */ };

// ------------------------------------------------------------------ signed-digit Montgomery product (modarith.h mont_mul_sd)
// a * w * 2^-64 mod q as a signed representative, |r| <= q/2 + |a| w / 2^64 + 1: 12 multiplier-class + 2 plain instructions.
// The two reduction rounds are spelled out as two asm blocks of v_mad_i64_i32 chains: written in C the compiler starts the
// products that do not depend on the carry word early and adds it with separate 64-bit additions (2 extra instructions per
// product), and written as one asm statement per instruction it pads every statement with an s_nop (it has to assume a
// transcendental result).  A block only names whole operands: the low word of the running sum that the next round multiplies by
// -q^-1 is passed in as its own 32-bit operand, which is why the chain is cut there.
// SW: the twiddle halves are SGPRs (phases A, B, stage 0, normalisation) or VGPRs (phases C, D); q0, q1, -q^-1 are always SGPRs.
template <bool SW> __device__ __forceinline__ i64 mm(i64 a, u64 ws, const MC& c) {
    const u32 al = lo32((u64)a);
    const i32 a0 = (i32)al;
    const i32 a1 = (i32)(hi32((u64)a) + (al >> 31));
    i32 w0 = (i32)lo32(ws), w1 = (i32)hi32(ws);
    // opaque 32-bit values: seen as the halves of a 64-bit constant the compiler multiplies by them as 64-bit values
    if constexpr (SW) asm("" : "+s"(w0), "+s"(w1)); else asm("" : "+v"(w0), "+v"(w1));
    i64 acc = (i64)a0 * w0;                                  // v_mad_i64_i32 acc, a0, w0, 0
    i32 m, m2; u64 k;
    if constexpr (SW) {
        asm("v_mul_lo_u32 %1, %3, %7\n\t"                   // m = lo(acc) * -q^-1
            "v_mad_i64_i32 %0, %2, %1, %8, %0\n\t"          // S = m*q0 + acc (low word zero)
            "v_ashrrev_i64 %0, 32, %0\n\t"
            "v_mad_i64_i32 %0, %2, %4, %6, %0\n\t"          // + a0*w1
            "v_mad_i64_i32 %0, %2, %1, %9, %0\n\t"          // + m*q1
            "v_mad_i64_i32 %0, %2, %5, %10, %0"              // + a1*w0
            : "+v"(acc), "=&v"(m), "=&s"(k)
            : "v"(lo32((u64)acc)), "v"(a0), "v"(a1), "s"(w1), "s"(c.ninv), "s"(c.q0), "s"(c.q1), "s"(w0));
        asm("v_mul_lo_u32 %1, %3, %6\n\t"
            "v_mad_i64_i32 %0, %2, %1, %7, %0\n\t"
            "v_ashrrev_i64 %0, 32, %0\n\t"
            "v_mad_i64_i32 %0, %2, %4, %5, %0\n\t"          // + a1*w1
            "v_mad_i64_i32 %0, %2, %1, %8, %0"               // + m2*q1
            : "+v"(acc), "=&v"(m2), "=&s"(k)
            : "v"(lo32((u64)acc)), "v"(a1), "s"(w1), "s"(c.ninv), "s"(c.q0), "s"(c.q1));
    } else {
        asm("v_mul_lo_u32 %1, %3, %7\n\t"
            "v_mad_i64_i32 %0, %2, %1, %8, %0\n\t"
            "v_ashrrev_i64 %0, 32, %0\n\t"
            "v_mad_i64_i32 %0, %2, %4, %6, %0\n\t"
            "v_mad_i64_i32 %0, %2, %1, %9, %0\n\t"
            "v_mad_i64_i32 %0, %2, %5, %10, %0"
            : "+v"(acc), "=&v"(m), "=&s"(k)
            : "v"(lo32((u64)acc)), "v"(a0), "v"(a1), "v"(w1), "s"(c.ninv), "s"(c.q0), "s"(c.q1), "v"(w0));
        asm("v_mul_lo_u32 %1, %3, %6\n\t"
            "v_mad_i64_i32 %0, %2, %1, %7, %0\n\t"
            "v_ashrrev_i64 %0, 32, %0\n\t"
            "v_mad_i64_i32 %0, %2, %4, %5, %0\n\t"
            "v_mad_i64_i32 %0, %2, %1, %8, %0"
            : "+v"(acc), "=&v"(m2), "=&s"(k)
            : "v"(lo32((u64)acc)), "v"(a1), "v"(w1), "s"(c.ninv), "s"(c.q0), "s"(c.q1));
    }
    return acc;
}

// ------------------------------------------------------------------ one-round product (round 2)
// The twiddle enters as TWO pre-reduced constants, u = w 2^31 mod q and v = w 2^63 mod q (balanced, radix-2^31 digits u = u1 2^31 + u0,
// |u0| <= 2^30: NttBatch::psi31).  With a = a1 2^32 + a0 (balanced digits, as in mm):  a w 2^31 = a0 u + a1 v  (mod q) is a 92-bit
// number, and ONE Montgomery round of radix 2^31 brings it back to 61 bits where the 128-bit product a * (w 2^64) of mm needs two:
//   C = a0 u0 + a1 v0 ; m = balanced31(lo(C) * -q^-1) ; T = ((C + m p0) >> 31) + a0 u1 + a1 v1 + m p1  =  (a0 u + a1 v + m q) / 2^31
// = a w mod q exactly, |T| <= q + |a| q / 2^64 (|a0 u| / 2^31 <= q/2, |m q| / 2^31 <= q/2).  Every column sum stays below 2^63 for
// |a| < 2^62.9: 2^61 + 2^61 + 2^60 in column 0.  8 multiplier-class + 1 plain instruction against 12: the bare butterfly goes from
// 73 to 61 cycles per wave and the clock from 1.98 to 2.10 GHz (tools/ubench/bfly31_rate.hip: 36.8 -> 28.8 ns).  The price: twice
// the twiddle words, and values that grow by q (not q/2) per stage -- see the range notes in limb().
template <bool SW> __device__ __forceinline__ i64 mm31(i64 a, u64 us, u64 vs, const MC& c) {
    const u32 al = lo32((u64)a);
    const i32 a0 = (i32)al;
    const i32 a1 = (i32)(hi32((u64)a) + (al >> 31));
    i32 u0 = (i32)lo32(us), u1 = (i32)hi32(us), v0 = (i32)lo32(vs), v1 = (i32)hi32(vs);
    if constexpr (SW) asm("" : "+s"(u0), "+s"(u1), "+s"(v0), "+s"(v1)); else if (MKHE_MM_VOPAQUE) asm("" : "+v"(u0), "+v"(u1), "+v"(v0), "+v"(v1));
    i64 acc = (i64)a0 * u0;                                  // v_mad_i64_i32 acc, a0, u0, 0
    i32 m; u64 k;
    if constexpr (SW) {
        asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=&s"(k) : "v"(a1), "s"(v0));          // + a1 * v0
        asm("v_mul_lo_u32 %1, %3, %7\n\t"                  // lo(C) * -q^-1
            "v_bfe_i32 %1, %1, 0, 31\n\t"                  // balanced 31-bit digit
            "v_mad_i64_i32 %0, %2, %1, %8, %0\n\t"         // + m * p0: low 31 bits zero
            "v_ashrrev_i64 %0, 31, %0\n\t"
            "v_mad_i64_i32 %0, %2, %4, %6, %0\n\t"         // + a0 * u1
            "v_mad_i64_i32 %0, %2, %5, %10, %0\n\t"        // + a1 * v1
            "v_mad_i64_i32 %0, %2, %1, %9, %0"               // + m * p1
            : "+v"(acc), "=&v"(m), "=&s"(k)
            : "v"(lo32((u64)acc)), "v"(a0), "v"(a1), "s"(u1), "s"(c.ninv), "s"(c.p0), "s"(c.p1), "s"(v1));
    } else {
        asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=&s"(k) : "v"(a1), "v"(v0));
        asm("v_mul_lo_u32 %1, %3, %7\n\t"
            "v_bfe_i32 %1, %1, 0, 31\n\t"
            "v_mad_i64_i32 %0, %2, %1, %8, %0\n\t"
            "v_ashrrev_i64 %0, 31, %0\n\t"
            "v_mad_i64_i32 %0, %2, %4, %6, %0\n\t"
            "v_mad_i64_i32 %0, %2, %5, %10, %0\n\t"
            "v_mad_i64_i32 %0, %2, %1, %9, %0"
            : "+v"(acc), "=&v"(m), "=&s"(k)
            : "v"(lo32((u64)acc)), "v"(a0), "v"(a1), "v"(u1), "s"(c.ninv), "s"(c.p0), "s"(c.p1), "v"(v1));
    }
    return acc;
}

// ------------------------------------------------------------------ one-round product with an UNSIGNED low data digit (round 3)
// For the moduli with 160 q < 2^62 (the 2^54 - delta primes of PN15QP880, the 45-bit ones of PN16QP1761: "U class") the data word is taken
// as it stands, a = hi 2^32 + lo with hi signed and lo UNSIGNED: no digit fix-up (v_lshrrev + v_add per product) in front of the chain.
// The twiddle pair is then  u = w 2^30 mod q in [0, q)  (radix-2^30 digits u0, u1 >= 0: they meet lo in v_mad_u64_u32)  and
// v = w 2^62 mod q, balanced (radix-2^30 digits, signed: they meet hi in v_mad_i64_i32); one Montgomery round of radix 2^30:
//   C = lo u0 + hi v0 ; m = balanced30(lo(C) * -q^-1) ; T = ((C + m p0) >> 30) + lo u1 + hi v1 + m p1  =  (lo u + hi v + m q) / 2^30 = a w mod q.
// Columns: lo u0 < 2^62, |hi v0| < 2^60, |m p0| < 2^58; lo u1 < 2^32 q / 2^30, ... all far below 2^63 for |a| < 2^62.  The price is the range:
// lo u / 2^30 < 4 q, |hi v| / 2^30 < |a| q / 2^63 < q / 2, |m q| / 2^30 <= q / 2, so T lies in (-q, 5q) and the never-reduced values grow by up to
// 5q on either side per stage -- 75q over the 15 stages of a pass, which is why only moduli with (4 + 75 + 75) q < 2^62 take this path.
// 8 multiplier-class + 1 plain instruction like mm31, but 2 plain instructions less in front: tools/ubench/bfly30u_rate.hip.
template <bool SW> __device__ __forceinline__ i64 mm30u(i64 a, u64 us, u64 vs, const MC& c) {
    const u32 lo = lo32((u64)a);
    const i32 hi = (i32)hi32((u64)a);
    i32 u0 = (i32)lo32(us), u1 = (i32)hi32(us), v0 = (i32)lo32(vs), v1 = (i32)hi32(vs);
    if constexpr (SW) asm("" : "+s"(u0), "+s"(u1), "+s"(v0), "+s"(v1)); else if (MKHE_MM_VOPAQUE) asm("" : "+v"(u0), "+v"(u1), "+v"(v0), "+v"(v1));
    i64 acc; i32 m; u64 k;
    // (the chain is cut after the low column: a block only names whole operands, and the multiply by -q^-1 reads the low word of the sum)
    if constexpr (SW) {
        asm("v_mad_u64_u32 %0, %1, %2, %4, 0\n\t"           // lo * u0
            "v_mad_i64_i32 %0, %1, %3, %5, %0"                // + hi * v0
            : "=&v"(acc), "=&s"(k) : "v"(lo), "v"(hi), "s"(u0), "s"(v0));
        asm("v_mul_lo_u32 %1, %3, %8\n\t"                   // lo(C) * -q^-1
            "v_bfe_i32 %1, %1, 0, 30\n\t"                   // balanced 30-bit digit
            "v_mad_i64_i32 %0, %2, %1, %9, %0\n\t"          // + m * p0: low 30 bits zero
            "v_ashrrev_i64 %0, 30, %0\n\t"
            "v_mad_u64_u32 %0, %2, %4, %6, %0\n\t"          // + lo * u1
            "v_mad_i64_i32 %0, %2, %5, %7, %0\n\t"          // + hi * v1
            "v_mad_i64_i32 %0, %2, %1, %10, %0"               // + m * p1
            : "+v"(acc), "=&v"(m), "=&s"(k)
            : "v"(lo32((u64)acc)), "v"(lo), "v"(hi), "s"(u1), "s"(v1), "s"(c.ninv), "s"(c.p0), "s"(c.p1));
    } else {
        asm("v_mad_u64_u32 %0, %1, %2, %4, 0\n\t"
            "v_mad_i64_i32 %0, %1, %3, %5, %0"
            : "=&v"(acc), "=&s"(k) : "v"(lo), "v"(hi), "v"(u0), "v"(v0));
        asm("v_mul_lo_u32 %1, %3, %8\n\t"
            "v_bfe_i32 %1, %1, 0, 30\n\t"
            "v_mad_i64_i32 %0, %2, %1, %9, %0\n\t"
            "v_ashrrev_i64 %0, 30, %0\n\t"
            "v_mad_u64_u32 %0, %2, %4, %6, %0\n\t"
            "v_mad_i64_i32 %0, %2, %5, %7, %0\n\t"
            "v_mad_i64_i32 %0, %2, %1, %10, %0"
            : "+v"(acc), "=&v"(m), "=&s"(k)
            : "v"(lo32((u64)acc)), "v"(lo), "v"(hi), "v"(u1), "v"(v1), "s"(c.ninv), "s"(c.p0), "s"(c.p1));
    }
    return acc;
}

// Cheap partial reduction: x -> x - round(x / q) * q, |result| <= q/2 + q * 2^-19, for any |x| < 2^62.9.  The quotient is estimated
// from the high word in float32 (3 plain VALU instructions: convert, fma with the 1.5 * 2^23 rounding constant, subtract) and is at
// most a few dozen, so the product is one v_mad_i64_i32 for the low digit of q plus a 32-bit multiply-add for the high digit:
// 7 instructions, 3 of them multiplier-class, against 14 / 12 for a Montgomery product by R mod q.
__device__ __forceinline__ i64 pred(i64 x, const MC& c) {
    const float f = __builtin_fmaf((float)(i32)hi32((u64)x), c.finv, 12582912.0f);
    const i32 nt = (i32)(0x4B400000u - __builtin_bit_cast(u32, f));              // -round(x / q)
    i64 y = (i64)nt * c.q0 + x;                                                   // v_mad_i64_i32
    return (i64)((u64)y + ((u64)(u32)(nt * c.q1) << 32));                         // high word += nt * q1 (v_mul_lo_u32 + v_add_u32)
}


template <bool CROSS> __device__ __forceinline__ void xsync() {
    if constexpr (CROSS) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
}
// lane index, recomputed where it is needed (two plain VALU instructions) instead of a register that lives -- or is spilled and
// reloaded -- across the whole limb; the wave index is an SGPR
__device__ __forceinline__ int lane_id() {
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));            // not hoisted, not common-subexpression'd across phases
    return l;
}
// n / d for 0 <= n < 2^16, 2 <= d < 2^16 as ONE scalar multiply-high: magic = floor(2^32 / d) + 1 (host: magic_of), exact in that range;
// d = 1 has no 32-bit magic (2^32 + 1): magic_of returns 0 for it and the quotient is n itself
// (the error term n * (magic * d - 2^32) / (d * 2^32) stays below 1 / d)
__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned magic) { return magic ? (unsigned)(((unsigned long long)n * magic) >> 32) : n; }
// what one limb needs, all wave-uniform
struct Job { gcptr src; gptr dst; const u64* psi; const u64* psi31; const u64* psi31n; smodptr mp; bool red; bool skip_norm; int root; u64* trace; int sched; const u64* psif; };
}  // namespace h16
}  // namespace mkhe
