// modarith.h -- 64-bit Montgomery arithmetic for gfx950 (device side).
//
// CDNA4 has no 64x64->128 multiply: every product is built from v_mad_u64_u32
// (32x32+64 -> 64) and v_mul_lo_u32, both measured at half the rate of a simple VALU op
// (tools/ubench/imul_rate.hip).  The Montgomery product below is word-serial (radix 2^32,
// two rounds) so that every addition rides on a mad's 64-bit addend (see mont_mul_lazy).
//
// Semantics mirror lattigo v2.3.0 ring.MRed / MRedConstant / MForm / CRed as used by the
// reference (mkrlwe/keyswitch_hoisted.go:28-30, basis_extension.go:220,551): radix R = 2^64,
//   mont_mul(a, b) = a*b*R^-1 mod q.
// Intermediate (lazy) representatives may differ from lattigo's; every value that leaves a
// kernel for another modulus or for the caller is canonical or an explicitly restated
// lazy formula (mult_sum).
#pragma once
#include "switches.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mkhe {

typedef uint64_t u64;
typedef uint32_t u32;
typedef int64_t i64;
typedef int32_t i32;

// Per-modulus constants, uniform per workgroup (live in SGPRs).
struct Mod {
    u64 q;        // modulus, < 2^60 (enforced by the Context constructor)
    u64 q2;       // 2q
    u32 ninv32;   // -q^-1 mod 2^32
    u32 finv;     // float32 bits of 2^32 / q: quotient estimate of the cheap partial reduction of ntt16_kernels.hip
    u64 qinv;     // q^-1 mod 2^64   (lattigo MRedParams; used by the literal mult_sum)
    u64 r1;       // 2^64  mod q     (MForm(1))
    u64 r2;       // 2^128 mod q     (mont_mul(a, r2) = MForm(a))
    u64 qs;       // q  in signed-split form (sd_split), for mont_mul_sd
    u64 r1s;      // r1 in signed-split form
};

// Signed-split form of a 64-bit constant v: the pair (hi, lo) with v = (i32)hi * 2^32 + (i32)lo, i.e. the high word
// absorbs the carry of reading the low word as signed.  Twiddle tables and per-modulus constants of the NTT kernels
// are stored like this (host side: Context constructor).
__host__ __device__ inline u64 sd_split(u64 v) { return v + ((u64)((u32)v >> 31) << 32); }

__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) { return (u64)a * b + c; }
__device__ __forceinline__ u32 lo32(u64 x) { return (u32)x; }
__device__ __forceinline__ u32 hi32(u64 x) { return (u32)(x >> 32); }

// A wave-uniform 1 the compiler cannot see through: mad64(x, one, acc) then stays ONE v_mad_u64_u32 that adds
// a 32-bit VGPR to a 64-bit accumulator.  Written as acc + x it is lowered to a 64-bit add whose 32-bit operand
// must first be widened into an even-aligned register pair (two v_mov_b32 per addend: about a third of all
// VALU instructions of the NTT kernels before this form; every VALU instruction costs one issue slot here).
// K only tells the instances apart: with ONE shared value the compiler factors x*one + y*one into (x + y)*one
// and is back to widening adds.
template <int K> __device__ __forceinline__ u32 opaque_one() { u32 o; asm("s_mov_b32 %0, 1 ; %1" : "=s"(o) : "i"(K)); return o; }

// a*w*R^-1 mod q, result in [0, 2q).  Requires a < 2^62, w < q < 2^61.
// Word-serial Montgomery (radix 2^32, two rounds), 14 v_mad_u64_u32 + 2 v_mul_lo_u32 and nothing else:
// every addition rides on a mad's 64-bit addend.
__device__ __forceinline__ u64 mont_mul_lazy(u64 a, u64 w, u64 q, u32 ninv32) {
    const u32 a0 = lo32(a), a1 = hi32(a), w0 = lo32(w), w1 = hi32(w), q0 = lo32(q), q1 = hi32(q);
    const u32 one_a = opaque_one<0>(), one_b = opaque_one<1>(), one_c = opaque_one<2>();
    // round 0: T = (a0*w + m*q) >> 32 with m = -a0*w0/q mod 2^32
    const u64 p0 = mad64(a0, w0, 0);
    const u32 m = lo32(p0) * ninv32;
    const u64 r0 = mad64(lo32(p0), one_a, mad64(m, q0, 0));             // low word is zero by construction
    const u64 T = mad64(hi32(r0), one_b, mad64(hi32(p0), one_c, mad64(m, q1, mad64(a0, w1, 0))));
    // round 1: result = (T + a1*w + m2*q) >> 32
    const u64 s0 = mad64(a1, w0, T);
    const u32 m2 = lo32(s0) * ninv32;
    const u64 r2 = mad64(lo32(s0), one_a, mad64(m2, q0, 0));
    return mad64(hi32(r2), one_b, mad64(hi32(s0), one_c, mad64(m2, q1, mad64(a1, w1, 0))));
}

// d = a * b + c as ONE v_mad_i64_i32 (the carry-out SGPR pair is unused)
__device__ __forceinline__ i64 mad_i64(i32 a, i32 b, i64 c) {
    i64 d;
    asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
    return d;
}

// Signed word-serial Montgomery product for the NTT kernels: a*w*R^-1 mod q as a SIGNED representative r with
// |r| <= q/2 + |a|*w/2^64 + 1.  a: any signed 64-bit value with |a| < 2^62; ws, qs: signed-split forms of w < q < 2^61.
// With balanced 32-bit digits every partial sum fits a signed 64-bit accumulator, so each round is
//   P = a_i*w0 [+ carry-in]; m = lo(P)*(-q^-1); S = m*q0 + P (low word 0); next = m*q1 + a_i*w1 + (S >> 32)
// = 4 v_mad_i64_i32 + 1 v_mul_lo_u32 + 1 v_ashrrev_i64 and nothing else (no widening moves, no carry chains): 12
// multiplier-class instructions + 2 plain ones for the digit split of a, against 16 for mont_mul_lazy.
__device__ __forceinline__ i64 mont_mul_sd(i64 a, u64 ws, u64 qs, u32 ninv32) {
    const i32 a0 = (i32)lo32((u64)a);
    const i32 a1 = (i32)(hi32((u64)a) + (lo32((u64)a) >> 31));
    i32 w0 = (i32)lo32(ws), q0 = (i32)lo32(qs), w1 = (i32)hi32(ws), q1 = (i32)hi32(qs);
    // opaque 32-bit values: seen as the sign-extended halves of a wave-uniform 64-bit constant, the compiler
    // multiplies by them as 64-bit values (4 instructions instead of one v_mad_i64_i32)
    asm("" : "+v"(w0), "+v"(w1), "+v"(q0), "+v"(q1));
    // the shift amount is opaque: knowing that S >> 32 is a sign-extended 32-bit value, the compiler splits the
    // 64-bit multiply-adds that consume it into 32-bit pieces (3x the instructions)
    u32 sh; asm("s_mov_b32 %0, 32" : "=s"(sh));
#ifdef MKHE_ASM_MAD
    // Experiment (tools/ubench/bfly_rate.hip -DMKHE_ASM_MAD): written in C the compiler starts the products that do not
    // depend on S early with a zero addend and adds S >> 32 with a separate v_lshl_add_u64 afterwards (two extra
    // multiplier-class instructions per product).  Spelling the chain out with inline asm takes the bare butterfly from 85 to
    // 77 cycles, but in the 128-VGPR NTT kernels the asm operands raise the scratch from 76 to 132 B per lane and the kernel
    // gets SLOWER (231 -> 243 us per launch): not enabled.
    const i64 P0 = mad_i64(a0, w0, 0);
    const i32 m = (i32)(lo32((u64)P0) * ninv32);
    const i64 S = mad_i64(m, q0, P0);                    // |S| < 2^63 (q0 is odd, so |q0| < 2^31); low word is zero
    const i64 Y = mad_i64(m, q1, mad_i64(a0, w1, S >> sh));
    const i64 U = mad_i64(a1, w0, Y);
    const i32 m2 = (i32)(lo32((u64)U) * ninv32);
    const i64 S2 = mad_i64(m2, q0, U);
#if MKHE_ASM_MAD >= 2
    return mad_i64(m2, q1, mad_i64(a1, w1, S2 >> sh));
#else
    return (i64)m2 * q1 + ((i64)a1 * w1 + (S2 >> sh));      // second round left to the compiler (fusing it too costs 48 B more scratch)
#endif
#else
    const i64 P0 = (i64)a0 * w0;
    const i32 m = (i32)(lo32((u64)P0) * ninv32);
    const i64 S = (i64)m * q0 + P0;                      // |S| < 2^63 (q0 is odd, so |q0| < 2^31); low word is zero
    const i64 Y = (i64)m * q1 + ((i64)a0 * w1 + (S >> sh));
    const i64 U = (i64)a1 * w0 + Y;
    const i32 m2 = (i32)(lo32((u64)U) * ninv32);
    const i64 S2 = (i64)m2 * q0 + U;
    return (i64)m2 * q1 + ((i64)a1 * w1 + (S2 >> sh));
#endif
}
// the same as an unsigned lazy representative in [0, 2q) (needs |a|*w < q*2^63, e.g. 0 <= a < 4q, q < 2^60)
__device__ __forceinline__ u64 mont_mul_sdu(u64 a, u64 ws, u64 qs, u64 q, u32 ninv32) {
    return (u64)(mont_mul_sd((i64)a, ws, qs, ninv32) + (i64)q);
}

__device__ __forceinline__ u64 csub(u64 a, u64 q) { return a >= q ? a - q : a; }

// canonical product (lattigo MRed)
__device__ __forceinline__ u64 mont_mul(u64 a, u64 w, u64 q, u32 ninv32) {
    return csub(mont_mul_lazy(a, w, q, ninv32), q);
}

// full 128-bit product
__device__ __forceinline__ void mul64x64(u64 a, u64 b, u64& hi, u64& lo) {
    const u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    u64 p00 = mad64(a0, b0, 0);
    u64 p01 = mad64(a0, b1, hi32(p00));
    u64 p10 = mad64(a1, b0, lo32(p01));
    u64 p11 = mad64(a1, b1, (u64)hi32(p01) + hi32(p10));
    lo = ((u64)lo32(p10) << 32) | lo32(p00);
    hi = p11;
}
__device__ __forceinline__ u64 mulhi64(u64 a, u64 b) { u64 h, l; mul64x64(a, b, h, l); return h; }

}  // namespace mkhe
