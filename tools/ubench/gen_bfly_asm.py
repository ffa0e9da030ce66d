#!/usr/bin/env python3
"""Generates tools/ubench/bfly_asm_rate.hip (round 4): the U-class butterfly (one-round product mm30u + add + subtract) as ONE hand-allocated asm
block -- data x[r] in v[2r:2r+1], product temporaries in fixed registers, twiddles in SGPRs -- against the form the compiler emits around the
two-statement inline asm of h16_arith.h (s_nop at every asm boundary, s_nop 1 between v_sub_co and v_subb: the 2 wait states gfx940 asks for between a
VALU write of VCC and a VALU read of it).  VARIANT 0: no nops, the next butterfly's first two multiply-adds between v_sub_co and v_subb;
1: the same instruction order with the compiler's nops put back.  Prints cycles per wave-butterfly.  python tools/ubench/gen_bfly_asm.py"""
import os
HERE = os.path.dirname(os.path.abspath(__file__))
def body(variant):
    L = []
    A = "v[40:41]"; Al = "v40"; M = "v42"; B_ = "v[44:45]"; Bl = "v44"; Bh = "v45"; M2 = "v46"
    # 4 stages x 8 butterflies on 16 registers; twiddle g>>B of stage: SGPRs s[20+4k .. 23+4k] = u0,u1,v0,v1 of twiddle k (k < 8)
    bf = []
    for Bb in (3, 2, 1, 0):
        for g in range(8):
            i0 = ((g >> Bb) << (Bb + 1)) | (g & ((1 << Bb) - 1)); i1 = i0 | (1 << Bb)
            bf.append((i0, i1, g >> Bb))
    pend = None      # (subb text) of the previous butterfly, to be placed after this one's first two multiply-adds
    tmp = [(A, Al, "v41", M), (B_, Bl, Bh, M2)]
    for n, (i0, i1, k) in enumerate(bf):
        acc, accl, acch, m = tmp[n & 1]
        U = "v[%d:%d]" % (2 * i0, 2 * i0 + 1); Ul, Uh = "v%d" % (2 * i0), "v%d" % (2 * i0 + 1)
        Vl, Vh = "v%d" % (2 * i1), "v%d" % (2 * i1 + 1)
        u0, u1, v0, v1 = ("s%d" % (20 + 4 * k + j) for j in range(4))
        L.append("v_mad_u64_u32 %s, s[16:17], %s, %s, 0" % (acc, Vl, u0))
        L.append("v_mad_i64_i32 %s, s[16:17], %s, %s, %s" % (acc, Vh, v0, acc))
        if pend:
            L.append(pend); pend = None
        if variant == 1: L.append("s_nop 0")
        L.append("v_mul_lo_u32 %s, %s, s12" % (m, accl))
        L.append("v_bfe_i32 %s, %s, 0, 30" % (m, m))
        L.append("v_mad_i64_i32 %s, s[16:17], %s, s13, %s" % (acc, m, acc))
        L.append("v_ashrrev_i64 %s, 30, %s" % (acc, acc))
        L.append("v_mad_u64_u32 %s, s[16:17], %s, %s, %s" % (acc, Vl, u1, acc))
        L.append("v_mad_i64_i32 %s, s[16:17], %s, %s, %s" % (acc, Vh, v1, acc))
        L.append("v_mad_i64_i32 %s, s[16:17], %s, s14, %s" % (acc, m, acc))
        if variant == 1: L.append("s_nop 0")
        L.append("v_sub_co_u32 %s, vcc, %s, %s" % (Vl, Ul, accl))
        if variant == 1:
            L.append("s_nop 1")
            L.append("v_subb_co_u32 %s, vcc, %s, %s, vcc" % (Vh, Uh, acch))
            L.append("v_lshl_add_u64 %s, %s, 0, %s" % (U, acc, U))
        else:
            # U' = U + T after the subtraction's low half has read U's low word; the high half of the subtraction reads U's OLD high word: take it first
            L.append("v_mov_b32 v47, %s" % Uh) if False else None
            pend = "v_subb_co_u32 %s, vcc, %s, %s, vcc" % (Vh, "v%d" % (48 + (n & 1)), acch)
            L.append("v_mov_b32 v%d, %s" % (48 + (n & 1), Uh))      # old high word of U (plain instruction; see the note in main)
            L.append("v_lshl_add_u64 %s, %s, 0, %s" % (U, acc, U))
    if pend: L.append("s_nop 1"); L.append(pend)
    return [x for x in L if x]
def body2(variant):
    """variant 2: subtraction placed BEFORE the addition needs no copy: V = U - T (both halves, two independent instructions in between), then U += T"""
    L = []
    tmp = [("v[40:41]", "v40", "v41", "v42"), ("v[44:45]", "v44", "v45", "v46")]
    bf = []
    for Bb in (3, 2, 1, 0):
        for g in range(8):
            i0 = ((g >> Bb) << (Bb + 1)) | (g & ((1 << Bb) - 1)); i1 = i0 | (1 << Bb)
            bf.append((i0, i1, g >> Bb))
    pend = []
    for n, (i0, i1, k) in enumerate(bf):
        acc, accl, acch, m = tmp[n & 1]
        U = "v[%d:%d]" % (2 * i0, 2 * i0 + 1); Ul, Uh = "v%d" % (2 * i0), "v%d" % (2 * i0 + 1)
        Vl, Vh = "v%d" % (2 * i1), "v%d" % (2 * i1 + 1)
        u0, u1, v0, v1 = ("s%d" % (20 + 4 * k + j) for j in range(4))
        L.append("v_mad_u64_u32 %s, s[16:17], %s, %s, 0" % (acc, Vl, u0))
        L.append("v_mad_i64_i32 %s, s[16:17], %s, %s, %s" % (acc, Vh, v0, acc))
        L += pend; pend = []
        L.append("v_mul_lo_u32 %s, %s, s12" % (m, accl))
        L.append("v_bfe_i32 %s, %s, 0, 30" % (m, m))
        L.append("v_mad_i64_i32 %s, s[16:17], %s, s13, %s" % (acc, m, acc))
        L.append("v_ashrrev_i64 %s, 30, %s" % (acc, acc))
        L.append("v_mad_u64_u32 %s, s[16:17], %s, %s, %s" % (acc, Vl, u1, acc))
        L.append("v_mad_i64_i32 %s, s[16:17], %s, %s, %s" % (acc, Vh, v1, acc))
        L.append("v_mad_i64_i32 %s, s[16:17], %s, s14, %s" % (acc, m, acc))
        L.append("v_sub_co_u32 %s, vcc, %s, %s" % (Vl, Ul, accl))
        # two wait states between the write of VCC and its use: the next butterfly's first two multiply-adds (they write s[16:17], not VCC)
        pend = ["v_subb_co_u32 %s, vcc, %s, %s, vcc" % (Vh, Uh, acch), "v_lshl_add_u64 %s, %s, 0, %s" % (U, acc, U)]
    L.append("s_nop 1"); L += pend
    return L
SRC = r'''// GENERATED by tools/ubench/gen_bfly_asm.py -- see there.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef unsigned long long u64;
template <int V> __global__ void __launch_bounds__(256) k(u64* out, const u64* tw, unsigned reps, unsigned long long* clk) {
    u64 acc;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (V == 0) asm volatile(
@B0@
        : "=v"(acc) : "s"(tw), "s"(reps), "v"(threadIdx.x) : @CLOB@);
    else if (V == 1) asm volatile(
@B1@
        : "=v"(acc) : "s"(tw), "s"(reps), "v"(threadIdx.x) : @CLOB@);
    else asm volatile(
@B2@
        : "=v"(acc) : "s"(tw), "s"(reps), "v"(threadIdx.x) : @CLOB@);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int V> int run(const char* name, int blocks) {
    u64 *out, *tw; unsigned long long* clk;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8)); CHECK(hipMalloc(&tw, 1024)); CHECK(hipMalloc(&clk, blocks * 16));
    const u64 q = 0x3fffffffd60001ull; u64 qi = q; for (int i = 0; i < 6; ++i) qi *= 2 - q * qi;
    unsigned h[256]; for (int i = 0; i < 256; ++i) h[i] = (0x9E3779B9u * (i + 1)) & 0x3fffffff;
    h[0] = (unsigned)(0 - qi); h[1] = (unsigned)(q & 0x3fffffff); h[2] = (unsigned)(q >> 30);      // s12, s13, s14 = -q^-1, p0, p1
    CHECK(hipMemcpy(tw, h, sizeof(h), hipMemcpyHostToDevice));
    const unsigned reps = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k<V><<<blocks, 256>>>(out, tw, reps, clk); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); k<V><<<blocks, 256>>>(out, tw, reps, clk); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    static unsigned long long hc[16384]; CHECK(hipMemcpy(hc, clk, blocks * 16, hipMemcpyDeviceToHost));
    double ticks = 0, rt = 0; for (int i = 0; i < blocks; ++i) { ticks += hc[2 * i]; rt += hc[2 * i + 1]; }
    const double ghz = ticks / rt / 10.0, wps = (double)blocks * 4 / (256.0 * 4.0);
    const double bf_per_simd = (double)blocks * 4 * reps * 32 / (256.0 * 4.0);
    printf("%-34s waves/SIMD %4.1f  %7.3f ms  clock %.2f GHz  %6.1f cycles / wave-butterfly  %5.1f ns\n", name, wps, ms, ghz, ms * 1e6 * ghz / bf_per_simd, ms * 1e6 / bf_per_simd);
    return 0;
}
int main() {
    for (int blocks : {256, 1024, 2048}) {
        run<1>("compiler's form (nops)", blocks);
        run<0>("no nops, copy of U's high word", blocks);
        run<2>("no nops, subtract before add", blocks);
    }
    return 0;
}
'''
def wrap(lines):
    pro = ["s_load_dwordx4 s[12:15], %1, 0", "s_load_dwordx16 s[20:35], %1, 64", "s_load_dwordx16 s[36:51], %1, 128", "s_mov_b32 s18, %2"]
    for r in range(16):
        pro += ["v_mul_u32_u24 v%d, %d, %%3" % (2 * r, 977 + 131 * r), "v_mov_b32 v%d, %d" % (2 * r + 1, r)]
    pro += ["s_waitcnt lgkmcnt(0)", "1:"]
    epi = ["s_sub_u32 s18, s18, 1", "s_cmp_lg_u32 s18, 0", "s_cbranch_scc1 1b", "v_mov_b32 v40, v0", "v_mov_b32 v41, v1"]
    for r in range(1, 16):
        epi.append("v_lshl_add_u64 v[40:41], v[%d:%d], 0, v[40:41]" % (2 * r, 2 * r + 1))
    epi.append("v_mov_b64 %0, v[40:41]")
    allL = pro + lines + epi
    return "\n".join('        "%s\\n\\t"' % x for x in allL)
clob = ", ".join(['"v%d"' % i for i in range(50)] + ['"s%d"' % i for i in range(12, 52)] + ['"vcc"', '"scc"', '"memory"'])
open(os.path.join(HERE, "bfly_asm_rate.hip"), "w").write(SRC.replace("@B0@", wrap(body(0))).replace("@B1@", wrap(body(1))).replace("@B2@", wrap(body2(2))).replace("@CLOB@", clob))
print("wrote bfly_asm_rate.hip")
