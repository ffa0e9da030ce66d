// keygen_kernels.h -- elementwise kernels of key generation and CRS expansion:
// mkrlwe/keygen.go, mkbfv/keygen.go, mkrlwe/params.go:16-61,77-99.  All HBM-streaming; the NTTs in between are the
// batched kernels of ntt_kernels.hip.
#pragma once
#include "modarith.h"

namespace mkhe {

// sampler write + ExtendBasisSmallNormAndCenter (keygen.go:50-51,129-130): dst[c][j][n] = s >= 0 ? s : q_j - |s|
void launch_small_expand(u64* dst, const i32* small, const Mod* mods, int count, int limbs, int N, hipStream_t st);

// One pass over a switching key [beta][mtot][N] that holds E_i = NTT(e_i) (canonical, not Montgomery) on entry:
//   val = mform_e ? MForm(E) : E
//   g[i][j] != 0 :  val = CRed(val + MRed(skA[j], g[i][j]))              gadget term, g = MForm(scalar)
//   crs          :  val = CRed(val +- MRed(crs[i][j], skB[j]))            sign = +1 / -1
//   neg          :  val = q - val                                          ring.Neg (0 -> q, like lattigo)
// Covers GenSwitchingKey (keygen.go:270-327), the b / d / v loops of GenRelinearizationKey (:163-185), GenRotationKey
// (:221-227), GenConjugationKey (:259-265), GenPublicKey (:98-106), mkbfv GenBFVSwitchingKey / GenRelinearizationKey.
struct KeygenArgs {
    u64* out;               // [beta][mtot][N], E on entry
    const u64* skA;         // [mtot][N] or null
    const u64* g;           // device [beta][mtot] or null
    const u64* crs;         // [beta][mtot][N] or null
    const u64* skB;         // [mtot][N]
    const Mod* mods;
    int beta, mtot, N, sign, neg, mform_e;
};
void launch_keygen_combine(const KeygenArgs& a, hipStream_t st);

// ring.PermuteNTTIndex + PermuteNTTWithIndexLvl (lattigo ring_automorphism.go; keygen.go:215-217,253-255):
// dst[j][i] = src[j][bitrev(((galEl * (2 bitrev(i) + 1) mod 2N) - 1) / 2)]
void launch_permute_ntt(u64* dst, const u64* src, int limbs, int logN, u64 galEl, hipStream_t st);

// CRS expansion (params.go:47-59,91-98): uniform limbs from Philox4x32-10 keyed by a public seed (counter = coefficient,
// digit*mtot + limb, idx, block; two masked 64-bit candidates per block, first one < q wins), then MForm.
void launch_crs_expand(u64* out, const Mod* mods, u64 seed, i32 idx, int beta, int mtot, int N, hipStream_t st);

}  // namespace mkhe
