/* ora_keygen.h -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Restatement of the reference's key generation (mkrlwe/keygen.go, mkbfv/keygen.go) and CRS generation
 * (mkrlwe/params.go:16-61,77-99) with the RANDOM SAMPLES AS INPUTS: the reference draws secrets, errors and CRS from
 * lattigo's crypto PRNG (utils.NewPRNG, keygen.go:26, params.go:28,79), so there is no output of the Go code to be
 * bit-compatible with -- only the ring arithmetic applied to the samples is.  PARITY UNPINNED vs Go (see ora_ring.h).
 *   - secrets / errors: N small signed coefficients (what ring.TernarySampler / ring.GaussianSampler produce before
 *     they are written limb by limb as s or q_i - |s|)
 *   - CRS: uniform limbs expanded from a public seed with Philox4x32-10 (Salmon et al., SC'11) + mask-and-reject, the
 *     same sampling shape as lattigo's ring.UniformSampler; the reference then applies MFormLvl (params.go:56,97).
 * Only tests/ may load this.
 */
#ifndef ORA_KEYGEN_H
#define ORA_KEYGEN_H
#include "ora_mkbfv.h"
#ifdef __cplusplus
extern "C" {
#endif

void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* uniform in [0, q): candidates = 64-bit words of Philox(key = seed, ctr = {coeff, row, idx, block}) masked to
 * bitlen(q) bits, first one < q wins (block 0 holds candidates 0 and 1, block 1 candidates 2 and 3, ...). */
uint64_t ora_crs_sample(uint64_t seed, int32_t idx, uint32_t row, uint32_t coeff, uint64_t q);
/* params.go:47-59 / :91-98: CRS[idx] = betaMax PolyQP, uniform then MForm; row = digit*(nq+np) + limb */
void ora_crs_expand(const ora_ks* ks, uint64_t seed, int32_t idx, uint64_t* swk_out);

/* sampler write + RingQP.ExtendBasisSmallNormAndCenter (keygen.go:50-51,129-130): small signed -> PolyQP, coefficient domain */
void ora_small_to_qp(const ora_ks* ks, const int32_t* s, uint64_t* out);
/* genSecretKeyFromSampler keygen.go:44-55: NTT + MForm */
void ora_gen_secret_key(const ora_ks* ks, const int32_t* s, uint64_t* sk);
/* GenGaussianError keygen.go:124-134: NTT, no MForm */
void ora_gen_gaussian_error(const ora_ks* ks, const int32_t* e, uint64_t* out);
/* GenSwitchingKey keygen.go:270-327; e = [betaMax][N] */
void ora_gen_switching_key(const ora_ks* ks, const uint64_t* sk, const int32_t* e, uint64_t* swk);
/* GenPublicKey keygen.go:88-109; pk = [2][nq+np][N] */
void ora_gen_public_key(const ora_ks* ks, const uint64_t* sk, const int32_t* e, const uint64_t* crs_a, uint64_t* pk);
/* GenRelinearizationKey keygen.go:137-187; e = [3][betaMax][N] for b, d, v */
void ora_gen_relin_key(const ora_ks* ks, const uint64_t* sk, const uint64_t* r, const int32_t* e,
                       const uint64_t* crs_a, const uint64_t* crs_u, uint64_t* b, uint64_t* d, uint64_t* v);
/* ring.PermuteNTTIndex + PermuteNTTWithIndexLvl (lattigo ring/ring_automorphism.go) on a PolyQP */
void ora_permute_ntt_qp(const ora_ks* ks, uint64_t galEl, const uint64_t* in, uint64_t* out);
/* GenRotationKey keygen.go:190-229; galEl = GaloisElementForColumnRotationBy(rotidx) = 5^rotidx mod 2N */
void ora_gen_rotation_key(const ora_ks* ks, uint64_t galEl, const uint64_t* sk, const int32_t* e, const uint64_t* crs, uint64_t* rk);
/* GenConjugationKey keygen.go:240-268; crs = CRS[-2] */
void ora_gen_conjugation_key(const ora_ks* ks, const uint64_t* sk, const int32_t* e, const uint64_t* crs, uint64_t* ck);

/* mkbfv GenBFVSwitchingKey (mkbfv/keygen.go:91-162), one of its two loops: g = [betaMax][nq+np] residues of the
 * big-integer gadget scalars Gi (computed by the caller with big integers, :104-116 / :137-149) */
void ora_bfv_gen_switching_key(const ora_ks* ks, const uint64_t* sk, const uint64_t* g, const int32_t* e, uint64_t* swk);
/* mkbfv GenRelinearizationKey (mkbfv/keygen.go:24-88); e = [5][betaMax][N] for b1, b2, d1, d2, v */
void ora_bfv_gen_relin_key(const ora_ks* ks, const uint64_t* sk, const uint64_t* r, const uint64_t* g1, const uint64_t* g2,
                           const int32_t* e, const uint64_t* crs_a1, const uint64_t* crs_a2, const uint64_t* crs_u,
                           uint64_t* b1, uint64_t* b2, uint64_t* d1, uint64_t* d2, uint64_t* v);
#ifdef __cplusplus
}
#endif
#endif
