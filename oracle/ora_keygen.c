/* ora_keygen.c -- CPU ORACLE (test infrastructure): key generation / CRS expansion, see ora_keygen.h */
#include <stdlib.h>
#include <string.h>
#include "ora_keygen.h"

/* ---- Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) */
void ora_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

uint64_t ora_crs_sample(uint64_t seed, int32_t idx, uint32_t row, uint32_t coeff, uint64_t q) {
    int bits = 0;
    while (bits < 64 && (q >> bits)) ++bits;
    const uint64_t mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (uint32_t block = 0;; ++block) {
        const uint32_t ctr[4] = {coeff, row, (uint32_t)idx, block};
        uint32_t o[4];
        ora_philox4x32_10(ctr, key, o);
        const uint64_t c0 = (((uint64_t)o[1] << 32) | o[0]) & mask, c1 = (((uint64_t)o[3] << 32) | o[2]) & mask;
        if (c0 < q) return c0;
        if (c1 < q) return c1;
    }
}

static const ora_ring* ring_of(const ora_ks* ks, int j, int* i) {
    if (j < ks->nq) { *i = j; return ks->rq; }
    *i = j - ks->nq; return ks->rp;
}

void ora_crs_expand(const ora_ks* ks, uint64_t seed, int32_t idx, uint64_t* out) {
    const int m = ks->nq + ks->np; const size_t N = (size_t)ks->N;
    for (int d = 0; d < ks->beta_max; ++d)
        for (int j = 0; j < m; ++j) {
            int i; const ora_ring* r = ring_of(ks, j, &i);
            uint64_t* z = out + ((size_t)d * m + j) * N;
            for (size_t w = 0; w < N; ++w)                                            /* uniformSampler.Read, params.go:54-55 */
                z[w] = ora_crs_sample(seed, idx, (uint32_t)(d * m + j), (uint32_t)w, r->mod[i]);
            ora_limb_mform(r, i, z, z);                                               /* MFormLvl, params.go:56 */
        }
}

void ora_small_to_qp(const ora_ks* ks, const int32_t* s, uint64_t* out) {
    const size_t N = (size_t)ks->N;
    for (int j = 0; j < ks->nq; ++j)                       /* sampler.Read: s or q_j - |s| on every Q limb */
        for (size_t w = 0; w < N; ++w) out[(size_t)j * N + w] = s[w] < 0 ? ks->rq->mod[j] - (uint64_t)(-(int64_t)s[w]) : (uint64_t)s[w];
    const uint64_t Q0 = ks->rq->mod[0], half = Q0 >> 1;    /* ExtendBasisSmallNormAndCenter: sign and magnitude from limb 0 */
    for (size_t w = 0; w < N; ++w) {
        uint64_t c = out[w]; int neg = 0;
        if (c > half) { c = Q0 - c; neg = 1; }
        for (int i = 0; i < ks->np; ++i) out[(size_t)(ks->nq + i) * N + w] = neg ? ks->rp->mod[i] - c : c;
    }
}

static void qp_ntt(const ora_ks* ks, uint64_t* p) {
    const size_t N = (size_t)ks->N;
    for (int j = 0; j < ks->nq + ks->np; ++j) { int i; const ora_ring* r = ring_of(ks, j, &i); ora_ntt(r, i, p + j * N, p + j * N); }
}
static void qp_mform(const ora_ks* ks, uint64_t* p) {
    const size_t N = (size_t)ks->N;
    for (int j = 0; j < ks->nq + ks->np; ++j) { int i; const ora_ring* r = ring_of(ks, j, &i); ora_limb_mform(r, i, p + j * N, p + j * N); }
}

void ora_gen_secret_key(const ora_ks* ks, const int32_t* s, uint64_t* sk) { ora_small_to_qp(ks, s, sk); qp_ntt(ks, sk); qp_mform(ks, sk); }
void ora_gen_gaussian_error(const ora_ks* ks, const int32_t* e, uint64_t* out) { ora_small_to_qp(ks, e, out); qp_ntt(ks, out); }

void ora_gen_switching_key(const ora_ks* ks, const uint64_t* sk, const int32_t* e, uint64_t* swk) {
    const int m = ks->nq + ks->np; const size_t N = (size_t)ks->N;
    uint64_t* psk = (uint64_t*)malloc((size_t)ks->nq * N * 8);
    for (int j = 0; j < ks->nq; ++j) {                      /* MulScalarBigintLvl(levelQ, sk.Q, P, poolQ), keygen.go:288 */
        uint64_t pm = 1;
        for (int i = 0; i < ks->np; ++i) pm = ora_mulmod(pm, ks->rp->mod[i] % ks->rq->mod[j], ks->rq->mod[j]);
        ora_limb_mul_scalar(ks->rq, j, sk + (size_t)j * N, pm, psk + (size_t)j * N);
    }
    for (int d = 0; d < ks->beta_max; ++d) {
        uint64_t* z = swk + (size_t)d * m * N;
        ora_gen_gaussian_error(ks, e + (size_t)d * N, z);  /* :296-298 */
        qp_mform(ks, z);                                   /* :299 */
        for (int j = 0; j < ks->alpha; ++j) {              /* :308-323 */
            const int index = d * ks->alpha + j;
            if (index >= ks->nq) break;
            ora_limb_add(ks->rq, index, z + (size_t)index * N, psk + (size_t)index * N, z + (size_t)index * N);
        }
    }
    free(psk);
}

static void qp_mul_sub(const ora_ks* ks, const uint64_t* x, const uint64_t* y, uint64_t* z) {     /* MulCoeffsMontgomeryAndSubLvl */
    const size_t N = (size_t)ks->N;
    for (int j = 0; j < ks->nq + ks->np; ++j) { int i; const ora_ring* r = ring_of(ks, j, &i); ora_limb_mul_sub(r, i, x + j * N, y + j * N, z + j * N); }
}

void ora_gen_public_key(const ora_ks* ks, const uint64_t* sk, const int32_t* e, const uint64_t* crs_a, uint64_t* pk) {
    const size_t pw = (size_t)(ks->nq + ks->np) * ks->N;
    ora_gen_gaussian_error(ks, e, pk);                       /* :98-100 */
    memcpy(pk + pw, crs_a, pw * 8);                          /* :103-104: CRS[0].Value[0] */
    qp_mul_sub(ks, sk, pk + pw, pk);                         /* :106 */
}

void ora_gen_relin_key(const ora_ks* ks, const uint64_t* sk, const uint64_t* r, const int32_t* e,
                       const uint64_t* a, const uint64_t* u, uint64_t* b, uint64_t* d, uint64_t* v) {
    const int m = ks->nq + ks->np, beta = ks->beta_max; const size_t N = (size_t)ks->N, pw = (size_t)m * N;
    uint64_t* tmp = (uint64_t*)malloc(pw * 8);
    for (int i = 0; i < beta; ++i) {                         /* b = -s a + e, keygen.go:163-169 */
        uint64_t* bi = b + i * pw;
        ora_gen_gaussian_error(ks, e + (size_t)i * N, tmp);
        for (int j = 0; j < m; ++j) {
            int l; const ora_ring* rg = ring_of(ks, j, &l);
            ora_limb_mul(rg, l, a + i * pw + j * N, sk + j * N, bi + j * N);
            ora_limb_invmform(rg, l, bi + j * N, bi + j * N);
            ora_limb_sub(rg, l, tmp + j * N, bi + j * N, bi + j * N);
            ora_limb_mform(rg, l, bi + j * N, bi + j * N);
        }
    }
    ora_gen_switching_key(ks, sk, e + (size_t)beta * N, d);  /* d = -r a + s g + e, :172-176 */
    for (int i = 0; i < beta; ++i) qp_mul_sub(ks, a + i * pw, r, d + i * pw);
    ora_gen_switching_key(ks, r, e + (size_t)2 * beta * N, v);   /* v = -s u - r g + e, :179-185 */
    for (int i = 0; i < beta; ++i)
        for (int j = 0; j < m; ++j) {
            int l; const ora_ring* rg = ring_of(ks, j, &l);
            uint64_t* z = v + i * pw + j * N;
            ora_limb_mul_add(rg, l, u + i * pw + j * N, sk + j * N, z);
            ora_limb_neg(rg, l, z, z);
        }
    free(tmp);
}

static uint64_t bitrev64(uint64_t x, int bits) { uint64_t r = 0; for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i); return r; }

void ora_permute_ntt_qp(const ora_ks* ks, uint64_t galEl, const uint64_t* in, uint64_t* out) {
    const uint64_t N = (uint64_t)ks->N, mask = 2 * N - 1; const int logN = ks->rq->logN;
    for (uint64_t i = 0; i < N; ++i) {                       /* PermuteNTTIndex */
        const uint64_t t1 = 2 * bitrev64(i, logN) + 1, t2 = ((galEl * t1 & mask) - 1) >> 1, idx = bitrev64(t2, logN);
        for (int j = 0; j < ks->nq + ks->np; ++j) out[(size_t)j * N + i] = in[(size_t)j * N + idx];   /* PermuteNTTWithIndexLvl */
    }
}

static uint64_t powmod_u64(uint64_t b, uint64_t e, uint64_t mod) { uint64_t r = 1; b %= mod; while (e) { if (e & 1) r = (uint64_t)((u128)r * b % mod); b = (uint64_t)((u128)b * b % mod); e >>= 1; } return r; }

void ora_gen_rotation_key(const ora_ks* ks, uint64_t galEl, const uint64_t* sk, const int32_t* e, const uint64_t* crs, uint64_t* rk) {
    const size_t pw = (size_t)(ks->nq + ks->np) * ks->N; const uint64_t N2 = 2 * (uint64_t)ks->N;
    uint64_t* sk_out = (uint64_t*)malloc(pw * 8);
    ora_permute_ntt_qp(ks, powmod_u64(galEl, N2 - 1, N2), sk, sk_out);          /* InverseGaloisElement, keygen.go:213-217 */
    ora_gen_switching_key(ks, sk, e, rk);                                       /* :221 */
    for (int i = 0; i < ks->beta_max; ++i) qp_mul_sub(ks, crs + i * pw, sk_out, rk + i * pw);   /* :225-227 */
    free(sk_out);
}

void ora_gen_conjugation_key(const ora_ks* ks, const uint64_t* sk, const int32_t* e, const uint64_t* crs, uint64_t* ck) {
    const size_t pw = (size_t)(ks->nq + ks->np) * ks->N;
    uint64_t* sk_out = (uint64_t*)malloc(pw * 8);
    ora_permute_ntt_qp(ks, 2 * (uint64_t)ks->N - 1, sk, sk_out);                /* GaloisElementForRowRotation, :252-255 */
    ora_gen_switching_key(ks, sk_out, e, ck);                                   /* :259 */
    for (int i = 0; i < ks->beta_max; ++i) qp_mul_sub(ks, crs + i * pw, sk, ck + i * pw);       /* :263-265 */
    free(sk_out);
}

void ora_bfv_gen_switching_key(const ora_ks* ks, const uint64_t* sk, const uint64_t* g, const int32_t* e, uint64_t* swk) {
    const int m = ks->nq + ks->np; const size_t N = (size_t)ks->N, pw = (size_t)m * N;
    uint64_t* err = (uint64_t*)malloc(pw * 8);
    for (int d = 0; d < ks->beta_max; ++d) {
        uint64_t* z = swk + d * pw;
        ora_gen_gaussian_error(ks, e + (size_t)d * N, err);                     /* mkbfv/keygen.go:122-123 */
        for (int j = 0; j < m; ++j) {
            int l; const ora_ring* rg = ring_of(ks, j, &l);
            ora_limb_invmform(rg, l, sk + j * N, z + j * N);                    /* :118 */
            ora_limb_mul_scalar(rg, l, z + j * N, g[(size_t)d * m + j], z + j * N);   /* MulScalarBigint :119-120 */
            ora_limb_add(rg, l, z + j * N, err + j * N, z + j * N);             /* :125 */
            ora_limb_mform(rg, l, z + j * N, z + j * N);                        /* :126 */
        }
    }
    free(err);
}

void ora_bfv_gen_relin_key(const ora_ks* ks, const uint64_t* sk, const uint64_t* r, const uint64_t* g1, const uint64_t* g2,
                           const int32_t* e, const uint64_t* a1, const uint64_t* a2, const uint64_t* u,
                           uint64_t* b1, uint64_t* b2, uint64_t* d1, uint64_t* d2, uint64_t* v) {
    const int m = ks->nq + ks->np, beta = ks->beta_max; const size_t N = (size_t)ks->N, pw = (size_t)m * N;
    uint64_t* tmp = (uint64_t*)malloc(pw * 8);
    for (int which = 0; which < 2; ++which) {                /* b1, b2 = -s a + e, mkbfv/keygen.go:52-64 */
        const uint64_t* a = which ? a2 : a1; uint64_t* b = which ? b2 : b1;
        for (int i = 0; i < beta; ++i) {
            ora_gen_gaussian_error(ks, e + ((size_t)which * beta + i) * N, tmp);
            for (int j = 0; j < m; ++j) {
                int l; const ora_ring* rg = ring_of(ks, j, &l);
                uint64_t* z = b + i * pw + j * N;
                ora_limb_mul(rg, l, a + i * pw + j * N, sk + j * N, z);
                ora_limb_invmform(rg, l, z, z);
                ora_limb_sub(rg, l, tmp + j * N, z, z);
                ora_limb_mform(rg, l, z, z);
            }
        }
    }
    ora_bfv_gen_switching_key(ks, sk, g1, e + (size_t)2 * beta * N, d1);        /* :70-74 */
    ora_bfv_gen_switching_key(ks, sk, g2, e + (size_t)3 * beta * N, d2);
    for (int i = 0; i < beta; ++i) { qp_mul_sub(ks, a1 + i * pw, r, d1 + i * pw); qp_mul_sub(ks, a2 + i * pw, r, d2 + i * pw); }
    ora_gen_switching_key(ks, r, e + (size_t)4 * beta * N, v);                  /* :79-87 */
    for (int i = 0; i < beta; ++i)
        for (int j = 0; j < m; ++j) {
            int l; const ora_ring* rg = ring_of(ks, j, &l);
            uint64_t* z = v + i * pw + j * N;
            ora_limb_mul_add(rg, l, u + i * pw + j * N, sk + j * N, z);
            ora_limb_neg(rg, l, z, z);
        }
    free(tmp);
}
