/* ora_ring.c -- CPU ORACLE (test infrastructure only; see ora_ring.h header).
 *
 * Restates lattigo v2.3.0 ring/{modular_reduction,ntt,ring,ring_operations,
 * ring_scaling,primes}.go as called by the reference at
 *   mkrlwe/keyswitch.go:29-30,58,88,106-115,161-206 ; mkrlwe/keyswitch_hoisted.go:28-37,84-143
 *   mkckks/evaluator.go:388 (DivRoundByLastModulusManyLvl)
 *   mkrlwe/keyswitch.go:267-296,316 (coefficient-domain Galois permutation).
 */
#include "ora_ring.h"
#include <stdlib.h>
#include <string.h>

uint64_t ora_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }

uint64_t ora_powmod(uint64_t x, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    x %= q;
    for (; e > 0; e >>= 1) {
        if (e & 1) r = ora_mulmod(r, x, q);
        x = ora_mulmod(x, x, q);
    }
    return r;
}

/* ring.MRedParams: q^-1 mod 2^64 by repeated squaring (q odd). */
uint64_t ora_mredparams(uint64_t q) {
    uint64_t qinv = 1, x = q;
    for (int i = 0; i < 63; ++i) { qinv *= x; x *= x; }
    return qinv;
}

/* ring.BRedParams: floor(2^128/q) split (hi, lo). */
void ora_bredparams(uint64_t q, uint64_t* uhi, uint64_t* ulo) {
    /* 2^128 / q  = ((2^128-1) / q) unless q | 2^128 (impossible for odd q>1) */
    u128 all = ~(u128)0;
    u128 u = all / q;
    *uhi = (uint64_t)(u >> 64);
    *ulo = (uint64_t)u;
}

static int factor_distinct(uint64_t n, uint64_t* f) {
    int k = 0;
    for (uint64_t p = 2; p * p <= n && p < (1ull << 32); p += (p == 2 ? 1 : 2)) {
        if (n % p == 0) { f[k++] = p; while (n % p == 0) n /= p; }
    }
    if (n > 1) f[k++] = n;
    return k;
}

/* lattigo ring/primes.go primitiveRoot: g starts at 2 and is incremented BEFORE the
 * first test, so the search begins at g = 3. */
uint64_t ora_primitive_root(uint64_t q) {
    uint64_t f[64];
    int nf = factor_distinct(q - 1, f);
    uint64_t g = 2;
    for (;;) {
        g++;
        int ok = 1;
        for (int i = 0; i < nf; ++i)
            if (ora_powmod(g, (q - 1) / f[i], q) == 1) { ok = 0; break; }
        if (ok) return g;
    }
}

static uint64_t bitrev(uint64_t x, int bits) {
    uint64_t r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

ora_ring* ora_ring_new(int logN, const uint64_t* moduli, int nmod, const uint64_t* psi_or_null) {
    if (nmod > ORA_MAXMOD || logN < 1 || logN > 17) return NULL;
    ora_ring* r = (ora_ring*)calloc(1, sizeof(ora_ring));
    r->logN = logN; r->N = 1 << logN; r->nmod = nmod;
    const uint64_t N = (uint64_t)r->N;
    for (int i = 0; i < nmod; ++i) {
        uint64_t q = moduli[i];
        r->mod[i] = q;
        r->qinv[i] = ora_mredparams(q);
        ora_bredparams(q, &r->uhi[i], &r->ulo[i]);
        r->ninv[i] = ora_mform(ora_powmod(N, q - 2, q), q, r->uhi[i], r->ulo[i]);
        uint64_t psi;
        if (psi_or_null) psi = psi_or_null[i];
        else {
            uint64_t g = ora_primitive_root(q);
            psi = ora_powmod(g, (q - 1) / (2 * N), q);
        }
        r->psi_plain[i] = psi;
        uint64_t psiinv = ora_powmod(psi, q - 2, q);
        uint64_t psiM = ora_mform(psi, q, r->uhi[i], r->ulo[i]);
        uint64_t psiinvM = ora_mform(psiinv, q, r->uhi[i], r->ulo[i]);
        r->psi[i] = (uint64_t*)malloc(sizeof(uint64_t) * N);
        r->psiinv[i] = (uint64_t*)malloc(sizeof(uint64_t) * N);
        r->psi[i][0] = ora_mform(1, q, r->uhi[i], r->ulo[i]);
        r->psiinv[i][0] = r->psi[i][0];
        for (uint64_t j = 1; j < N; ++j) {
            uint64_t prev = bitrev(j - 1, logN), next = bitrev(j, logN);
            r->psi[i][next] = ora_mred(r->psi[i][prev], psiM, q, r->qinv[i]);
            r->psiinv[i][next] = ora_mred(r->psiinv[i][prev], psiinvM, q, r->qinv[i]);
        }
    }
    /* RescaleParams[L-1][i] = MForm(qL^-1 mod qi), i < L */
    for (int L = 1; L < nmod; ++L) {
        r->rescale[L - 1] = (uint64_t*)malloc(sizeof(uint64_t) * L);
        for (int i = 0; i < L; ++i) {
            uint64_t qi = r->mod[i];
            uint64_t inv = ora_powmod(r->mod[L] % qi, qi - 2, qi);
            r->rescale[L - 1][i] = ora_mform(inv, qi, r->uhi[i], r->ulo[i]);
        }
    }
    return r;
}

void ora_ring_free(ora_ring* r) {
    if (!r) return;
    for (int i = 0; i < r->nmod; ++i) { free(r->psi[i]); free(r->psiinv[i]); }
    for (int i = 0; i + 1 < r->nmod; ++i) free(r->rescale[i]);
    free(r);
}
int ora_ring_n(const ora_ring* r) { return r->N; }
uint64_t ora_ring_psi(const ora_ring* r, int i) { return r->psi_plain[i]; }
uint64_t ora_ring_qinv(const ora_ring* r, int i) { return r->qinv[i]; }
const uint64_t* ora_ring_psi_table(const ora_ring* r, int i, int inverse) { return inverse ? r->psiinv[i] : r->psi[i]; }

/* ---- NTT: Cooley-Tukey, natural in -> bit-reversed out, lazy [0,4q)/[0,8q) inside,
 * final BRedAdd => canonical whatever the input magnitude (< 2^61). ---- */
void ora_ntt(const ora_ring* r, int i, const uint64_t* in, uint64_t* out) {
    const int N = r->N;
    const uint64_t q = r->mod[i], qinv = r->qinv[i], twoq = q << 1, fourq = q << 2;
    const uint64_t* psi = r->psi[i];
    int t = N >> 1;
    uint64_t F = psi[1];
    for (int j = 0; j < t; ++j) {
        uint64_t V = ora_mred_lazy(in[j + t], F, q, qinv);
        uint64_t U = in[j];
        out[j] = U + V;
        out[j + t] = U + twoq - V;
    }
    for (int m = 2; m < N; m <<= 1) {
        int lenm = 0; for (int x = m; x; x >>= 1) lenm++;   /* bits.Len64(m) */
        int reduce = (lenm & 1) == 1;
        t >>= 1;
        for (int g = 0; g < m; ++g) {
            int j1 = (g * t) << 1;
            F = psi[m + g];
            for (int j = j1; j < j1 + t; ++j) {
                uint64_t U = out[j];
                if (reduce && U >= fourq) U -= fourq;
                uint64_t V = ora_mred_lazy(out[j + t], F, q, qinv);
                out[j] = U + V;
                out[j + t] = U + twoq - V;
            }
        }
    }
    for (int j = 0; j < N; ++j) out[j] = ora_bred_add(out[j], q, r->uhi[i]);
}

static void intt_core(const ora_ring* r, int i, const uint64_t* in, uint64_t* out) {
    const int N = r->N;
    const uint64_t q = r->mod[i], qinv = r->qinv[i], twoq = q << 1, fourq = q << 2;
    const uint64_t* psi = r->psiinv[i];
    /* first layer: t = 1, h = N/2 */
    int h = N >> 1;
    for (int g = 0; g < h; ++g) {
        uint64_t U = in[2 * g], V = in[2 * g + 1];
        uint64_t X = U + V; if (X >= twoq) X -= twoq;
        out[2 * g] = X;
        out[2 * g + 1] = ora_mred_lazy(U + fourq - V, psi[h + g], q, qinv);
    }
    int t = 2;
    for (int m = N >> 1; m > 1; m >>= 1) {
        h = m >> 1;
        for (int g = 0; g < h; ++g) {
            int j1 = 2 * g * t;
            uint64_t F = psi[h + g];
            for (int j = j1; j < j1 + t; ++j) {
                uint64_t U = out[j], V = out[j + t];
                uint64_t X = U + V; if (X >= twoq) X -= twoq;
                out[j] = X;
                out[j + t] = ora_mred_lazy(U + fourq - V, F, q, qinv);
            }
        }
        t <<= 1;
    }
}

void ora_intt(const ora_ring* r, int i, const uint64_t* in, uint64_t* out) {
    intt_core(r, i, in, out);
    const uint64_t q = r->mod[i], qinv = r->qinv[i], ninv = r->ninv[i];
    for (int j = 0; j < r->N; ++j) out[j] = ora_mred(out[j], ninv, q, qinv);
}
void ora_intt_lazy(const ora_ring* r, int i, const uint64_t* in, uint64_t* out) {
    intt_core(r, i, in, out);
    const uint64_t q = r->mod[i], qinv = r->qinv[i], ninv = r->ninv[i];
    for (int j = 0; j < r->N; ++j) out[j] = ora_mred_lazy(out[j], ninv, q, qinv);
}

#define LIMB_LOOP(expr) do { const uint64_t q = r->mod[i], qinv = r->qinv[i]; (void)qinv; \
    for (int j = 0; j < r->N; ++j) { expr; } } while (0)

void ora_limb_mform(const ora_ring* r, int i, const uint64_t* a, uint64_t* z) { LIMB_LOOP(z[j] = ora_mform(a[j], q, r->uhi[i], r->ulo[i])); }
void ora_limb_invmform(const ora_ring* r, int i, const uint64_t* a, uint64_t* z) { LIMB_LOOP(z[j] = ora_invmform(a[j], q, qinv)); }
void ora_limb_mul(const ora_ring* r, int i, const uint64_t* a, const uint64_t* b, uint64_t* z) { LIMB_LOOP(z[j] = ora_mred(a[j], b[j], q, qinv)); }
void ora_limb_mul_add(const ora_ring* r, int i, const uint64_t* a, const uint64_t* b, uint64_t* z) { LIMB_LOOP(z[j] = ora_cred(z[j] + ora_mred(a[j], b[j], q, qinv), q)); }
void ora_limb_mul_sub(const ora_ring* r, int i, const uint64_t* a, const uint64_t* b, uint64_t* z) { LIMB_LOOP(z[j] = ora_cred(z[j] + (q - ora_mred(a[j], b[j], q, qinv)), q)); }
void ora_limb_add(const ora_ring* r, int i, const uint64_t* a, const uint64_t* b, uint64_t* z) { LIMB_LOOP(z[j] = ora_cred(a[j] + b[j], q)); }
void ora_limb_sub(const ora_ring* r, int i, const uint64_t* a, const uint64_t* b, uint64_t* z) { LIMB_LOOP(z[j] = ora_cred(a[j] + q - b[j], q)); }
void ora_limb_neg(const ora_ring* r, int i, const uint64_t* a, uint64_t* z) { LIMB_LOOP(z[j] = q - a[j]); }
void ora_limb_reduce(const ora_ring* r, int i, const uint64_t* a, uint64_t* z) { LIMB_LOOP(z[j] = ora_bred_add(a[j], q, r->uhi[i])); }
void ora_limb_mul_scalar(const ora_ring* r, int i, const uint64_t* a, uint64_t s, uint64_t* z) {
    uint64_t sm = ora_mform(ora_bred_add(s, r->mod[i], r->uhi[i]), r->mod[i], r->uhi[i], r->ulo[i]);
    LIMB_LOOP(z[j] = ora_mred(a[j], sm, q, qinv));
}

/* Coefficient-domain automorphism X -> X^galEl with sign (mkrlwe/keyswitch.go:267-296);
 * note 0 with a sign flip is written as q (non-canonical), exactly like the reference. */
void ora_permute(const ora_ring* r, int level, uint64_t galEl, const uint64_t* in, uint64_t* out) {
    const uint64_t N = (uint64_t)r->N, mask = N - 1;
    const int logN = r->logN;
    for (uint64_t k = 0; k < N; ++k) {
        uint64_t raw = k * galEl, idx = raw & mask, sgn = (raw >> logN) & 1;
        for (int l = 0; l <= level; ++l) {
            uint64_t v = in[(size_t)l * N + k];
            out[(size_t)l * N + idx] = sgn ? r->mod[l] - v : v;
        }
    }
}

/* lattigo ring_scaling.go DivRoundByLastModulusLvl (coefficient domain), App. A.6. */
static void div_round_last(const ora_ring* r, int level, uint64_t* p0, uint64_t* p1) {
    const size_t N = (size_t)r->N;
    const uint64_t qL = r->mod[level], h = (qL - 1) >> 1;
    uint64_t* last = p0 + (size_t)level * N;
    for (size_t j = 0; j < N; ++j) last[j] = ora_cred(last[j] + h, qL);
    for (int i = 0; i < level; ++i) {
        const uint64_t qi = r->mod[i], qinv = r->qinv[i], twoqi = qi << 1;
        const uint64_t hneg = qi - ora_bred_add(h, qi, r->uhi[i]);
        const uint64_t rp = qi - r->rescale[level - 1][i];
        const uint64_t* x = p0 + (size_t)i * N;
        uint64_t* z = p1 + (size_t)i * N;
        for (size_t j = 0; j < N; ++j) z[j] = ora_mred(last[j] + hneg + twoqi - x[j], rp, qi, qinv);
    }
}

void ora_div_round_last_many(const ora_ring* r, int level, int nb, uint64_t* in, uint64_t* out) {
    const size_t N = (size_t)r->N;
    if (nb == 0) { if (in != out) memcpy(out, in, sizeof(uint64_t) * N * (size_t)(level + 1)); return; }
    if (nb == 1) { div_round_last(r, level, in, out); return; }
    uint64_t* pool = (uint64_t*)malloc(sizeof(uint64_t) * N * (size_t)(level + 1));
    div_round_last(r, level, in, pool);
    for (int k = 1; k < nb; ++k) {
        if (k == nb - 1) div_round_last(r, level - k, pool, out);
        else div_round_last(r, level - k, pool, pool);
    }
    free(pool);
}
