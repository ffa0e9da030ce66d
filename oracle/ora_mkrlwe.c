/* ora_mkrlwe.c -- CPU ORACLE (test infrastructure only; see ora_mkrlwe.h header). */
#include "ora_mkrlwe.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define NEWA(T, n) ((T*)calloc((size_t)(n), sizeof(T)))

/* ------------------------------------------------------------------------
 * basisextenderparameters  (mkrlwe/basis_extension.go:83-153).
 * All three tables are outputs of MRed/MForm/CRed chains, i.e. canonical, so
 * they are computed here from their closed forms:
 *   qoverqiinvqi[i]   = ((Q/q_i)^-1 mod q_i) * 2^64 mod q_i
 *   qoverqimodp[j][i] = ((Q/q_i) mod p_j)    * 2^64 mod p_j
 *   vtimesqmodp[j][v] = v * (p_j - Q mod p_j) mod p_j ,  v = 0..ns
 * ---------------------------------------------------------------------- */
static uint64_t to_mont(uint64_t a, uint64_t q) { return (uint64_t)((((u128)a) << 64) % q); }

static ora_modup modup_params(const uint64_t* Q, int ns, const uint64_t* P, int nt) {
    ora_modup m;
    m.ns = ns; m.nt = nt;
    m.qoverqiinvqi = NEWA(uint64_t, ns);
    m.qoverqimodp = NEWA(uint64_t, (size_t)nt * ns);
    m.vtimesqmodp = NEWA(uint64_t, (size_t)nt * (ns + 1));
    for (int i = 0; i < ns; ++i) {
        uint64_t qi = Q[i], star = 1 % qi;
        for (int j = 0; j < ns; ++j) if (j != i) star = ora_mulmod(star, Q[j] % qi, qi);
        m.qoverqiinvqi[i] = to_mont(ora_powmod(star, qi - 2, qi), qi);
        for (int j = 0; j < nt; ++j) {
            uint64_t pj = P[j], s = 1 % pj;
            for (int u = 0; u < ns; ++u) if (u != i) s = ora_mulmod(s, Q[u] % pj, pj);
            m.qoverqimodp[(size_t)j * ns + i] = to_mont(s, pj);
        }
    }
    for (int j = 0; j < nt; ++j) {
        uint64_t pj = P[j], qm = 1 % pj;
        for (int i = 0; i < ns; ++i) qm = ora_mulmod(qm, Q[i] % pj, pj);
        uint64_t v = pj - qm;
        uint64_t* t = m.vtimesqmodp + (size_t)j * (ns + 1);
        t[0] = 0;
        for (int i = 1; i <= ns; ++i) t[i] = ora_cred(t[i - 1] + v, pj);
    }
    return m;
}
static void modup_free(ora_modup* m) { free(m->qoverqiinvqi); free(m->qoverqimodp); free(m->vtimesqmodp); }

/* genModDownParams(ringQ, ringP)[j][i] = ((p_0..p_j)^-1 mod q_i) * 2^64 mod q_i  (basis_extension.go:34-54) */
static uint64_t* moddown_params(const ora_ring* rq, const ora_ring* rp) {
    uint64_t* t = NEWA(uint64_t, (size_t)rp->nmod * rq->nmod);
    for (int i = 0; i < rq->nmod; ++i) {
        uint64_t qi = rq->mod[i], acc = 1 % qi;
        for (int j = 0; j < rp->nmod; ++j) {
            acc = ora_mulmod(acc, ora_powmod(rp->mod[j] % qi, qi - 2, qi), qi);
            t[(size_t)j * rq->nmod + i] = to_mont(acc, qi);
        }
    }
    return t;
}

/* reconstructRNS + multSum  (basis_extension.go:537-646), one coefficient at a time.
 * The float64 correction index and the un-reduced multSum representative
 *   rhi - hhi + p_j + vtimesqmodp[v]
 * are reproduced literally. */
static inline uint64_t reconstruct(int ns, const uint64_t* const* src, size_t x, const uint64_t* Q,
                                   const uint64_t* Qinv, const uint64_t* qoverqiinvqi, uint64_t* y) {
    double vi = 0.0;
    for (int i = 0; i < ns; ++i) {
        y[i] = ora_mred(src[i][x], qoverqiinvqi[i], Q[i], Qinv[i]);
        vi += (double)y[i] / (double)Q[i];
    }
    return (uint64_t)vi;
}
static inline uint64_t multsum(int ns, const uint64_t* y, uint64_t v, uint64_t pj, uint64_t pinv,
                               const uint64_t* vtimesqmodp, const uint64_t* qoverqimodp) {
    uint64_t rlo = 0, rhi = 0;
    for (int i = 0; i < ns; ++i) {
        u128 m = (u128)y[i] * qoverqimodp[i];
        uint64_t mlo = (uint64_t)m, mhi = (uint64_t)(m >> 64);
        uint64_t s = rlo + mlo, c = s < rlo;
        rlo = s; rhi += mhi + c;
    }
    uint64_t hhi = ora_mulhi(rlo * pinv, pj);
    return rhi - hhi + pj + vtimesqmodp[v];
}

/* modUpExact (basis_extension.go:337-357): src limbs (ring rs, first ns moduli) -> dst limbs (ring rd, first nt). */
static void modup_exact(const ora_ring* rs, int ns, const uint64_t* src, const ora_ring* rd, int nt,
                        uint64_t* dst, const ora_modup* mp) {
    const size_t N = (size_t)rs->N;
    const uint64_t* sp[32]; uint64_t y[32];
    for (int i = 0; i < ns; ++i) sp[i] = src + (size_t)i * N;
    for (size_t x = 0; x < N; ++x) {
        uint64_t v = reconstruct(ns, sp, x, rs->mod, rs->qinv, mp->qoverqiinvqi, y);
        for (int j = 0; j < nt; ++j)
            dst[(size_t)j * N + x] = multsum(ns, y, v, rd->mod[j], rd->qinv[j],
                                             mp->vtimesqmodp + (size_t)j * (mp->ns + 1),
                                             mp->qoverqimodp + (size_t)j * mp->ns);
    }
}

ora_fbe* ora_fbe_new(const ora_ring* ra, const ora_ring* rb) {
    ora_fbe* f = NEWA(ora_fbe, 1);
    f->ra = ra; f->rb = rb;
    f->a2b = NEWA(ora_modup, ra->nmod);
    for (int i = 0; i < ra->nmod; ++i) f->a2b[i] = modup_params(ra->mod, i + 1, rb->mod, rb->nmod);
    f->b2a = NEWA(ora_modup, rb->nmod);
    for (int i = 0; i < rb->nmod; ++i) f->b2a[i] = modup_params(rb->mod, i + 1, ra->mod, ra->nmod);
    f->down_b2a = moddown_params(ra, rb);
    f->down_a2b = moddown_params(rb, ra);
    return f;
}
void ora_fbe_free(ora_fbe* f) {
    if (!f) return;
    for (int i = 0; i < f->ra->nmod; ++i) modup_free(&f->a2b[i]);
    for (int i = 0; i < f->rb->nmod; ++i) modup_free(&f->b2a[i]);
    free(f->a2b); free(f->b2a); free(f->down_b2a); free(f->down_a2b); free(f);
}
/* ModUpQtoP / ModUpPtoQ (basis_extension.go:177-186) */
void ora_fbe_modup_a2b(const ora_fbe* f, int levelA, int levelB, const uint64_t* pa, uint64_t* pb) {
    modup_exact(f->ra, levelA + 1, pa, f->rb, levelB + 1, pb, &f->a2b[levelA]);
}
void ora_fbe_modup_b2a(const ora_fbe* f, int levelB, int levelA, const uint64_t* pb, uint64_t* pa) {
    modup_exact(f->rb, levelB + 1, pb, f->ra, levelA + 1, pa, &f->b2a[levelB]);
}
/* ModDownQPtoQ (basis_extension.go:192-232) */
void ora_fbe_moddown_ab2a(const ora_fbe* f, int levelA, int levelB, const uint64_t* p1a, const uint64_t* p1b, uint64_t* p2a) {
    const ora_ring* ra = f->ra; const size_t N = (size_t)ra->N;
    uint64_t* pool = NEWA(uint64_t, (size_t)(levelA + 1) * N);
    ora_fbe_modup_b2a(f, levelB, levelA, p1b, pool);
    for (int i = 0; i <= levelA; ++i) {
        const uint64_t qi = ra->mod[i], twoqi = qi << 1, qinv = ra->qinv[i];
        const uint64_t params = qi - f->down_b2a[(size_t)levelB * ra->nmod + i];
        const uint64_t *x = p1a + (size_t)i * N, *y = pool + (size_t)i * N;
        uint64_t* z = p2a + (size_t)i * N;
        for (size_t j = 0; j < N; ++j) z[j] = ora_mred(y[j] + twoqi - x[j], params, qi, qinv);
    }
    free(pool);
}
/* ModDownQPtoP (basis_extension.go:292-334).  NB the reference indexes modDownparamsQtoP[levelP][i]. */
void ora_fbe_moddown_ab2b(const ora_fbe* f, int levelA, int levelB, const uint64_t* p1a, const uint64_t* p1b, uint64_t* p2b) {
    const ora_ring* rb = f->rb; const size_t N = (size_t)rb->N;
    uint64_t* pool = NEWA(uint64_t, (size_t)(levelB + 1) * N);
    ora_fbe_modup_a2b(f, levelA, levelB, p1a, pool);
    for (int i = 0; i <= levelB; ++i) {
        const uint64_t qi = rb->mod[i], twoqi = qi << 1, qinv = rb->qinv[i];
        const uint64_t params = qi - f->down_a2b[(size_t)levelB * rb->nmod + i];
        const uint64_t *x = p1b + (size_t)i * N, *y = pool + (size_t)i * N;
        uint64_t* z = p2b + (size_t)i * N;
        for (size_t j = 0; j < N; ++j) z[j] = ora_mred(y[j] + twoqi - x[j], params, qi, qinv);
    }
    free(pool);
}

/* ------------------------------------------------------------------------
 * KeySwitcher
 * ---------------------------------------------------------------------- */
static int ceil_div(int a, int b) { return (a + b - 1) / b; }

ora_ks* ora_ks_new(int logN, const uint64_t* Q, int nq, const uint64_t* P, int np, int gamma,
                   const uint64_t* psiQ, const uint64_t* psiP) {
    if (np < 1 || gamma < 1 || np / gamma < 1) return NULL;
    ora_ks* ks = NEWA(ora_ks, 1);
    ks->rq = ora_ring_new(logN, Q, nq, psiQ);
    ks->rp = ora_ring_new(logN, P, np, psiP);
    ks->gamma = gamma; ks->nq = nq; ks->np = np; ks->N = 1 << logN;
    ks->alpha = np / gamma;                              /* mkrlwe/params.go:63-65 */
    ks->beta_max = ceil_div(nq, ks->alpha);              /* params.go:67-71        */
    ks->conv = ora_fbe_new(ks->rq, ks->rp);
    /* NewDecomposer (basis_extension.go:368-424) */
    ks->dec_nlvl = np - 1;
    ks->dec = NEWA(ora_modup**, ks->dec_nlvl > 0 ? ks->dec_nlvl : 1);
    ks->dec_beta = NEWA(int, ks->dec_nlvl > 0 ? ks->dec_nlvl : 1);
    ks->dec_cnt = NEWA(int*, ks->dec_nlvl > 0 ? ks->dec_nlvl : 1);
    uint64_t* QP = NEWA(uint64_t, nq + np);
    for (int l = 0; l < ks->dec_nlvl; ++l) {
        int lenP = l + 2;
        int alpha = ceil_div(lenP, gamma);
        int beta = ceil_div(nq, alpha);
        ks->dec_beta[l] = beta;
        ks->dec[l] = NEWA(ora_modup*, beta);
        ks->dec_cnt[l] = NEWA(int, beta);
        for (int k = 0; k < nq; ++k) QP[k] = Q[k];
        for (int k = 0; k < lenP; ++k) QP[nq + k] = P[k];
        for (int i = 0; i < beta; ++i) {
            int xa = alpha;
            if (i == beta - 1 && nq % alpha != 0) xa = nq % alpha;
            ks->dec_cnt[l][i] = xa - 1;
            ks->dec[l][i] = NEWA(ora_modup, xa - 1 > 0 ? xa - 1 : 1);
            for (int j = 0; j < xa - 1; ++j)
                ks->dec[l][i][j] = modup_params(Q + i * alpha, j + 2, QP, nq + lenP);
        }
    }
    free(QP);
    const size_t N = (size_t)ks->N, m = (size_t)(nq + np);
    ks->pool0 = NEWA(uint64_t, m * N);
    ks->pool1 = NEWA(uint64_t, m * N);
    ks->pool_invntt = NEWA(uint64_t, (size_t)nq * N);
    for (int i = 0; i < 3; ++i) ks->polyq[i] = NEWA(uint64_t, (size_t)nq * N);
    ks->swk1 = NEWA(uint64_t, ora_ks_swk_words(ks));
    ks->swk2 = NEWA(uint64_t, ora_ks_swk_words(ks));
    ks->swk3 = NEWA(uint64_t, ora_ks_swk_words(ks));
    return ks;
}
void ora_ks_free(ora_ks* ks) {
    if (!ks) return;
    for (int l = 0; l < ks->dec_nlvl; ++l) {
        for (int i = 0; i < ks->dec_beta[l]; ++i) {
            for (int j = 0; j < ks->dec_cnt[l][i]; ++j) modup_free(&ks->dec[l][i][j]);
            free(ks->dec[l][i]);
        }
        free(ks->dec[l]); free(ks->dec_cnt[l]);
    }
    free(ks->dec); free(ks->dec_beta); free(ks->dec_cnt);
    ora_fbe_free(ks->conv);
    free(ks->pool0); free(ks->pool1); free(ks->pool_invntt);
    for (int i = 0; i < 3; ++i) free(ks->polyq[i]);
    free(ks->swk1); free(ks->swk2); free(ks->swk3);
    ora_ring_free(ks->rq); ora_ring_free(ks->rp);
    free(ks);
}
int ora_ks_alpha(const ora_ks* ks) { return ks->alpha; }
int ora_ks_beta(const ora_ks* ks, int levelQ) { return ceil_div(levelQ + 1, ks->alpha); }
size_t ora_ks_swk_words(const ora_ks* ks) { return (size_t)ks->beta_max * (size_t)(ks->nq + ks->np) * (size_t)ks->N; }
const ora_ring* ora_ks_ringq(const ora_ks* ks) { return ks->rq; }
const ora_ring* ora_ks_ringp(const ora_ks* ks) { return ks->rp; }

/* DecomposeAndSplit (basis_extension.go:428-535).  `beta` is the DIGIT INDEX. */
void ora_decompose_and_split(const ora_ks* ks, int levelQ, int levelP, int alpha, int beta, int gamma,
                             const uint64_t* p0q, uint64_t* p1q, uint64_t* p1p) {
    const ora_ring *rq = ks->rq, *rp = ks->rp;
    const size_t N = (size_t)ks->N;
    const int start = beta * alpha;
    int dl;
    if (levelQ > alpha * (beta + 1) - 1) dl = alpha - 2; else dl = (levelQ % alpha) - 1;
    if (dl == -1) {
        for (int j = 0; j <= levelQ; ++j) memcpy(p1q + (size_t)j * N, p0q + (size_t)start * N, N * 8);
        for (int j = 0; j <= levelP; ++j) memcpy(p1p + (size_t)j * N, p0q + (size_t)start * N, N * 8);
        return;
    }
    const ora_modup* mp = &ks->dec[gamma * alpha - 2][beta][dl];
    const int nd = dl + 2, ns = mp->ns;
    uint64_t y[32];
    for (size_t x = 0; x < N; ++x) {
        double vi = 0.0;
        for (int i = 0, j = start; i < nd; ++i, ++j) {
            uint64_t px = p0q[(size_t)j * N + x];
            p1q[(size_t)j * N + x] = px;
            y[i] = ora_mred(px, mp->qoverqiinvqi[i], rq->mod[j], rq->qinv[j]);
            vi += (double)y[i] / (double)rq->mod[j];
        }
        uint64_t v = (uint64_t)vi;
        for (int j = 0; j < start; ++j)
            p1q[(size_t)j * N + x] = multsum(nd, y, v, rq->mod[j], rq->qinv[j],
                mp->vtimesqmodp + (size_t)j * (ns + 1), mp->qoverqimodp + (size_t)j * ns);
        /* the reference starts this loop at alpha*beta, i.e. it re-writes the digit's own limbs too */
        for (int j = alpha * beta; j <= levelQ; ++j)
            p1q[(size_t)j * N + x] = multsum(nd, y, v, rq->mod[j], rq->qinv[j],
                mp->vtimesqmodp + (size_t)j * (ns + 1), mp->qoverqimodp + (size_t)j * ns);
        for (int j = 0, u = rq->nmod; j <= levelP; ++j, ++u)
            p1p[(size_t)j * N + x] = multsum(nd, y, v, rp->mod[j], rp->qinv[j],
                mp->vtimesqmodp + (size_t)u * (ns + 1), mp->qoverqimodp + (size_t)u * ns);
    }
}

/* limb-level parallelism for the multi-thread CPU baseline figure of bench.py (the reference is a single goroutine: the
 * default is 1 thread; every limb loop is independent, results are identical for any thread count) */
int ora_threads = 1;
void ora_set_threads(int n) { ora_threads = n < 1 ? 1 : n; }

#define SWK_Q(ks, swk, i) ((swk) + (size_t)(i) * (size_t)((ks)->nq + (ks)->np) * (size_t)(ks)->N)
#define SWK_P(ks, swk, i) (SWK_Q(ks, swk, i) + (size_t)(ks)->nq * (size_t)(ks)->N)

/* DecomposeSingleNTT (keyswitch.go:21-31) */
static void decompose_single_ntt(const ora_ks* ks, int levelQ, int levelP, int digit,
                                 const uint64_t* a_invntt, uint64_t* cq, uint64_t* cp) {
    const size_t N = (size_t)ks->N;
    ora_decompose_and_split(ks, levelQ, levelP, ks->alpha, digit, ks->gamma, a_invntt, cq, cp);
#pragma omp parallel for schedule(static) if (ora_threads > 1) num_threads(ora_threads)
    for (int j = 0; j <= levelQ; ++j) ora_ntt(ks->rq, j, cq + (size_t)j * N, cq + (size_t)j * N);
#pragma omp parallel for schedule(static) if (ora_threads > 1) num_threads(ora_threads)
    for (int j = 0; j <= levelP; ++j) ora_ntt(ks->rp, j, cp + (size_t)j * N, cp + (size_t)j * N);
}

/* Decompose (keyswitch.go:49-73) */
void ora_decompose(ora_ks* ks, int levelQ, int is_ntt, const uint64_t* a, uint64_t* ad) {
    const size_t N = (size_t)ks->N;
    const uint64_t* ainv = a;
    if (is_ntt) {
        for (int j = 0; j <= levelQ; ++j) ora_intt(ks->rq, j, a + (size_t)j * N, ks->pool_invntt + (size_t)j * N);
        ainv = ks->pool_invntt;
    }
    const int levelP = ks->np - 1, beta = ora_ks_beta(ks, levelQ);
    for (int i = 0; i < beta; ++i)
        decompose_single_ntt(ks, levelQ, levelP, i, ainv, SWK_Q(ks, ad, i), SWK_P(ks, ad, i));
}

static void qp_mul(const ora_ks* ks, int levelQ, int levelP, const uint64_t* a, const uint64_t* b, uint64_t* z, int add) {
    const size_t N = (size_t)ks->N, po = (size_t)ks->nq * N;
#pragma omp parallel for schedule(static) if (ora_threads > 1) num_threads(ora_threads)
    for (int j = 0; j <= levelQ; ++j) {
        if (add) ora_limb_mul_add(ks->rq, j, a + j * N, b + j * N, z + j * N);
        else     ora_limb_mul(ks->rq, j, a + j * N, b + j * N, z + j * N);
    }
    for (int j = 0; j <= levelP; ++j) {
        if (add) ora_limb_mul_add(ks->rp, j, a + po + j * N, b + po + j * N, z + po + j * N);
        else     ora_limb_mul(ks->rp, j, a + po + j * N, b + po + j * N, z + po + j * N);
    }
}
static void qp_mform(const ora_ks* ks, int levelQ, int levelP, uint64_t* a) {
    const size_t N = (size_t)ks->N, po = (size_t)ks->nq * N;
#pragma omp parallel for schedule(static) if (ora_threads > 1) num_threads(ora_threads)
    for (int j = 0; j <= levelQ; ++j) ora_limb_mform(ks->rq, j, a + j * N, a + j * N);
    for (int j = 0; j <= levelP; ++j) ora_limb_mform(ks->rp, j, a + po + j * N, a + po + j * N);
}
/* InvNTTLazy on Q and P halves, then ks.Baseconverter.ModDownQPtoQ (keyswitch.go:114-117) */
static void invntt_moddown(ora_ks* ks, int levelQ, int levelP, uint64_t* c1qp, uint64_t* c) {
    const size_t N = (size_t)ks->N, po = (size_t)ks->nq * N;
#pragma omp parallel for schedule(static) if (ora_threads > 1) num_threads(ora_threads)
    for (int j = 0; j <= levelQ; ++j) ora_intt_lazy(ks->rq, j, c1qp + j * N, c1qp + j * N);
    for (int j = 0; j <= levelP; ++j) ora_intt_lazy(ks->rp, j, c1qp + po + j * N, c1qp + po + j * N);
    ora_fbe_moddown_ab2a(ks->conv, levelQ, levelP, c1qp, c1qp + po, c);
}

/* ExternalProduct (keyswitch.go:79-118) */
void ora_external_product(ora_ks* ks, int levelQ, int is_ntt, const uint64_t* a, const uint64_t* bg, uint64_t* c) {
    const size_t N = (size_t)ks->N;
    const uint64_t* ainv = a;
    if (is_ntt) {
        for (int j = 0; j <= levelQ; ++j) ora_intt(ks->rq, j, a + (size_t)j * N, ks->pool_invntt + (size_t)j * N);
        ainv = ks->pool_invntt;
    }
    const int levelP = ks->np - 1, beta = ora_ks_beta(ks, levelQ);
    uint64_t *c0 = ks->pool0, *c1 = ks->pool1;
    for (int i = 0; i < beta; ++i) {
        decompose_single_ntt(ks, levelQ, levelP, i, ainv, c0, c0 + (size_t)ks->nq * N);
        qp_mul(ks, levelQ, levelP, SWK_Q(ks, bg, i), c0, c1, i != 0);
    }
    invntt_moddown(ks, levelQ, levelP, c1, c);
}

/* ExternalProductHoisted (keyswitch_hoisted.go:10-40) */
void ora_external_product_hoisted(ora_ks* ks, int levelQ, const uint64_t* ah, const uint64_t* bg, uint64_t* c) {
    const int levelP = ks->np - 1, beta = ora_ks_beta(ks, levelQ);
    uint64_t* c1 = ks->pool1;
    for (int i = 0; i < beta; ++i)
        qp_mul(ks, levelQ, levelP, SWK_Q(ks, bg, i), SWK_Q(ks, ah, i), c1, i != 0);
    invntt_moddown(ks, levelQ, levelP, c1, c);
}

static int find_id(int n, const int* ids, int id) { for (int i = 0; i < n; ++i) if (ids[i] == id) return i; return -1; }

/* Steps A-C of MulAndRelin[Hoisted] (keyswitch_hoisted.go:70-117): x = sum_i d_i (.) h(c0_i),
 * y = sum_j b_j (.) h(c1_j), followed by MFormLvl when mform != 0.  mform == 0 exposes the canonical
 * partial sums (used only to check the party-sharded multi-GPU orchestration). */
void ora_mr_xy(ora_ks* ks, int level,
    int n0, const int* ids0, const uint64_t* op0, int op0_limbs,
    int n1, const int* ids1, const uint64_t* op1, int op1_limbs,
    const uint64_t* const* hoist0, const uint64_t* const* hoist1,
    const uint64_t* const* rlk_b, const uint64_t* const* rlk_d,
    uint64_t* x, uint64_t* y, int mform) {
    const size_t N = (size_t)ks->N;
    const int levelP = ks->np - 1, beta = ora_ks_beta(ks, level);
    const size_t s0 = (size_t)op0_limbs * N, s1 = (size_t)op1_limbs * N;
    const size_t swkpoly = (size_t)(ks->nq + ks->np) * N;
    memset(x, 0, (size_t)beta * swkpoly * 8);
    memset(y, 0, (size_t)beta * swkpoly * 8);
    for (int a = 0; a < n0; ++a) {
        int id = ids0[a];
        const uint64_t* h;
        if (!hoist0) { ora_decompose(ks, level, 0, op0 + (size_t)(1 + a) * s0, ks->swk3); h = ks->swk3; }
        else h = hoist0[id];
        for (int i = 0; i < beta; ++i) qp_mul(ks, level, levelP, SWK_Q(ks, rlk_d[id], i), SWK_Q(ks, h, i), SWK_Q(ks, x, i), 1);
    }
    if (mform) for (int i = 0; i < beta; ++i) qp_mform(ks, level, levelP, SWK_Q(ks, x, i));
    for (int a = 0; a < n1; ++a) {
        int id = ids1[a];
        const uint64_t* h;
        if (!hoist1) { ora_decompose(ks, level, 0, op1 + (size_t)(1 + a) * s1, ks->swk3); h = ks->swk3; }
        else h = hoist1[id];
        for (int i = 0; i < beta; ++i) qp_mul(ks, level, levelP, SWK_Q(ks, rlk_b[id], i), SWK_Q(ks, h, i), SWK_Q(ks, y, i), 1);
    }
    if (mform) for (int i = 0; i < beta; ++i) qp_mform(ks, level, levelP, SWK_Q(ks, y, i));
}

/* Steps D-F (keyswitch_hoisted.go:119-178) with x, y given in Montgomery form.  with_c0 == 0 leaves
 * c0_0*c1_0 out of out_0 (sharded evaluation: exactly one rank adds it). */
void ora_mr_finish(ora_ks* ks, int level,
    int n0, const int* ids0, const uint64_t* op0, int op0_limbs,
    int n1, const int* ids1, const uint64_t* op1, int op1_limbs,
    const uint64_t* const* hoist0, const uint64_t* const* hoist1,
    const uint64_t* x, const uint64_t* y,
    const uint64_t* const* rlk_v, const uint64_t* crs_u, int with_c0,
    int nout, const int* ids_out, uint64_t* out) {
    const size_t N = (size_t)ks->N, L = (size_t)(level + 1);
    const size_t s0 = (size_t)op0_limbs * N, s1 = (size_t)op1_limbs * N, so = L * N;
    /* D: tensor */
    uint64_t *p0 = ks->polyq[0], *p1 = ks->polyq[1], *p2 = ks->polyq[2];
    for (size_t j = 0; j < L; ++j) {
        ora_ntt(ks->rq, (int)j, op0 + j * N, p0 + j * N);
        ora_ntt(ks->rq, (int)j, op1 + j * N, p1 + j * N);
        ora_limb_mform(ks->rq, (int)j, p0 + j * N, p0 + j * N);
        if (with_c0) ora_limb_mul(ks->rq, (int)j, p0 + j * N, p1 + j * N, out + j * N);
        else memset(out + j * N, 0, N * 8);
        ora_limb_mform(ks->rq, (int)j, p1 + j * N, p1 + j * N);
    }
    for (int a = 0; a < n0; ++a) {
        int o = find_id(nout, ids_out, ids0[a]);
        for (size_t j = 0; j < L; ++j) {
            ora_ntt(ks->rq, (int)j, op0 + (size_t)(1 + a) * s0 + j * N, p2 + j * N);
            ora_limb_mul(ks->rq, (int)j, p1 + j * N, p2 + j * N, out + (size_t)(1 + o) * so + j * N);
        }
    }
    for (int a = 0; a < n1; ++a) {
        int o = find_id(nout, ids_out, ids1[a]);
        int both = find_id(n0, ids0, ids1[a]) >= 0;
        for (size_t j = 0; j < L; ++j) {
            ora_ntt(ks->rq, (int)j, op1 + (size_t)(1 + a) * s1 + j * N, p2 + j * N);
            if (both) ora_limb_mul_add(ks->rq, (int)j, p0 + j * N, p2 + j * N, out + (size_t)(1 + o) * so + j * N);
            else      ora_limb_mul(ks->rq, (int)j, p0 + j * N, p2 + j * N, out + (size_t)(1 + o) * so + j * N);
        }
    }
    for (int o = 0; o <= nout; ++o)
        for (size_t j = 0; j < L; ++j) ora_intt(ks->rq, (int)j, out + (size_t)o * so + j * N, out + (size_t)o * so + j * N);
    /* E: out_j += <h(c1_j), x>_P */
    for (int a = 0; a < n1; ++a) {
        int id = ids1[a], o = find_id(nout, ids_out, id);
        if (!hoist1) ora_external_product(ks, level, 0, op1 + (size_t)(1 + a) * s1, x, p0);
        else ora_external_product_hoisted(ks, level, hoist1[id], x, p0);
        for (size_t j = 0; j < L; ++j) ora_limb_add(ks->rq, (int)j, out + (size_t)(1 + o) * so + j * N, p0 + j * N, out + (size_t)(1 + o) * so + j * N);
    }
    /* F: t = <h(c0_i), y>_P ; out_0 += <h(t), v_i>_P ; out_i += <h(t), u>_P */
    for (int a = 0; a < n0; ++a) {
        int id = ids0[a], o = find_id(nout, ids_out, id);
        if (!hoist0) ora_external_product(ks, level, 0, op0 + (size_t)(1 + a) * s0, y, p0);
        else ora_external_product_hoisted(ks, level, hoist0[id], y, p0);
        ora_decompose(ks, level, 0, p0, ks->swk3);
        ora_external_product_hoisted(ks, level, ks->swk3, rlk_v[id], p1);
        for (size_t j = 0; j < L; ++j) ora_limb_add(ks->rq, (int)j, out + j * N, p1 + j * N, out + j * N);
        ora_external_product_hoisted(ks, level, ks->swk3, crs_u, p2);
        for (size_t j = 0; j < L; ++j) ora_limb_add(ks->rq, (int)j, out + (size_t)(1 + o) * so + j * N, p2 + j * N, out + (size_t)(1 + o) * so + j * N);
    }
}

/* MulAndRelinHoisted (keyswitch_hoisted.go:44-179); with hoist == NULL it is MulAndRelin
 * (keyswitch.go:122-230), which computes the same values. */
void ora_mul_and_relin(ora_ks* ks, int level,
    int n0, const int* ids0, const uint64_t* op0, int op0_limbs,
    int n1, const int* ids1, const uint64_t* op1, int op1_limbs,
    const uint64_t* const* hoist0, const uint64_t* const* hoist1,
    const uint64_t* const* rlk_b, const uint64_t* const* rlk_d, const uint64_t* const* rlk_v,
    const uint64_t* crs_u, int nout, const int* ids_out, uint64_t* out) {
    ora_mr_xy(ks, level, n0, ids0, op0, op0_limbs, n1, ids1, op1, op1_limbs, hoist0, hoist1, rlk_b, rlk_d, ks->swk1, ks->swk2, 1);
    ora_mr_finish(ks, level, n0, ids0, op0, op0_limbs, n1, ids1, op1, op1_limbs, hoist0, hoist1, ks->swk1, ks->swk2,
                  rlk_v, crs_u, 1, nout, ids_out, out);
}

/* Rotate / RotateHoisted (keyswitch.go:234-298, keyswitch_hoisted.go:183-247) */
void ora_rotate(ora_ks* ks, int level, uint64_t galEl, int n, const int* ids,
    const uint64_t* ct_in, int in_limbs, const uint64_t* const* hoist,
    const uint64_t* const* rk, const uint64_t* crs, uint64_t* ct_out) {
    (void)ids;
    const size_t N = (size_t)ks->N, L = (size_t)(level + 1), si = (size_t)in_limbs * N, so = L * N;
    uint64_t* p0 = ks->polyq[0];
    memcpy(ct_out, ct_in, so * 8);
    for (int a = 0; a < n; ++a) {
        if (!hoist) ora_external_product(ks, level, 0, ct_in + (size_t)(1 + a) * si, rk[a], p0);
        else ora_external_product_hoisted(ks, level, hoist[a], rk[a], p0);
        for (size_t j = 0; j < L; ++j) ora_limb_add(ks->rq, (int)j, ct_out + j * N, p0 + j * N, ct_out + j * N);
        if (!hoist) ora_external_product(ks, level, 0, ct_in + (size_t)(1 + a) * si, crs, ct_out + (size_t)(1 + a) * so);
        else ora_external_product_hoisted(ks, level, hoist[a], crs, ct_out + (size_t)(1 + a) * so);
    }
    for (int a = 0; a <= n; ++a) {
        ora_permute(ks->rq, level, galEl, ct_out + (size_t)a * so, p0);
        memcpy(ct_out + (size_t)a * so, p0, so * 8);
    }
}

/* Conjugate (keyswitch.go:302-332) */
void ora_conjugate(ora_ks* ks, int level, uint64_t galEl, int n, const int* ids,
    const uint64_t* ct_in, int in_limbs, const uint64_t* const* ck, const uint64_t* crs, uint64_t* ct_out) {
    (void)ids;
    const size_t N = (size_t)ks->N, L = (size_t)(level + 1), si = (size_t)in_limbs * N, so = L * N;
    uint64_t* p0 = ks->polyq[0];
    for (int a = 0; a <= n; ++a) ora_permute(ks->rq, level, galEl, ct_in + (size_t)a * si, ct_out + (size_t)a * so);
    for (int a = 0; a < n; ++a) {
        ora_external_product(ks, level, 0, ct_out + (size_t)(1 + a) * so, ck[a], p0);
        for (size_t j = 0; j < L; ++j) ora_limb_add(ks->rq, (int)j, ct_out + j * N, p0 + j * N, ct_out + j * N);
    }
    for (int a = 0; a < n; ++a) {
        ora_external_product(ks, level, 0, ct_out + (size_t)(1 + a) * so, crs, p0);
        memcpy(ct_out + (size_t)(1 + a) * so, p0, so * 8);
    }
}

/* mkckks.Evaluator.Rescale scale loop (evaluator.go:376-384) */
int ora_ckks_nb_rescales(const ora_ring* rq, int level, double* scale, double min_scale) {
    int nb = 0;
    while (level - nb >= 0 && *scale / (double)rq->mod[level - nb] >= min_scale / 2) {
        *scale /= (double)rq->mod[level - nb];
        nb++;
    }
    return nb;
}
