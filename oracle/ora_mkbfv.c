/* ora_mkbfv.c -- CPU ORACLE (test infrastructure only; see ora_mkbfv.h header). */
#include "ora_mkbfv.h"
#include <stdlib.h>
#include <string.h>

#define NEWA(T, n) ((T*)calloc((size_t)(n), sizeof(T)))
#define SWK_AT(b, swk, i) ((swk) + (size_t)(i) * (size_t)((b)->ks->nq + (b)->ks->np) * (size_t)(b)->N)

static uint64_t to_mont(uint64_t a, uint64_t q) { return (uint64_t)((((u128)a) << 64) % q); }
static int find_id(int n, const int* ids, int id) { for (int i = 0; i < n; ++i) if (ids[i] == id) return i; return -1; }

ora_bfv* ora_bfv_new(int logN, const uint64_t* Q, const uint64_t* QMul, int nq,
                     const uint64_t* P, int np, int gamma, uint64_t t) {
    if (np < 1 || gamma < 1 || np / gamma != 1) return NULL;      /* alpha = 1 only, see header */
    ora_bfv* b = NEWA(ora_bfv, 1);
    b->ks = ora_ks_new(logN, Q, nq, P, np, gamma, NULL, NULL);
    if (!b->ks) { free(b); return NULL; }
    b->rqm = ora_ring_new(logN, QMul, nq, NULL);
    b->conv = ora_fbe_new(b->ks->rq, b->rqm);
    b->t = t; b->nq = nq; b->N = 1 << logN;
    /* mFormQMul = MForm(0 + QMul mod q_i)  (basis_extension.go:42-44) */
    b->mform_qmul = NEWA(uint64_t, nq);
    for (int i = 0; i < nq; ++i) {
        uint64_t qi = Q[i], m = 1 % qi;
        for (int j = 0; j < nq; ++j) m = ora_mulmod(m, QMul[j] % qi, qi);
        b->mform_qmul[i] = to_mont(m, qi);
    }
    const size_t N = (size_t)b->N;
    b->poolq = NEWA(uint64_t, (size_t)nq * N);
    b->poolqm = NEWA(uint64_t, (size_t)nq * N);
    b->poolr = NEWA(uint64_t, 2 * (size_t)nq * N);
    for (int i = 0; i < 6; ++i) b->swk[i] = NEWA(uint64_t, ora_ks_swk_words(b->ks));
    for (int i = 0; i < 2; ++i) b->pq[i] = NEWA(uint64_t, (size_t)nq * N);
    for (int i = 0; i < 4; ++i) b->pr[i] = NEWA(uint64_t, 2 * (size_t)nq * N);
    return b;
}
void ora_bfv_free(ora_bfv* b) {
    if (!b) return;
    ora_fbe_free(b->conv);
    ora_ring_free(b->rqm);
    ora_ks_free(b->ks);
    free(b->mform_qmul); free(b->poolq); free(b->poolqm); free(b->poolr);
    for (int i = 0; i < 6; ++i) free(b->swk[i]);
    for (int i = 0; i < 2; ++i) free(b->pq[i]);
    for (int i = 0; i < 4; ++i) free(b->pr[i]);
    free(b);
}
ora_ks* ora_bfv_ks(ora_bfv* b) { return b->ks; }
const ora_ring* ora_bfv_ringqmul(const ora_bfv* b) { return b->rqm; }

/* ring R = Q || QMul: limb j of a PolyR lives in ring Q (j < nq) or QMul (params.go:36-38,62-69) */
static const ora_ring* r_ring(const ora_bfv* b, int j, int* idx) {
    if (j < b->nq) { *idx = j; return b->ks->rq; }
    *idx = j - b->nq; return b->rqm;
}
static void r_ntt(const ora_bfv* b, const uint64_t* in, uint64_t* out) {
    const size_t N = (size_t)b->N;
    for (int j = 0; j < 2 * b->nq; ++j) { int i; const ora_ring* r = r_ring(b, j, &i); ora_ntt(r, i, in + j * N, out + j * N); }
}
static void r_mform(const ora_bfv* b, uint64_t* a) {
    const size_t N = (size_t)b->N;
    for (int j = 0; j < 2 * b->nq; ++j) { int i; const ora_ring* r = r_ring(b, j, &i); ora_limb_mform(r, i, a + j * N, a + j * N); }
}
static void r_mul(const ora_bfv* b, const uint64_t* x, const uint64_t* y, uint64_t* z, int add) {
    const size_t N = (size_t)b->N;
    for (int j = 0; j < 2 * b->nq; ++j) {
        int i; const ora_ring* r = r_ring(b, j, &i);
        if (add) ora_limb_mul_add(r, i, x + j * N, y + j * N, z + j * N);
        else     ora_limb_mul(r, i, x + j * N, y + j * N, z + j * N);
    }
}

/* ModUpQtoR (basis_extension.go:49-64) */
void ora_bfv_modup_q_to_r(ora_bfv* b, const uint64_t* polyq, uint64_t* polyr) {
    const size_t N = (size_t)b->N; const int lq = b->nq - 1;
    ora_fbe_modup_a2b(b->conv, lq, lq, polyq, b->poolqm);
    memcpy(polyr, polyq, (size_t)b->nq * N * 8);
    memcpy(polyr + (size_t)b->nq * N, b->poolqm, (size_t)b->nq * N * 8);
}

/* Quantize (basis_extension.go:66-80): round(t/QMul * x), x given in the NTT domain over R */
void ora_bfv_quantize(ora_bfv* b, const uint64_t* polyr, uint64_t* polyq) {
    const size_t N = (size_t)b->N; const int lq = b->nq - 1;
    for (int j = 0; j < 2 * b->nq; ++j) {
        int i; const ora_ring* r = r_ring(b, j, &i);
        ora_limb_mul_scalar(r, i, polyr + j * N, b->t, b->poolr + j * N);
        ora_intt(r, i, b->poolr + j * N, b->poolr + j * N);
    }
    memcpy(b->poolq, b->poolr, (size_t)b->nq * N * 8);
    memcpy(b->poolqm, b->poolr + (size_t)b->nq * N, (size_t)b->nq * N * 8);
    ora_fbe_moddown_ab2a(b->conv, lq, lq, b->poolq, b->poolqm, polyq);
}

/* Rescale (basis_extension.go:82-96): x -> round(QMul/Q * x) carried to the basis R */
void ora_bfv_rescale(ora_bfv* b, const uint64_t* polyq, uint64_t* polyr) {
    const size_t N = (size_t)b->N; const int lq = b->nq - 1;
    const ora_ring* rq = b->ks->rq;
    for (int i = 0; i < b->nq; ++i) {                        /* MulCoeffsMontgomery(polyQ, mFormQMul) */
        const uint64_t q = rq->mod[i], qinv = rq->qinv[i], c = b->mform_qmul[i];
        const uint64_t* x = polyq + (size_t)i * N; uint64_t* z = b->poolq + (size_t)i * N;
        for (size_t j = 0; j < N; ++j) z[j] = ora_mred(x[j], c, q, qinv);
    }
    memset(b->poolqm, 0, (size_t)b->nq * N * 8);             /* MulScalar(.., 0, ..)                  */
    ora_fbe_moddown_ab2b(b->conv, lq, lq, b->poolq, b->poolqm, b->poolqm);
    ora_fbe_modup_b2a(b->conv, lq, lq, b->poolqm, b->poolq);
    memcpy(polyr, b->poolq, (size_t)b->nq * N * 8);
    memcpy(polyr + (size_t)b->nq * N, b->poolqm, (size_t)b->nq * N * 8);
}

/* DecomposeBFV (keyswitch.go:67-90) with alpha = 1: digit i = limb i of aR, copied under every
 * modulus of Q and P (DecomposeAndSplit, mkrlwe/basis_extension.go:443-451) and NTT'd there
 * (DecomposeSingleNTT, mkrlwe/keyswitch.go:21-31; the first nQ moduli of ring R are those of Q). */
void ora_bfv_decompose(ora_bfv* b, const uint64_t* ar, uint64_t* ad1, uint64_t* ad2) {
    const ora_ks* ks = b->ks; const size_t N = (size_t)b->N;
    const int nq = ks->nq, np = ks->np;
    for (int d = 0; d < 2 * nq; ++d) {
        uint64_t* dst = d < nq ? SWK_AT(b, ad1, d) : SWK_AT(b, ad2, d - nq);
        const uint64_t* src = ar + (size_t)d * N;
        for (int j = 0; j < nq; ++j) ora_ntt(ks->rq, j, src, dst + (size_t)j * N);
        for (int j = 0; j < np; ++j) ora_ntt(ks->rp, j, src, dst + (size_t)(nq + j) * N);
    }
}

static void qp_mul(const ora_ks* ks, const uint64_t* a, const uint64_t* x, uint64_t* z, int add) {
    const size_t N = (size_t)ks->N, po = (size_t)ks->nq * N;
    for (int j = 0; j < ks->nq; ++j) {
        if (add) ora_limb_mul_add(ks->rq, j, a + j * N, x + j * N, z + j * N);
        else     ora_limb_mul(ks->rq, j, a + j * N, x + j * N, z + j * N);
    }
    for (int j = 0; j < ks->np; ++j) {
        if (add) ora_limb_mul_add(ks->rp, j, a + po + j * N, x + po + j * N, z + po + j * N);
        else     ora_limb_mul(ks->rp, j, a + po + j * N, x + po + j * N, z + po + j * N);
    }
}
static void swk_mform(const ora_bfv* b, uint64_t* s) {
    const ora_ks* ks = b->ks; const size_t N = (size_t)b->N, po = (size_t)ks->nq * N;
    for (int i = 0; i < ks->beta_max; ++i) {
        uint64_t* a = SWK_AT(b, s, i);
        for (int j = 0; j < ks->nq; ++j) ora_limb_mform(ks->rq, j, a + j * N, a + j * N);
        for (int j = 0; j < ks->np; ++j) ora_limb_mform(ks->rp, j, a + po + j * N, a + po + j * N);
    }
}

/* ExternalProductBFVHoisted (keyswitch_hoisted.go:6-34) */
void ora_bfv_external_product_hoisted(ora_bfv* b, const uint64_t* ah1, const uint64_t* ah2,
                                      const uint64_t* bg1, const uint64_t* bg2, uint64_t* c) {
    ora_ks* ks = b->ks; const size_t N = (size_t)b->N, po = (size_t)ks->nq * N;
    uint64_t* c1 = ks->pool1;
    for (int i = 0; i < ks->beta_max; ++i) {
        qp_mul(ks, SWK_AT(b, bg1, i), SWK_AT(b, ah1, i), c1, i != 0);
        qp_mul(ks, SWK_AT(b, bg2, i), SWK_AT(b, ah2, i), c1, 1);
    }
    for (int j = 0; j < ks->nq; ++j) ora_intt_lazy(ks->rq, j, c1 + j * N, c1 + j * N);
    for (int j = 0; j < ks->np; ++j) ora_intt_lazy(ks->rp, j, c1 + po + j * N, c1 + po + j * N);
    ora_fbe_moddown_ab2a(ks->conv, ks->nq - 1, ks->np - 1, c1, c1 + po, c);
}
/* ExternalProductBFV (keyswitch.go:92-114) */
void ora_bfv_external_product(ora_bfv* b, const uint64_t* ar, const uint64_t* bg1, const uint64_t* bg2, uint64_t* c) {
    ora_bfv_decompose(b, ar, b->swk[0], b->swk[1]);
    ora_bfv_external_product_hoisted(b, b->swk[0], b->swk[1], bg1, bg2, c);
}

/* MulAndRelinBFVHoisted (keyswitch_hoisted.go:36-206); all hoisted lists NULL = MulAndRelinBFV (keyswitch.go:116-250) */
void ora_bfv_mul_and_relin(ora_bfv* b,
    int n0, const int* ids0, const uint64_t* op0r, int n1, const int* ids1, const uint64_t* op1r,
    const uint64_t* const* h0a, const uint64_t* const* h0b, const uint64_t* const* h1a, const uint64_t* const* h1b,
    const uint64_t* const* rlk_b1, const uint64_t* const* rlk_b2,
    const uint64_t* const* rlk_d1, const uint64_t* const* rlk_d2, const uint64_t* const* rlk_v,
    const uint64_t* crs_u, int nout, const int* ids_out, uint64_t* out) {
    ora_ks* ks = b->ks;
    const size_t N = (size_t)b->N, PR = 2 * (size_t)b->nq * N, PQ = (size_t)b->nq * N;
    const int beta = ks->beta_max, level = ks->nq - 1;
    uint64_t *x1 = b->swk[2], *x2 = b->swk[3], *y1 = b->swk[4], *y2 = b->swk[5];
    const size_t swkb = ora_ks_swk_words(ks) * 8;
    memset(x1, 0, swkb); memset(x2, 0, swkb); memset(y1, 0, swkb); memset(y2, 0, swkb);
    /* x1 = sum d1_id (.) h1(c0_id), x2 likewise with d2 / QMul digits; then MForm (:76-98) */
    for (int a = 0; a < n0; ++a) {
        const int id = ids0[a];
        const uint64_t *ha, *hb;
        if (!h0a) { ora_bfv_decompose(b, op0r + (size_t)(1 + a) * PR, b->swk[0], b->swk[1]); ha = b->swk[0]; hb = b->swk[1]; }
        else { ha = h0a[id]; hb = h0b[id]; }
        for (int i = 0; i < beta; ++i) {
            qp_mul(ks, SWK_AT(b, rlk_d1[id], i), SWK_AT(b, ha, i), SWK_AT(b, x1, i), 1);
            qp_mul(ks, SWK_AT(b, rlk_d2[id], i), SWK_AT(b, hb, i), SWK_AT(b, x2, i), 1);
        }
    }
    swk_mform(b, x1); swk_mform(b, x2);
    for (int a = 0; a < n1; ++a) {
        const int id = ids1[a];
        const uint64_t *ha, *hb;
        if (!h1a) { ora_bfv_decompose(b, op1r + (size_t)(1 + a) * PR, b->swk[0], b->swk[1]); ha = b->swk[0]; hb = b->swk[1]; }
        else { ha = h1a[id]; hb = h1b[id]; }
        for (int i = 0; i < beta; ++i) {
            qp_mul(ks, SWK_AT(b, rlk_b1[id], i), SWK_AT(b, ha, i), SWK_AT(b, y1, i), 1);
            qp_mul(ks, SWK_AT(b, rlk_b2[id], i), SWK_AT(b, hb, i), SWK_AT(b, y2, i), 1);
        }
    }
    swk_mform(b, y1); swk_mform(b, y2);

    /* tensor over R and Quantize (:128-166) */
    uint64_t *p1 = b->pr[0], *p2 = b->pr[1], *p3 = b->pr[2], *p4 = b->pr[3];
    r_ntt(b, op0r, p1);
    r_ntt(b, op1r, p2);
    r_mform(b, p1);
    r_mul(b, p1, p2, p3, 0);
    ora_bfv_quantize(b, p3, out);
    r_mform(b, p2);
    for (int a = 0; a < n0; ++a) {
        if (find_id(n1, ids1, ids0[a]) >= 0) continue;
        const int o = find_id(nout, ids_out, ids0[a]);
        r_ntt(b, op0r + (size_t)(1 + a) * PR, p3);
        r_mul(b, p2, p3, p3, 0);
        ora_bfv_quantize(b, p3, out + (size_t)(1 + o) * PQ);
    }
    for (int a = 0; a < n1; ++a) {
        const int o = find_id(nout, ids_out, ids1[a]);
        const int a0 = find_id(n0, ids0, ids1[a]);
        r_ntt(b, op1r + (size_t)(1 + a) * PR, p3);
        r_mul(b, p1, p3, p3, 0);
        if (a0 >= 0) {
            r_ntt(b, op0r + (size_t)(1 + a0) * PR, p4);
            r_mul(b, p2, p4, p3, 1);
        }
        ora_bfv_quantize(b, p3, out + (size_t)(1 + o) * PQ);
    }
    /* out_j += <h(c1_j), (x1,x2)> (:168-176) */
    for (int a = 0; a < n1; ++a) {
        const int id = ids1[a], o = find_id(nout, ids_out, id);
        if (!h1a) ora_bfv_external_product(b, op1r + (size_t)(1 + a) * PR, x1, x2, b->pq[0]);
        else ora_bfv_external_product_hoisted(b, h1a[id], h1b[id], x1, x2, b->pq[0]);
        for (int j = 0; j <= level; ++j)
            ora_limb_add(ks->rq, j, out + (size_t)(1 + o) * PQ + j * N, b->pq[0] + j * N, out + (size_t)(1 + o) * PQ + j * N);
    }
    /* t = <h(c0_i), (y1,y2)> ; out_0 += <h(t), v_i> ; out_i += <h(t), u> (:182-205) */
    for (int a = 0; a < n0; ++a) {
        const int id = ids0[a], o = find_id(nout, ids_out, id);
        if (!h0a) ora_bfv_external_product(b, op0r + (size_t)(1 + a) * PR, y1, y2, b->pq[0]);
        else ora_bfv_external_product_hoisted(b, h0a[id], h0b[id], y1, y2, b->pq[0]);
        ora_decompose(ks, level, 0, b->pq[0], b->swk[2]);
        ora_external_product_hoisted(ks, level, b->swk[2], rlk_v[id], b->pq[1]);
        for (int j = 0; j <= level; ++j) ora_limb_add(ks->rq, j, out + j * N, b->pq[1] + j * N, out + j * N);
        ora_external_product_hoisted(ks, level, b->swk[2], crs_u, b->pq[1]);
        for (int j = 0; j <= level; ++j)
            ora_limb_add(ks->rq, j, out + (size_t)(1 + o) * PQ + j * N, b->pq[1] + j * N, out + (size_t)(1 + o) * PQ + j * N);
    }
}

/* Evaluator.mulRelinHoisted / mulRelin (evaluator.go:99-140) */
void ora_bfv_mul_relin_new(ora_bfv* b,
    int n0, const int* ids0, const uint64_t* op0, int n1, const int* ids1, const uint64_t* op1,
    const uint64_t* const* rlk_b1, const uint64_t* const* rlk_b2,
    const uint64_t* const* rlk_d1, const uint64_t* const* rlk_d2, const uint64_t* const* rlk_v,
    const uint64_t* crs_u, int hoisted, int nout, const int* ids_out, uint64_t* out) {
    const size_t N = (size_t)b->N, PR = 2 * (size_t)b->nq * N, PQ = (size_t)b->nq * N;
    uint64_t* r0 = NEWA(uint64_t, (size_t)(1 + n0) * PR);
    uint64_t* r1 = NEWA(uint64_t, (size_t)(1 + n1) * PR);
    for (int a = 0; a <= n0; ++a) ora_bfv_modup_q_to_r(b, op0 + (size_t)a * PQ, r0 + (size_t)a * PR);
    for (int a = 0; a <= n1; ++a) ora_bfv_rescale(b, op1 + (size_t)a * PQ, r1 + (size_t)a * PR);
    if (!hoisted) {
        ora_bfv_mul_and_relin(b, n0, ids0, r0, n1, ids1, r1, NULL, NULL, NULL, NULL,
                              rlk_b1, rlk_b2, rlk_d1, rlk_d2, rlk_v, crs_u, nout, ids_out, out);
    } else {
        int maxid = 0;
        for (int a = 0; a < n0; ++a) if (ids0[a] > maxid) maxid = ids0[a];
        for (int a = 0; a < n1; ++a) if (ids1[a] > maxid) maxid = ids1[a];
        const size_t sw = ora_ks_swk_words(b->ks);
        const uint64_t** h[4];
        for (int k = 0; k < 4; ++k) h[k] = NEWA(const uint64_t*, maxid + 1);
        for (int a = 0; a < n0; ++a) {
            uint64_t *u1 = NEWA(uint64_t, sw), *u2 = NEWA(uint64_t, sw);
            ora_bfv_decompose(b, r0 + (size_t)(1 + a) * PR, u1, u2);
            h[0][ids0[a]] = u1; h[1][ids0[a]] = u2;
        }
        for (int a = 0; a < n1; ++a) {
            uint64_t *u1 = NEWA(uint64_t, sw), *u2 = NEWA(uint64_t, sw);
            ora_bfv_decompose(b, r1 + (size_t)(1 + a) * PR, u1, u2);
            h[2][ids1[a]] = u1; h[3][ids1[a]] = u2;
        }
        ora_bfv_mul_and_relin(b, n0, ids0, r0, n1, ids1, r1, h[0], h[1], h[2], h[3],
                              rlk_b1, rlk_b2, rlk_d1, rlk_d2, rlk_v, crs_u, nout, ids_out, out);
        for (int k = 0; k < 4; ++k) { for (int i = 0; i <= maxid; ++i) free((void*)h[k][i]); free((void*)h[k]); }
    }
    free(r0); free(r1);
}
