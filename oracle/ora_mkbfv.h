/* ora_mkbfv.h -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C restatement of the reference's multi-key BFV multiplication path:
 *   mkbfv/params.go:30-76              rings Q, QMul, R = Q || QMul, P; paramsRP
 *   mkbfv/basis_extension.go:20-96     FastBasisExtender: ModUpQtoR, Quantize, Rescale
 *   mkbfv/keyswitch.go:67-114          DecomposeBFV, ExternalProductBFV
 *   mkbfv/keyswitch_hoisted.go:6-206   ExternalProductBFVHoisted, MulAndRelinBFVHoisted
 *   mkbfv/keyswitch.go:116-250         MulAndRelinBFV (same values, digits recomputed)
 *   mkbfv/evaluator.go:99-140          mulRelin / mulRelinHoisted (MulRelinNew)
 * Same operation order, single thread.  PARITY UNPINNED vs Go (see ora_ring.h).
 *
 * Only alpha = PCount/gamma = 1 is restated: with alpha >= 2 the reference's DecomposeBFV asks the
 * R-ring decomposer for digits with the Q level (keyswitch.go:73-76), which is only meaningful for
 * one prime per digit -- and alpha = 1 is what both mkbfv parameter sets use (mkbfv_test.go:29-112).
 * BFV ciphertexts live at the maximum level, in the coefficient domain (elements.go:9-11,
 * encryptor.go:49-58).
 *
 * Layouts: PolyR = uint64[2*nQ][N], limbs 0..nQ-1 under Q, nQ..2nQ-1 under QMul.
 *          BFV hoisted form of one polynomial = two SwitchingKey arrays (Q digits, QMul digits).
 */
#ifndef ORA_MKBFV_H
#define ORA_MKBFV_H
#include "ora_mkrlwe.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_bfv {
    ora_ks* ks;              /* mkrlwe.KeySwitcher over (Q, P)                 */
    ora_ring* rqm;           /* ring QMul                                      */
    ora_fbe* conv;           /* convQQMul = mkrlwe.NewFastBasisExtender(Q,QMul)*/
    uint64_t* mform_qmul;    /* [nq] MForm(QMul mod q_i)  basis_extension.go:42-44 */
    uint64_t t;              /* plaintext modulus                               */
    int nq, N;
    uint64_t *poolq, *poolqm, *poolr;   /* conv.polypoolQ / QMul / R            */
    uint64_t *swk[6];        /* swkPool1..6                                     */
    uint64_t *pq[2];         /* polyQPool1..2                                   */
    uint64_t *pr[4];         /* polyRPool1..4                                   */
} ora_bfv;

ora_bfv* ora_bfv_new(int logN, const uint64_t* Q, const uint64_t* QMul, int nq,
                     const uint64_t* P, int np, int gamma, uint64_t t);
void ora_bfv_free(ora_bfv* b);
ora_ks* ora_bfv_ks(ora_bfv* b);
const ora_ring* ora_bfv_ringqmul(const ora_bfv* b);

/* FastBasisExtender (mkbfv/basis_extension.go:49-96) */
void ora_bfv_modup_q_to_r(ora_bfv* b, const uint64_t* polyq, uint64_t* polyr);
void ora_bfv_rescale(ora_bfv* b, const uint64_t* polyq, uint64_t* polyr);
void ora_bfv_quantize(ora_bfv* b, const uint64_t* polyr_ntt, uint64_t* polyq);

/* DecomposeBFV (keyswitch.go:67-90): aR coefficient domain -> ad1 (Q digits), ad2 (QMul digits) */
void ora_bfv_decompose(ora_bfv* b, const uint64_t* ar, uint64_t* ad1, uint64_t* ad2);
/* ExternalProductBFV[Hoisted] (keyswitch.go:92-114, keyswitch_hoisted.go:6-34) */
void ora_bfv_external_product(ora_bfv* b, const uint64_t* ar, const uint64_t* bg1, const uint64_t* bg2, uint64_t* c);
void ora_bfv_external_product_hoisted(ora_bfv* b, const uint64_t* ah1, const uint64_t* ah2,
                                      const uint64_t* bg1, const uint64_t* bg2, uint64_t* c);

/* MulAndRelinBFV[Hoisted].  op0r/op1r: uint64[1+n][2nQ][N] (ModUpQtoR / Rescale outputs);
 * h0a/h0b/h1a/h1b: per party id hoisted forms or all NULL (non-hoisted twin);
 * rlk_*: per party id; out: uint64[1+nout][nQ][N]. */
void ora_bfv_mul_and_relin(ora_bfv* b,
    int n0, const int* ids0, const uint64_t* op0r, int n1, const int* ids1, const uint64_t* op1r,
    const uint64_t* const* h0a, const uint64_t* const* h0b, const uint64_t* const* h1a, const uint64_t* const* h1b,
    const uint64_t* const* rlk_b1, const uint64_t* const* rlk_b2,
    const uint64_t* const* rlk_d1, const uint64_t* const* rlk_d2, const uint64_t* const* rlk_v,
    const uint64_t* crs_u, int nout, const int* ids_out, uint64_t* out);

/* Evaluator.MulRelinNew = mulRelinHoisted (evaluator.go:78-82,118-140) on Q ciphertexts
 * uint64[1+n][nQ][N]; hoisted != 0 takes the hoisted route, 0 the mulRelin route (:99-116). */
void ora_bfv_mul_relin_new(ora_bfv* b,
    int n0, const int* ids0, const uint64_t* op0, int n1, const int* ids1, const uint64_t* op1,
    const uint64_t* const* rlk_b1, const uint64_t* const* rlk_b2,
    const uint64_t* const* rlk_d1, const uint64_t* const* rlk_d2, const uint64_t* const* rlk_v,
    const uint64_t* crs_u, int hoisted, int nout, const int* ids_out, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
