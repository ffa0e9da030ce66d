/* ora_mkrlwe.h -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C restatement of the reference's multi-key RLWE hot path:
 *   mkrlwe/basis_extension.go  (FastBasisExtender, Decomposer, modUpExact, multSum)
 *   mkrlwe/keyswitch.go        (Decompose, ExternalProduct, MulAndRelin, Rotate, Conjugate)
 *   mkrlwe/keyswitch_hoisted.go(ExternalProductHoisted, MulAndRelinHoisted, RotateHoisted)
 *   mkckks/evaluator.go:359-443,543-617 (Rescale, MulRelinNew, HoistedForm, Rotate*)
 *   mkbfv/{basis_extension,keyswitch,keyswitch_hoisted,evaluator}.go
 * Same operation order as the reference, single thread.  PARITY UNPINNED vs Go
 * (no golden vectors in the reference, no Go toolchain) -- see ora_ring.h.
 *
 * Layouts (shared with the HIP engine):
 *   PolyQ   at level L : uint64[L+1][N]            limb-major
 *   SwitchingKey       : uint64[betaMax][nQ+nP][N] Q limbs first, then P limbs
 *   Ciphertext         : uint64[1+n][limbs][N]     slot 0 = c_0 ("0"), slot 1+i = party ids[i]
 */
#ifndef ORA_MKRLWE_H
#define ORA_MKRLWE_H
#include "ora_ring.h"

#ifdef __cplusplus
extern "C" {
#endif

/* basisextenderparameters (mkrlwe/basis_extension.go:83-153) */
typedef struct ora_modup {
    int ns, nt;
    uint64_t* qoverqiinvqi;   /* [ns]        */
    uint64_t* qoverqimodp;    /* [nt][ns]    */
    uint64_t* vtimesqmodp;    /* [nt][ns+1]  */
} ora_modup;

/* FastBasisExtender between ring A ("Q") and ring B ("P") (basis_extension.go:56-81) */
typedef struct ora_fbe {
    const ora_ring *ra, *rb;
    ora_modup* a2b;           /* [na] : A[:i+1] -> B */
    ora_modup* b2a;           /* [nb] : B[:i+1] -> A */
    uint64_t* down_b2a;       /* [nb][na]  genModDownParams(ringA, ringB) */
    uint64_t* down_a2b;       /* [na][nb]  genModDownParams(ringB, ringA) */
} ora_fbe;

ora_fbe* ora_fbe_new(const ora_ring* ra, const ora_ring* rb);
void     ora_fbe_free(ora_fbe* f);
void ora_fbe_modup_a2b(const ora_fbe* f, int levelA, int levelB, const uint64_t* pa, uint64_t* pb);
void ora_fbe_modup_b2a(const ora_fbe* f, int levelB, int levelA, const uint64_t* pb, uint64_t* pa);
void ora_fbe_moddown_ab2a(const ora_fbe* f, int levelA, int levelB, const uint64_t* p1a, const uint64_t* p1b, uint64_t* p2a);
void ora_fbe_moddown_ab2b(const ora_fbe* f, int levelA, int levelB, const uint64_t* p1a, const uint64_t* p1b, uint64_t* p2b);

/* KeySwitcher (mkrlwe/keyswitch.go:8-47) */
typedef struct ora_ks {
    ora_ring *rq, *rp;
    int gamma, alpha, beta_max, nq, np, N;
    ora_fbe* conv;                     /* ks.Baseconverter                    */
    /* Decomposer tables: modup[lvlP][digit][j]  (basis_extension.go:368-424) */
    ora_modup*** dec; int dec_nlvl; int* dec_beta; int** dec_cnt;
    uint64_t *pool0, *pool1;           /* ks.Pool[0], ks.Pool[1]   (PolyQP)   */
    uint64_t *pool_invntt;             /* ks.PoolInvNTT            (PolyQ)    */
    uint64_t *polyq[3];                /* ks.polyQPool                         */
    uint64_t *swk1, *swk2, *swk3;      /* ks.swkPool1..3                       */
} ora_ks;

void ora_set_threads(int n);          /* limb-level OpenMP threads of the hot loops (default 1, like the single-goroutine reference) */
ora_ks* ora_ks_new(int logN, const uint64_t* Q, int nq, const uint64_t* P, int np, int gamma,
                   const uint64_t* psiQ_or_null, const uint64_t* psiP_or_null);
void ora_ks_free(ora_ks* ks);
int  ora_ks_alpha(const ora_ks* ks);
int  ora_ks_beta(const ora_ks* ks, int levelQ);
size_t ora_ks_swk_words(const ora_ks* ks);           /* betaMax*(nq+np)*N */
const ora_ring* ora_ks_ringq(const ora_ks* ks);
const ora_ring* ora_ks_ringp(const ora_ks* ks);

void ora_decompose_and_split(const ora_ks* ks, int levelQ, int levelP, int alpha, int beta, int gamma,
                             const uint64_t* p0q, uint64_t* p1q, uint64_t* p1p);
void ora_decompose(ora_ks* ks, int levelQ, int is_ntt, const uint64_t* a, uint64_t* ad);
void ora_external_product(ora_ks* ks, int levelQ, int is_ntt, const uint64_t* a, const uint64_t* bg, uint64_t* c);
void ora_external_product_hoisted(ora_ks* ks, int levelQ, const uint64_t* ah, const uint64_t* bg, uint64_t* c);

/* MulAndRelin[Hoisted] (keyswitch.go:122-230, keyswitch_hoisted.go:44-179).
 * ids are dense party indices >= 0; rlk_b/d/v and hoist pointers are indexed by party id.
 * hoist0/hoist1 may be NULL (non-hoisted twin).  out ids must be the union. */
void ora_mul_and_relin(ora_ks* ks, int level,
    int n0, const int* ids0, const uint64_t* op0, int op0_limbs,
    int n1, const int* ids1, const uint64_t* op1, int op1_limbs,
    const uint64_t* const* hoist0, const uint64_t* const* hoist1,
    const uint64_t* const* rlk_b, const uint64_t* const* rlk_d, const uint64_t* const* rlk_v,
    const uint64_t* crs_u,
    int nout, const int* ids_out, uint64_t* out);

/* the same in two phases (x/y accumulation, then tensor + relinearisation) -- used to check the
 * party-sharded orchestration; ora_mul_and_relin == ora_mr_xy(mform=1) + ora_mr_finish(with_c0=1). */
void ora_mr_xy(ora_ks* ks, int level,
    int n0, const int* ids0, const uint64_t* op0, int op0_limbs,
    int n1, const int* ids1, const uint64_t* op1, int op1_limbs,
    const uint64_t* const* hoist0, const uint64_t* const* hoist1,
    const uint64_t* const* rlk_b, const uint64_t* const* rlk_d,
    uint64_t* x, uint64_t* y, int mform);
void ora_mr_finish(ora_ks* ks, int level,
    int n0, const int* ids0, const uint64_t* op0, int op0_limbs,
    int n1, const int* ids1, const uint64_t* op1, int op1_limbs,
    const uint64_t* const* hoist0, const uint64_t* const* hoist1,
    const uint64_t* x, const uint64_t* y,
    const uint64_t* const* rlk_v, const uint64_t* crs_u, int with_c0,
    int nout, const int* ids_out, uint64_t* out);

/* Rotate[Hoisted] (keyswitch.go:234-298, keyswitch_hoisted.go:183-247); hoist may be NULL.
 * rk[i] is the rotation key of party ids[i]; crs = CRS[rotidx]. */
void ora_rotate(ora_ks* ks, int level, uint64_t galEl, int n, const int* ids,
    const uint64_t* ct_in, int in_limbs, const uint64_t* const* hoist,
    const uint64_t* const* rk, const uint64_t* crs, uint64_t* ct_out);
/* Conjugate (keyswitch.go:302-332) */
void ora_conjugate(ora_ks* ks, int level, uint64_t galEl, int n, const int* ids,
    const uint64_t* ct_in, int in_limbs, const uint64_t* const* ck, const uint64_t* crs, uint64_t* ct_out);

/* mkckks.Evaluator.Rescale scale loop (evaluator.go:376-384): returns nbRescales, updates *scale. */
int  ora_ckks_nb_rescales(const ora_ring* rq, int level, double* scale, double min_scale);

#ifdef __cplusplus
}
#endif
#endif
