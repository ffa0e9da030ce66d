/* ora_ring.h -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C restatement of the lattigo v2.3.0 `ring` semantics that the
 * reference SNUCP/MKHE-KKLSS calls limb-by-limb (go.mod:7; source not vendored,
 * restated from the published algorithm), plus the
 * in-repo fork mkrlwe/basis_extension.go.
 *
 * PARITY UNPINNED vs the Go reference: the reference holds no golden vectors
 *  and no Go toolchain exists here.  The oracle is pinned instead
 * against (i) an independent Python big-integer model (oracle/pymodel.py),
 * (ii) the reference's own property tests replayed with seeded inputs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.
 */
#ifndef ORA_RING_H
#define ORA_RING_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned __int128 u128;

/* ---- scalar arithmetic (lattigo ring/modular_reduction.go semantics) ---- */
static inline uint64_t ora_mulhi(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) >> 64); }

/* MRed: x*y*2^-64 mod q, canonical [0,q). */
static inline uint64_t ora_mred(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv) {
    u128 m = (u128)x * y;
    uint64_t mhi = (uint64_t)(m >> 64), mlo = (uint64_t)m;
    uint64_t hhi = ora_mulhi(mlo * qinv, q);
    uint64_t r = mhi - hhi + q;
    if (r >= q) r -= q;
    return r;
}
/* MRedConstant: same, result in [0,2q). */
static inline uint64_t ora_mred_lazy(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv) {
    u128 m = (u128)x * y;
    uint64_t mhi = (uint64_t)(m >> 64), mlo = (uint64_t)m;
    uint64_t hhi = ora_mulhi(mlo * qinv, q);
    return mhi - hhi + q;
}
/* MForm: a*2^64 mod q (Barrett with u = floor(2^128/q) = (uhi,ulo)). */
static inline uint64_t ora_mform(uint64_t a, uint64_t q, uint64_t uhi, uint64_t ulo) {
    uint64_t mhi = ora_mulhi(a, ulo);
    uint64_t r = (uint64_t)(0 - (a * uhi + mhi)) * q;
    if (r >= q) r -= q;
    return r;
}
/* InvMForm: a*2^-64 mod q. */
static inline uint64_t ora_invmform(uint64_t a, uint64_t q, uint64_t qinv) {
    uint64_t r = ora_mulhi(a * qinv, q);
    r = q - r;
    if (r >= q) r -= q;
    return r;
}
/* BRedAdd: a mod q for any 64-bit a. */
static inline uint64_t ora_bred_add(uint64_t a, uint64_t q, uint64_t uhi) {
    uint64_t mhi = ora_mulhi(a, uhi);
    uint64_t r = a - mhi * q;
    if (r >= q) r -= q;
    return r;
}
static inline uint64_t ora_cred(uint64_t a, uint64_t q) { return a >= q ? a - q : a; }

uint64_t ora_mulmod(uint64_t a, uint64_t b, uint64_t q);     /* plain a*b mod q  */
uint64_t ora_powmod(uint64_t x, uint64_t e, uint64_t q);     /* ring.ModExp      */
uint64_t ora_mredparams(uint64_t q);                         /* q^-1 mod 2^64    */
void     ora_bredparams(uint64_t q, uint64_t* uhi, uint64_t* ulo);
uint64_t ora_primitive_root(uint64_t q);                     /* lattigo rule, g starts at 3 */

/* ---- ring = degree + list of moduli with NTT tables ---- */
#define ORA_MAXMOD 96
typedef struct ora_ring {
    int logN, N, nmod;
    uint64_t mod[ORA_MAXMOD], qinv[ORA_MAXMOD], uhi[ORA_MAXMOD], ulo[ORA_MAXMOD];
    uint64_t psi_plain[ORA_MAXMOD];          /* the 2N-th root actually used      */
    uint64_t ninv[ORA_MAXMOD];               /* MForm(N^-1)  (NttNInv)            */
    uint64_t* psi[ORA_MAXMOD];               /* NttPsi    : psi^bitrev(j) * 2^64  */
    uint64_t* psiinv[ORA_MAXMOD];            /* NttPsiInv                          */
    uint64_t* rescale[ORA_MAXMOD];           /* rescale[L-1][i] = MForm(qL^-1 mod qi) */
} ora_ring;

ora_ring* ora_ring_new(int logN, const uint64_t* moduli, int nmod, const uint64_t* psi_or_null);
void      ora_ring_free(ora_ring* r);
int       ora_ring_n(const ora_ring* r);
uint64_t  ora_ring_psi(const ora_ring* r, int i);
uint64_t  ora_ring_qinv(const ora_ring* r, int i);
const uint64_t* ora_ring_psi_table(const ora_ring* r, int i, int inverse);

/* limb-level ops, modulus index i (lattigo ring.NTT / InvNTT / InvNTTLazy ...) */
void ora_ntt(const ora_ring* r, int i, const uint64_t* in, uint64_t* out);
void ora_intt(const ora_ring* r, int i, const uint64_t* in, uint64_t* out);
void ora_intt_lazy(const ora_ring* r, int i, const uint64_t* in, uint64_t* out);
void ora_limb_mform(const ora_ring* r, int i, const uint64_t* a, uint64_t* z);
void ora_limb_invmform(const ora_ring* r, int i, const uint64_t* a, uint64_t* z);
void ora_limb_mul(const ora_ring* r, int i, const uint64_t* a, const uint64_t* b, uint64_t* z);
void ora_limb_mul_add(const ora_ring* r, int i, const uint64_t* a, const uint64_t* b, uint64_t* z);
void ora_limb_mul_sub(const ora_ring* r, int i, const uint64_t* a, const uint64_t* b, uint64_t* z);
void ora_limb_add(const ora_ring* r, int i, const uint64_t* a, const uint64_t* b, uint64_t* z);
void ora_limb_sub(const ora_ring* r, int i, const uint64_t* a, const uint64_t* b, uint64_t* z);
void ora_limb_neg(const ora_ring* r, int i, const uint64_t* a, uint64_t* z);
void ora_limb_reduce(const ora_ring* r, int i, const uint64_t* a, uint64_t* z);
void ora_limb_mul_scalar(const ora_ring* r, int i, const uint64_t* a, uint64_t s, uint64_t* z);

/* poly-level (limb-major [level+1][N]) */
void ora_permute(const ora_ring* r, int level, uint64_t galEl, const uint64_t* in, uint64_t* out);
/* DivRoundByLastModulusManyLvl: in is mutated exactly like lattigo (last limbs get +h). */
void ora_div_round_last_many(const ora_ring* r, int level, int nb, uint64_t* in, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
