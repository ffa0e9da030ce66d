// poly_kernels.h -- coefficient-wise kernels of the multi-key key-switch path (gfx950).
//
// All of these are HBM-streaming kernels (one or a few modular products per 8-byte word):
// lanes walk consecutive coefficients of one limb (coalesced 8-B accesses, 512 B per wave
// instruction), blockIdx.y selects the limb so the modulus constants are wave-uniform
// (SGPRs), and reductions over gadget digits / parties are fused into one pass with the
// accumulator held in registers (read 2*terms words, write 1) instead of the reference's
// read-modify-write per term (mkrlwe/keyswitch_hoisted.go:28-30, 84-96).
#pragma once
#include "modarith.h"

namespace mkhe {

constexpr int MAX_TERMS = 40;   // >= 2*beta (BFV double gadget) and >= #parties
constexpr int MAXP = 8;         // special primes handled by the ModDown kernel

// out[o][m][n] = (mform?) sum_t a_t[o][m][n] (*) b_t[o][m][n]      (MulCoeffsMontgomeryAndAdd chain)
struct InnerProductArgs {
    const u64* a[MAX_TERMS];
    const u64* b[MAX_TERMS];
    u64* out;
    const Mod* mods;
    const int* map;          // [nslots] active limb -> limb index in a PolyQP-shaped buffer
    long term_outer;         // word stride between outer items inside every term buffer
    long out_outer;
    int nterms, nslots, nouter, N;
    int mform_out;           // apply MFormLvl to the sum (keyswitch_hoisted.go:94-96,115-117)
    const u64* addend;       // optional running sum (canonical, laid out like out; may be out itself): the products are added to it --
                             // MulCoeffsMontgomeryAndAddLvl onto a pool vector, mkbfv/keyswitch.go:160-189
};
void launch_inner_product(const InnerProductArgs& a, hipStream_t st);
// The same sums for up to IPB_MAX_BATCH independent inputs in ONE launch (batch.hip): the keys a[t] are shared, every input has its own
// digit vectors b[input][t] and its own output; nterms <= IPB_MAX_TERMS.
constexpr int IPB_MAX_TERMS = 8, IPB_MAX_BATCH = 16;
struct InnerProductBatchArgs {
    const u64* a[IPB_MAX_TERMS];
    const u64* b[IPB_MAX_BATCH][IPB_MAX_TERMS];
    u64* out[IPB_MAX_BATCH];
    const Mod* mods;
    const int* map;
    long term_outer, out_outer;
    int nterms, nslots, nouter, N, nbatch, mform_out;
};
void launch_inner_product_batch(const InnerProductBatchArgs& a, hipStream_t st);

// ModDownQPtoQ (mkrlwe/basis_extension.go:192-232 = lattigo Baseconverter.ModDownQPtoQ) on
// coefficient-domain lazy inputs, optionally accumulated into dst (ringQ.AddLvl).
struct ModDownTables {
    const u64* qoverqiinvqi;   // [np]            (P/p_i)^-1 * R mod p_i
    const u64* qoverqimodp;    // [nq][np]        (P/p_i) * R mod q_j
    const u64* vtimesqmodp;    // [nq][np+1]      v * (q_j - P mod q_j) mod q_j
    const u64* downparam;      // [nq]            q_j - (P^-1 * R mod q_j)
};
struct ModDownArgs {
    const u64* xq;             // [.. level+1 ..][N] Q part  (lazy, < 2q)
    const u64* xp;             // [np][N]            P part  (lazy, < 2p)
    u64* dst;                  // [level+1][N]
    const Mod* mods_q;         // [nq]
    const Mod* mods_p;         // [np]
    ModDownTables t;
    int level, np, N;
    int accumulate;            // dst = CRed(dst + result)
    int nbatch;                // independent (xq, xp, dst) triples
    long xq_batch, xp_batch, dst_batch;
};
void launch_moddown(const ModDownArgs& a, hipStream_t st);

// Batched ExternalProduct front half over independent items (parties):
//   c1[item][m][n] = sum_{i<beta} bg_item[i][m][n] (*) ah_item[i][m][n]        (keyswitch_hoisted.go:25-31)
constexpr int EXT_MAX_ITEMS = 64;
struct ExtInnerArgs {
    const u64* ah[EXT_MAX_ITEMS];
    const u64* bg[EXT_MAX_ITEMS];
    const u64* ah2[EXT_MAX_ITEMS];   // second gadget (BFV: QMul digits, keyswitch_hoisted.go:20-28) or all NULL
    const u64* bg2[EXT_MAX_ITEMS];
    unsigned char bg_once[EXT_MAX_ITEMS];  // this item's key is read by no other item of the launch: stream it past the caches
    unsigned char pair[EXT_MAX_ITEMS];   // (3: the item's products exist already at bg[item] [mtot][N]: copied into its c1 slot; 2 also: nothing to do)
                                         // 1: this item and the next one share `ah` (step F: <h(t_i), v_i> and <h(t_i), u>):
                                         // computed together, the digits are read once; 2: the follower (skipped); 0: single
    // groups (set by launch_ext_inner): single items that share `bg` (step E: every <h(c1_j), x>; F1: every <h(c0_i), y>) and pairs that
    // share their second key (step F2: the CRS u) are computed by ONE thread, up to four at a time, so that the shared operand is
    // loaded once per coefficient instead of once per item.  grp: 0 = on its own, 1 = leader, 2 = member (skipped); gnext: next
    // member's index, 255 = end of the chain
    unsigned char grp[EXT_MAX_ITEMS];
    unsigned char gnext[EXT_MAX_ITEMS];
    // Optional by-product of a group of single items (MulAndRelin step F1, where ah = h(c0_i)): xout[d] = sum_i xkey[i][d] (.) ah_i[d]
    // for every digit d -- the x of keyswitch_hoisted.go:79-96 (MForm'd when xmform) from the digits the thread holds anyway, instead of a
    // second pass over h(c0_i) by the inner-product kernel.  Only when ALL items of the launch form one group (nitems <= 4).
    const u64* xkey[EXT_MAX_ITEMS];
    u64* xout;
    int xmform;
    int xmulti;              // several groups, each with its own x (batch.hip: one group per input): xkey2[i] holds the xout of item i's group, no second gadget
    const u64* xkey2[EXT_MAX_ITEMS];     // the same for the second gadget (mkbfv: x2 = sum_i d2_i (.) h2(c0_i), keyswitch_hoisted.go:76-101)
    u64* xout2;
    u64* c1;                 // [nitems][mtot][N]
    const Mod* mods;
    const int* map;
    long digit_stride;       // mtot*N
    long c1_item;            // mtot*N
    int nitems, nb, nslots, N;
};
void launch_ext_inner(const ExtInnerArgs& a, hipStream_t st);

// Step F1 with x AND y computed in the thread (ext_inner_xy_kernel<G>, poly_kernels.hip): item g's digits ah[g] = h(c0_g) meet y = MForm(sum_j
// ykey[j] (.) yh[j]) for the product (c1 item g) and xkey[g] = d_g for x (xout, MForm'd).  ykey = the b_j, yh = the h(c1_j), g of each.
struct ExtXyArgs {
    const u64* ah[4]; const u64* xkey[4]; const u64* ykey[4]; const u64* yh[4];
    const u64* ah2[4]; const u64* xkey2[4]; const u64* ykey2[4]; const u64* yh2[4];      // second gadget (mkbfv: the QMul digits and keys) or ah2[0] = NULL
    u64* xout;               // x[d] stored (NULL: not needed, see e_out)
    u64* xout2;              // ... of the second gadget
    u64* e_out;              // non-NULL: step E as well -- <h(c1_j), x> for the g parties of op1, from the x[d] and h(c1_j)[d] the thread holds: g more
                             // products [mtot][N] at e_out + j * c1_item (the c1 slots of the E items of the tail batch), and neither x nor a second
                             // pass over the h(c1_j) exists
    u64* c1;                 // [g][mtot][N]
    const Mod* mods;
    const int* map;
    long digit_stride, c1_item;
    int g, g1, nb, nslots, N;        // g items (parties of op0), g1 terms of y / step-E products (parties of op1), one to four each
};
void launch_ext_inner_xy(const ExtXyArgs& a, hipStream_t st);
// ... for five to eight parties per operand (ext_inner_xy_wide_kernel<G, E>: the loads of a digit in chunks of four)
struct ExtXyWideArgs {
    const u64* ah[8]; const u64* xkey[8]; const u64* ykey[8]; const u64* yh[8];
    u64* xout;               // NULL with e_out
    u64* e_out;
    u64* c1;
    const Mod* mods;
    const int* map;
    long digit_stride, c1_item;
    int g, nb, nslots, N;
};
void launch_ext_inner_xy_wide(const ExtXyWideArgs& a, hipStream_t st);
// ... and for B inputs in lock step (mul_relin_batch): input b's digits ah[b][g] / yh[b][g] meet the shared keys; its x goes to xout[b], its products
// to the c1 items b * g .. b * g + g - 1 of the launch
constexpr int XYB_MAX = 16;
struct ExtXyBatchArgs {
    const u64* ah[XYB_MAX][4]; const u64* yh[XYB_MAX][4]; u64* xout[XYB_MAX];
    u64* eout[XYB_MAX];      // eout[0] non-NULL: step E too (input b's g1 products [mtot][N] at eout[b] + j * c1_item; xout is then not written)
    const u64* xkey[4]; const u64* ykey[4];
    u64* c1;                 // [nbatch * g][mtot][N]
    const Mod* mods;
    const int* map;
    long digit_stride, c1_item;
    int g, g1, nbatch, nb, nslots, N;      // g items per input (parties of op0), g1 terms of y (parties of op1)
};
void launch_ext_inner_xy_batch(const ExtXyBatchArgs& a, hipStream_t st);

// Batched ModDown tail: item b reads c1[b] and writes / accumulates into dst[b].  Items that share a
// destination are applied one after the other by the same thread (out_0 += sum_i ..., step F).
struct ModDownBatchArgs {
    const u64* c1;           // [nitems][mtot][N]  (Q part then P part, lazy)
    u64* dst[EXT_MAX_ITEMS];
    int accumulate[EXT_MAX_ITEMS];
    const Mod* mods_q;
    const Mod* mods_p;
    ModDownTables t;
    long c1_item;            // mtot*N
    long p_offset;           // nq*N
    int nitems, level, np, N;
    const int* qlist;        // limb-sharded evaluation: the Q limbs to produce (device list) or NULL = 0..level
    int nqlist;
    // items are processed group by group in parallel (blockIdx.z = group); the items of one group (same destination)
    // one after the other by the same thread.  group g = items order[gstart[g] .. gstart[g+1])
    unsigned char order[EXT_MAX_ITEMS];
    unsigned char gstart[EXT_MAX_ITEMS + 1];
    int ngroups;
    // Rotate tail fused in (keyswitch.go:251-296): addend[b] != NULL: an accumulating item adds onto addend[b] (read at the
    // un-permuted position, limb stride N) instead of onto its destination -- the c_0 of the input, no copy needed;
    // galEl != 0: every store goes to the signed-permuted position X -> X^galEl (0 with a sign flip is written as q, like
    // the reference's permutation loop), and later items of the group accumulate in that permuted image.
    const u64* addend[EXT_MAX_ITEMS];
    u64 galEl;
    int logN;
    // Context::rotate_multi: gal_v[b] != 0: item b's own Galois element (instead of galEl); post[b] != NULL: that polynomial (limb stride N) is added
    // to the finished destination at the stored position by the LAST item of the destination's group (AddNew(ct, RotateNew(ct, r)) in one pass)
    unsigned int gal_v[EXT_MAX_ITEMS];
    const u64* post[EXT_MAX_ITEMS];
};
void launch_moddown_batch(const ModDownBatchArgs& a, hipStream_t st);

// The same tail on MERGED items (Context::ext_batch, NttBatch::vi): ModDown is linear in the Q part and the sum the reference
// forms, sum_i ModDown(x_i) = sum_i (x_i - lift(xP_i)) * P^-1 mod q (every term canonical: MRed, then ring.Add), equals
// (sum_i x_i - sum_i lift(xP_i)) * P^-1 mod q.  A virtual item v is a set of <= MD_VI_MAX external products with ONE destination:
// the Q limbs of member 0 in c1 hold the lazy inverse NTT of the summed Q parts, every member's P limbs its own P part.  Per
// coefficient: y_k, v_k of every member literally (reconstructRNS, basis_extension.go:537-579), ONE 128-bit multSum over all
// members' y (the Montgomery fold of :623-645 is linear too) + sum_k vtimesqmodp[v_k], ONE MRed tail, ONE store.
constexpr int MD_VI_MAX = 4;
struct ModDownMergedArgs {
    const u64* c1;           // [nitems][mtot][N]
    u64* dst[EXT_MAX_ITEMS];                      // per virtual item
    const u64* addend[EXT_MAX_ITEMS];             // as ModDownBatchArgs::addend
    // (32-bit fields on purpose: indexed by a value that also indexes the 8-byte lists above, byte-sized lists made hipcc 7.2 fold
    // the index into the BASE of a scalar load, whose low address bits the hardware ignores -- wrong members were read)
    unsigned int accumulate[EXT_MAX_ITEMS];
    unsigned int cnt[EXT_MAX_ITEMS];              // members of virtual item v
    unsigned int mem[EXT_MAX_ITEMS];              // their item indices in c1, one byte each (member k: bits 8k .. 8k+7)
    const Mod* mods_q;
    const Mod* mods_p;
    ModDownTables t;
    long c1_item, p_offset;
    int nvi, level, np, N;
    // virtual items with the same destination are applied one after the other by the same thread, different destinations in
    // parallel (blockIdx.z), exactly like ModDownBatchArgs
    unsigned char order[EXT_MAX_ITEMS];
    unsigned char gstart[EXT_MAX_ITEMS + 1];
    int ngroups;
    u64 galEl;
    int logN;
    // Rescale folded into the store (round 3; mkckks.Evaluator.mulRelinHoisted always rescales right after, evaluator.go:558-581): when
    // rescale_row != NULL every destination is written exactly once by this launch (no accumulate, no permutation), and what is stored is
    // DivRoundByLastModulus of the ModDown result -- limbs 0 .. level-1 of rdst[v] (a polynomial of the output ciphertext one level down,
    // limb stride N) -- instead of the result itself; the thread computes limb `level` of its coefficient first (every limb slice does).
    const u64* rescale_row;                       // [level]: RescaleParams row of the level, as div_round_last_kernel takes it
    const u64* rescale_h;                         // [level]: (q_level - 1) / 2 mod q_j -- BRedAdd(h, q_j) of the same formula, from the host
    u64* rdst[EXT_MAX_ITEMS];
    unsigned int gal_v[EXT_MAX_ITEMS];            // as ModDownBatchArgs::gal_v / post, per virtual item
    const u64* post[EXT_MAX_ITEMS];
};
void launch_moddown_merged(const ModDownMergedArgs& a, hipStream_t st);

// Tensor step D of MulAndRelin (keyswitch_hoisted.go:120-140) on NTT-domain inputs.
//   out_0 = a0*b0 ; out_o = b0*a_o (o in ids0) (+)= a0*b_o (o in ids1)
// NTT(c0_i) / NTT(c1_j) of the party components are read either from a plain NTT buffer (limb stride N)
// or straight from the hoisted digits: for alpha = 1 digit l under its own modulus l IS NTT_l(c limb l),
// i.e. the diagonal h[l][l] (limb stride (nQ+nP+1)*N) -- those NTTs need not be recomputed.
struct TensorArgs {
    const u64* a0;             // [L][N]  NTT(c0_0)
    const u64* b0;             // [L][N]  NTT(c1_0)
    u64* out;                  // [1+nout][L][N]
    const Mod* mods;
    const u64* a[33];          // per out slot o>=1: NTT(c0_o) base or NULL
    const u64* b[33];          // per out slot o>=1: NTT(c1_o) base or NULL
    long a_ls[33], b_ls[33];   // limb strides of the above
    int nout, L, N;
    int with_c0;               // 0: leave c0_0*c1_0 out of out_0 (party-sharded evaluation)
    const int* limbs;          // limb-sharded evaluation: the limbs to produce (device list of nlimbs entries) or NULL = all L
    int nlimbs;
    const int* map;            // [L] modulus index of limb l (NULL: l itself)          -- ring R of mkbfv
    const u64* scale;          // [L] Montgomery constants multiplied into every output (NULL: none)
                               //     -- ringR.MulScalar(t) of Quantize, mkbfv/basis_extension.go:71
    int nbatch;                // > 1 (batch.hip): that many independent products in one launch, input / output pointers of product k at
    long in_batch, out_batch;  //     + k * in_batch / + k * out_batch words
};
void launch_tensor(const TensorArgs& a, hipStream_t st);

// Generic fast basis conversion between two RNS bases S (ns limbs) and T (nt limbs) -- the reference's
// modUpExact (mkrlwe/basis_extension.go:337-357, reconstructRNS :537-585, multSum :587-646) restated
// literally per coefficient, with the optional ModDown tail (:192-232 / :292-334):
//   raw  (downparam == NULL): dst_j = lazy lift of src to modulus t_j          (ModUpQtoP / ModUpPtoQ)
//   tail (downparam != NULL): dst_j = MRed(lift_j + 2 t_j - xsub_j, downparam_j) (ModDownQPtoQ / QPtoP)
// prescale: per-source-limb Montgomery constant applied first (mkbfv Rescale, basis_extension.go:87).
// copy_dst: also store the (unscaled) source limbs there (ModUpQtoR keeps the Q part, :56-58).
constexpr int BC_MAXS = 16;
struct BasisConvTables {
    const u64* qoverqiinvqi;   // [ns]
    const u64* qoverqimodp;    // [nt][ns]
    const u64* vtimesqmodp;    // [nt][ns+1]
};
struct BasisConvArgs {
    const u64* src;  long src_poly;      // [npolys][.. ns ..][N]
    const u64* xsub; long xsub_poly;     // tail minuend [npolys][.. nt ..][N] or NULL (= 0)
    u64* dst;        long dst_poly;      // [npolys][.. nt ..][N]
    u64* copy_dst;   long copy_poly;
    const Mod* mods_s;
    const Mod* mods_t;
    const u64* prescale;
    const u64* downparam;
    BasisConvTables t;
    int ns, nt, N, npolys;
};
void launch_basis_conv(const BasisConvArgs& a, hipStream_t st);

// Gadget digit spread for alpha >= 2 (Decomposer.DecomposeAndSplit, mkrlwe/basis_extension.go:428-535): digit d of a
// polynomial = its limbs [alpha*d, alpha*d + nd), CRT-reconstructed (float64 correction index, literal) and written as a
// lazy residue under every active modulus of Q and P into a SwitchingKey-shaped buffer [digit][nQ+nP][N] (coefficient
// domain; the forward NTT follows in place).  nd = 1 is the copy path (:443-451).
constexpr int DEC_MAXA = 4;
constexpr int DEC_MAX_ITEMS = 64;
struct DecompSpreadArgs {
    const u64* src[DEC_MAX_ITEMS];     // polynomials [.. level+1 ..][N]
    u64* dst[DEC_MAX_ITEMS];           // [beta][mtot][N]
    const Mod* mods;                   // Q then P
    const int* map;                    // [nslots] active slot -> modulus index (= limb position in dst)
    const u64* ta;                     // [ndig][alpha-1][DEC_MAXA]            qoverqiinvqi
    const u64* tb;                     // [ndig][alpha-1][mtot][DEC_MAXA]      qoverqimodp
    const u64* tc;                     // [ndig][alpha-1][mtot][DEC_MAXA+1]    vtimesqmodp
    int nd[64];                        // limbs of digit d at this level (1 = copy)
    int alpha, ndigits, nslots, mtot, N, nitems;
    // first_stage (N = 2^16, whose forward NTT always runs split): each thread produces the coefficients n and n + N/2 and
    // applies the first Cooley-Tukey stage (twiddle psi[m][1]) before storing, exactly as ntt_split_fwd_kernel would -- the
    // NTT that follows (NttBatch::prestaged) then skips its streaming pass (one read + one write of every digit limb).
    // first_stage = 2 (round 3): the coefficients n + k N/4, k = 0..3, and the first TWO stages (psi[m][1]; psi[m][2], psi[m][3]): what is
    // left of the transform are four 2^14-point sub-transforms per limb, ONE pass of the H16 kernel each (NttBatch::prestaged = 2)
    const u64* psi;                    // forward twiddle tables [mtot][N], signed-split form
    int first_stage;
    // first_stage = 2 only (alpha = 2, every modulus < 2^57): constants of the all-unsigned radix-2^30 one-round product (poly_kernels.hip
    // decomp_spread4_kernel), each a pair (t 2^30 mod p, t 2^62 mod p) as two words of radix-2^30 digits (low digit in the low half)
    const u64* tb30;                   // [ndig][mtot][2][2]   qoverqimodp of the two-limb digit d under modulus m
    const u64* tw30;                   // [mtot][4][2]         entry 0: t = 1; entries 1..3: t = psi[m][1..3]
};
void launch_decomp_spread(const DecompSpreadArgs& a, hipStream_t st);

// elementwise ring ops on ciphertext polynomials: dst = a + b / a - b / -a per limb (canonical inputs)
// (ring.Add / Sub / Neg as used by mkckks/evaluator.go:41-300 and mkbfv/evaluator.go:40-76)
void launch_sub(u64* dst, const u64* a, const u64* b, const Mod* mods, int L, int N, hipStream_t st);
void launch_neg(u64* dst, const u64* a, const Mod* mods, int L, int N, hipStream_t st);

// dst[p][l][n] = src[p][l][n] * c_l  (ring.MulScalar with c = MForm(scalar), canonical)
void launch_mul_const(u64* dst, const u64* src, const Mod* mods, const int* map, const u64* consts, int L, int N, int npolys,
                      long poly_stride, hipStream_t st);

// mkckks MultByConst body (mkckks/evaluator.go:150-196): coefficients [0, N/2) of limb l times c[0][l], [N/2, N) times
// c[1][l] (Montgomery constants, canonical results); npolys polynomials, possibly different limb counts in / out.
struct MulConstArgs {
    const u64* src; u64* dst;
    const Mod* mods;
    long src_poly, dst_poly;
    int L, N, npolys;
    u64 c[2][48];
};
void launch_mul_const_halves(const MulConstArgs& a, hipStream_t st);
// dst[p][l][n] = a[p][l][n] * MForm(b[l][n])  (one polynomial b against npolys polynomials: MulPtxtNew, evaluator.go:471-478)
void launch_mul_by_poly(u64* dst, const u64* a, const u64* b, const Mod* mods, int L, int N, int npolys, hipStream_t st);

// all components of a ciphertext Add / Sub in ONE launch (mkckks/evaluator.go:200-304 evaluateInPlace, mkbfv/evaluator.go:27-76):
// per component mode 0: a + b, 1: a - b, 2: copy a, 3: copy b, 4: -b (q - b, 0 -> q like ring.Neg)
constexpr int CTBIN_MAX = 65;
struct CtBinArgs {
    const u64* a[CTBIN_MAX]; const u64* b[CTBIN_MAX]; u64* dst[CTBIN_MAX];
    unsigned char mode[CTBIN_MAX];
    const Mod* mods;
    int L, N, ncomp;
};
void launch_ct_binary(const CtBinArgs& a, hipStream_t st);

// dst = in[0] + in[1] + ... + in[n-1] over whole ciphertexts of ONE shape ([npolys][L][N] each, dst may be one of them): the chain
// "out = AddNew(out, temp)" of cnn.Convolution / FC1Layer (cnn/cnn.go:19-30,58-62) in one launch; every partial sum canonical (csub), so the
// order is immaterial
constexpr int CTSUM_MAX = 16;
struct CtSumArgs {
    const u64* in[CTSUM_MAX];
    u64* dst;
    const Mod* mods;
    int n, L, N, npolys;
};
void launch_ct_sum(const CtSumArgs& a, hipStream_t st);

// dst = CRed(a + b) per limb
void launch_add(u64* dst, const u64* a, const u64* b, const Mod* mods, int L, int N, hipStream_t st);

// Coefficient-domain automorphism with sign (keyswitch.go:267-296): out[(i*g) mod N] = +-in[i]
void launch_automorphism(u64* dst, const u64* src, const Mod* mods, int L, int logN, u64 galEl, int npolys, hipStream_t st);

// DivRoundByLastModulus (lattigo ring_scaling.go; mkckks/evaluator.go:388), one division step:
// dst[i] (i < level) from src limbs 0..level;  src is not modified.
void launch_div_round_last(u64* dst, const u64* src, const Mod* mods, const u64* rescale_row /*[level]*/,
                           int level, int N, int npolys, long src_poly, long dst_poly, hipStream_t st);

// the same step on a contiguous source [ngroups * per_group polys] whose results go to ngroups separate destinations (the ciphertexts of a batch,
// batch.hip): polynomial p of group g -> dst[g] + p * dst_poly
constexpr int DRL_MAX = 64;
struct DivRoundListArgs {
    u64* dst[DRL_MAX];
    const u64* src;
    const Mod* mods;
    const u64* rescale_row;
    long src_poly, dst_poly;
    int level, N, per_group, ngroups;
};
void launch_div_round_last_list(const DivRoundListArgs& a, hipStream_t st);

// In-place fold of sums of canonical residues (< 2^63) back to [0,q), optionally to Montgomery form.
struct FoldArgs {
    u64* buf;
    const Mod* mods;
    const int* map;            // [nslots] limb index inside each poly
    long poly_stride;
    int nslots, npolys, N, mform;
};
void launch_fold(const FoldArgs& a, hipStream_t st);
// fold of a limb range of a switching-key-shaped buffer whose summands are `npieces` separate pieces (piece p, limb i of the range at
// pieces + p * piece_stride + i * N) into dst + i * N; active: bit m = modulus m is active at the level
struct FoldPiecesArgs {
    const u64* pieces;
    u64* dst;
    const Mod* mods;
    long piece_stride, first_limb;
    unsigned long long active;
    int npieces, nlimbs, mtot, ndigits, N, mform;
};
void launch_fold_pieces(const FoldPiecesArgs& a, hipStream_t st);

// z = MForm(a) / z = mont_mul(a, b) helpers on limb-major buffers
void launch_mform(u64* dst, const u64* src, const Mod* mods, const int* map, int nslots, int N, hipStream_t st);

}  // namespace mkhe
