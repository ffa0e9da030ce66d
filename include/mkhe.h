/* mkhe.h -- C ABI of the MI355X multi-key RLWE key-switch engine (libmkhe_hip.so).
 *
 * Drop-in boundary for the hot path of SNUCP/MKHE-KKLSS (pure Go + lattigo v2.3.0).  The
 * reference has no FFI; the boundary is the method set of mkrlwe.KeySwitcher plus the lattigo
 * helpers it calls.  Every entry point cites the reference interface it
 * replaces as <file>:<line> relative to the reference root.  The cgo stubs a maintainer would
 * add are shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C: opaque handles, pointers and sizes only.  All polynomial data is uint64, limb-major.
 *  - SwitchingKey / hoisted digit vector : uint64[betaMax][nQ+nP][N]  (Q limbs then P limbs),
 *    stored exactly as the Go side stores it: NTT domain, Montgomery form for keys and CRS
 *    (mkrlwe/params.go:56, keygen.go:299-300), NTT domain non-Montgomery for hoisted digits.
 *  - Ciphertext : uint64[1+n][limbs][N], slot 0 = Value["0"], slot 1+i = Value[ids[i]]
 *    (mkrlwe/elements.go:17-19); coefficient domain.  level = limbs-1.
 *  - Party ids are arbitrary ints chosen by the caller (the shim maps Go's string ids).
 *  - Return value 0 = ok; non-zero = error, text via mkhe_last_error() (the reference panics
 *    at the same sites: keyswitch.go:126-132, keys.go:151-162,190-198; the shim re-panics).
 *  - One context per Evaluator, calls on a context serialized by the caller (the reference is
 *    not reentrant either: shared pools keyswitch.go:12-15).  All work is enqueued on the
 *    context's HIP stream; *_download and mkhe_ctx_sync synchronize.
 *
 * Environment (the COMPLETE list of variables libmkhe_hip.so reads; each once per process)
 *  - MKHE_NTT32   = 0 | 1 | 2   forward NTT at N = 2^15 where two kernels apply (same bits): two-pass, single-pass, or (default 2)
 *                               whichever a measurement inside the caller's workload finds faster; mkhe_ctx_set_ntt_choice pins it per context
 *  - MKHE_POOL_GB = <GiB>       device-wide bound of the buffer pools that recycle freed handles (default 32; fractions allowed; 0 = keep nothing)
 * Every other MKHE_* variable named in DESIGN.md section 6 is A/B instrumentation and exists only in the diagnostic build
 * libmkhe_hip_switches.so (csrc/switches.h, `make switches`); the product library ignores them.
 */
#ifndef MKHE_H
#define MKHE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mkhe_ctx mkhe_ctx;
typedef struct mkhe_swk mkhe_swk;
typedef struct mkhe_ct  mkhe_ct;

const char* mkhe_last_error(void);
int mkhe_device_count(void);

/* ---- context: mkrlwe.NewKeySwitcher keyswitch.go:33-47 (+ NewDecomposer basis_extension.go:368,
 *      lattigo rlwe.NewKeySwitcher / ring.NewRing tables).  psiQ/psiP: optional primitive 2N-th
 *      roots (plain) per modulus, e.g. InvMForm(ring.NttPsi[i][N/2]); NULL = lattigo's own rule.
 *      Moduli: distinct primes < 2^60 with q = 1 mod 2N (every prime of the reference's parameter sets is <= 60 bits
 *      and < 2^60; the lazy butterflies need 4q < 2^62); logN in [10, 16]. */
int  mkhe_ctx_create(mkhe_ctx** out, int logN, const uint64_t* Q, int nQ, const uint64_t* P, int nP,
                     int gamma, const uint64_t* psiQ, const uint64_t* psiP, int device);
void mkhe_ctx_destroy(mkhe_ctx* ctx);
int  mkhe_ctx_sync(mkhe_ctx* ctx);
/* Several contexts over the same ring on one device (no reference counterpart: the Go evaluator is single-threaded): each
 * has its own stream and scratch pools; keys, CRS, hoisted forms and ciphertexts are plain device memory behind their
 * handles and may be used through any of them.  Independent operations issued through different contexts overlap on the
 * GPU.  mkhe_ctx_wait_for orders them without a host synchronisation: work enqueued on ctx after the call starts only
 * after everything enqueued on other before the call has finished.  A handle must be destroyed through the context that
 * created it, after every other context that used it has been waited for. */
int  mkhe_ctx_wait_for(mkhe_ctx* ctx, mkhe_ctx* other);
/* Capture of a fixed sequence of engine calls into a HIP graph (no reference counterpart).  Between mkhe_capture_begin and
 * mkhe_capture_end every call on ctx -- and on contexts ordered with mkhe_ctx_wait_for after the begin and joined back
 * with mkhe_ctx_wait_for(ctx, other) before the end -- is recorded instead of executed; mkhe_graph_launch replays the
 * whole sequence with one submission (launch-bound circuits of many small kernels: the cnn caller).  Rules: no upload /
 * download / sync / key generation inside a capture; the replay reads and writes exactly the device buffers the captured
 * calls used, so the handles created inside the capture must stay alive (their buffers are the graph's temporaries and
 * outputs) and new inputs are uploaded into the SAME input handles. */
typedef struct mkhe_graph mkhe_graph;
int  mkhe_capture_begin(mkhe_ctx* ctx);
int  mkhe_capture_end(mkhe_ctx* ctx, mkhe_graph** out);
int  mkhe_graph_launch(mkhe_ctx* ctx, mkhe_graph* graph);
void mkhe_graph_destroy(mkhe_graph* graph);
int  mkhe_ctx_alpha(const mkhe_ctx* ctx);                 /* Parameters.Alpha  params.go:63-65 */
int  mkhe_ctx_beta(const mkhe_ctx* ctx, int level);       /* Parameters.Beta   params.go:67-71 */
int  mkhe_ctx_n(const mkhe_ctx* ctx);
size_t mkhe_ctx_swk_words(const mkhe_ctx* ctx);           /* betaMax*(nQ+nP)*N */
uint64_t mkhe_ctx_psi(const mkhe_ctx* ctx, int modulus_index);   /* root in use (Q then P) */
void* mkhe_ctx_stream(mkhe_ctx* ctx);                     /* hipStream_t, for event timing */

/* ---- SwitchingKey handles: mkrlwe.SwitchingKey keys.go:23-25, NewSwitchingKey keys.go:245-255 */
int  mkhe_swk_create(mkhe_ctx* ctx, mkhe_swk** out);
/* same without the zero fill: for keys / hoisted forms that the next engine call writes (mkhe_hoisted_form, mkhe_decompose, key generation) */
int  mkhe_swk_create_uninit(mkhe_ctx* ctx, mkhe_swk** out);
void mkhe_swk_destroy(mkhe_ctx* ctx, mkhe_swk* swk);
int  mkhe_swk_upload(mkhe_ctx* ctx, mkhe_swk* swk, const uint64_t* host);
/* Go's []rlwe.PolyQP: one pointer per limb, order [digit][Q limbs..., P limbs...] */
int  mkhe_swk_upload_limbs(mkhe_ctx* ctx, mkhe_swk* swk, const uint64_t* const* limbs, int ndigits);
int  mkhe_swk_download(mkhe_ctx* ctx, const mkhe_swk* swk, uint64_t* host);
void* mkhe_swk_devptr(mkhe_swk* swk);

/* ---- Ciphertext handles: mkrlwe.Ciphertext elements.go:17-33 */
int  mkhe_ct_create(mkhe_ctx* ctx, int n, const int* ids, int limbs, mkhe_ct** out);
/* same without the zero fill: for ciphertexts that are the output of the next engine call (every entry point writes all
 * limbs of its ctOut), e.g. the result of Evaluator.MulRelinNew */
int  mkhe_ct_create_uninit(mkhe_ctx* ctx, int n, const int* ids, int limbs, mkhe_ct** out);
void mkhe_ct_destroy(mkhe_ctx* ctx, mkhe_ct* ct);
int  mkhe_ct_upload(mkhe_ctx* ctx, mkhe_ct* ct, const uint64_t* host);
int  mkhe_ct_upload_poly_limbs(mkhe_ctx* ctx, mkhe_ct* ct, int slot, const uint64_t* const* limbs);
int  mkhe_ct_download(mkhe_ctx* ctx, const mkhe_ct* ct, uint64_t* host);
int  mkhe_ct_download_poly_limbs(mkhe_ctx* ctx, const mkhe_ct* ct, int slot, uint64_t* const* limbs);
/* device-to-device copy of a ciphertext with the same ids and number of limbs (rlwe Ciphertext.CopyNew; rotation by 0) */
int  mkhe_ct_copy(mkhe_ctx* ctx, const mkhe_ct* in, mkhe_ct* out);
int  mkhe_ct_limbs(const mkhe_ct* ct);
int  mkhe_ct_nparties(const mkhe_ct* ct);
void* mkhe_ct_devptr(mkhe_ct* ct);

/* ---- raw device buffers of uint64 words (callers that keep polynomials resident themselves) */
int  mkhe_buf_alloc(mkhe_ctx* ctx, size_t words, void** dev_out);
void mkhe_buf_free(mkhe_ctx* ctx, void* dev);
int  mkhe_buf_upload(mkhe_ctx* ctx, void* dev, const uint64_t* host, size_t words);
int  mkhe_buf_download(mkhe_ctx* ctx, const void* dev, uint64_t* host, size_t words);

/* ---- lattigo ring.NTTLvl / InvNTTLvl / InvNTTLazyLvl on a raw device buffer [count][limbs][N];
 *      limb l uses modulus index mod_base+l (0..nQ-1 = Q, nQ.. = P).  keyswitch.go:29-30,58,114-115 */
int  mkhe_ntt(mkhe_ctx* ctx, const void* dev_src, void* dev_dst, int count, int limbs, int mod_base, int inverse, int lazy);

/* ---- KeySwitcher.Decompose keyswitch.go:49-73 (DecomposeSingleNTT :21-31, DecomposeAndSplit
 *      basis_extension.go:428-535).  Input poly = slot `slot` of ct; is_ntt mirrors ring.Poly.IsNTT. */
int  mkhe_decompose(mkhe_ctx* ctx, int level, int is_ntt, const mkhe_ct* ct, int slot, mkhe_swk* out);

/* ---- mkckks.Evaluator.HoistedForm evaluator.go:543-553: Decompose of every party component ct.Value[id] (slots 1..n, not
 *      slot 0) in one batched launch; out[i] receives h(ct.Value[ids[i]]). */
int  mkhe_hoisted_form(mkhe_ctx* ctx, int level, const mkhe_ct* ct, mkhe_swk* const* out);

/* ---- KeySwitcher.ExternalProduct keyswitch.go:79-118 / ExternalProductHoisted keyswitch_hoisted.go:10-40.
 *      Result (coefficient domain, canonical) is written to slot out_slot of out. */
int  mkhe_external_product(mkhe_ctx* ctx, int level, int is_ntt, const mkhe_ct* a, int slot,
                           const mkhe_swk* bg, mkhe_ct* out, int out_slot);
int  mkhe_external_product_hoisted(mkhe_ctx* ctx, int level, const mkhe_swk* a_hoisted,
                                   const mkhe_swk* bg, mkhe_ct* out, int out_slot);

/* ---- KeySwitcher.MulAndRelin keyswitch.go:122-230 / MulAndRelinHoisted keyswitch_hoisted.go:44-179.
 *      hoist0/hoist1: per-party hoisted forms aligned with op0/op1 ids, or NULL (engine hoists
 *      internally = mkckks.Evaluator.MulRelinNew evaluator.go:416-443).
 *      rlk_d0, rlk_v0 aligned with op0 ids (rlk.Value[1], Value[2]); rlk_b1 aligned with op1 ids
 *      (rlk.Value[0]) keys.go:34-37;  crs_u = params.CRS[-1] params.go:37.
 *      level is taken from out (keyswitch_hoisted.go:46); out ids must be the union. */
int  mkhe_mul_and_relin(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                        const mkhe_swk* const* hoist0, const mkhe_swk* const* hoist1,
                        const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_d0,
                        const mkhe_swk* const* rlk_v0, const mkhe_swk* crs_u, mkhe_ct* out);
/* mkckks.Evaluator.mulRelinHoisted (mkckks/evaluator.go:558-581): MulAndRelin[Hoisted] followed by ONE Rescale (the usual case: the scale of
 * the product drops below 2 * params.Scale() after one division), as one call.  `out` is the RESCALED ciphertext: one level below
 * min(level(op0), level(op1)), ids = the union.  On one device with at most four parties per operand the DivRoundByLastModulus rides on the
 * store of the last ModDown (the level-L product is never written); otherwise the engine runs mkhe_mul_and_relin into a pooled temporary and
 * mkhe_rescale after it.  The same ciphertext, bit for bit, as the two calls (tests/test_gpu_parity.py). */
int  mkhe_mul_relin_rescale(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                            const mkhe_swk* const* hoist0, const mkhe_swk* const* hoist1,
                            const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_d0,
                            const mkhe_swk* const* rlk_v0, const mkhe_swk* crs_u, mkhe_ct* out);

/* ---- the same MulAndRelinHoisted split in phases for party-sharded multi-GPU evaluation
 *      (the reference is single-process).  Each rank passes sub-ciphertexts holding c_0
 *      and the party components it owns; x_part / y_part receive the rank's canonical partial sums
 *      sum_i d_i (.) h(c0_i), sum_j b_j (.) h(c1_j) WITHOUT MForm (keyswitch_hoisted.go:79-92,99-113).
 *      The caller sums them over ranks as uint64 (RCCL all-reduce; exact while ranks*q < 2^63), calls
 *      mkhe_swk_fold(.., mform=1) (= MFormLvl of the total, :94-96,115-117) and then mkhe_mr_finish
 *      (steps E-F, :146-178; the tensor step D, :119-144, is started by mkhe_mr_partial and written into
 *      `out`).  with_c0 != 0 on exactly one rank adds c0_0*c1_0 to out_0; out_0 and any
 *      out_i whose two operand components live on different ranks are partial sums to be reduced and
 *      folded with mkhe_ct_fold. */
int  mkhe_mr_partial(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                     const mkhe_swk* const* hoist0, const mkhe_swk* const* hoist1,
                     const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_d0,
                     int with_c0, mkhe_ct* out, mkhe_swk* x_part, mkhe_swk* y_part);
int  mkhe_swk_fold(mkhe_ctx* ctx, mkhe_swk* swk, int level, int mform);
/* The reduction of x / y on a point-to-point mesh (xGMI: 7 links per GPU; SURVEY.md 8e(2) "prefer reduce-scatter + all-gather"): rank r receives slice
 * r of every rank's partial sum (one all-to-all), sums, folds and MForm's ITS slice -- this call: limbs [first_limb, first_limb + nlimbs) of a
 * switching-key buffer ([digit][modulus][N]: limb l = digit l / (nQ + nP), modulus l % (nQ + nP)), summand p of limb i at the device address
 * pieces + (p * piece_stride_words + i * N) words, result at dst + i * N words; limbs of inactive digits / moduli are skipped -- and the folded slices
 * are all-gathered.  Same integers as the all-reduce + mkhe_swk_fold (a sum of canonical residues, then MFormLvl: keyswitch_hoisted.go:94-96,115-117). */
int  mkhe_swk_fold_pieces(mkhe_ctx* ctx, const void* pieces, int npieces, long piece_stride_words, long first_limb, long nlimbs, int level, int mform, void* dst);
int  mkhe_mr_finish(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, const mkhe_swk* x, const mkhe_swk* y,
                    const mkhe_swk* const* rlk_v0, const mkhe_swk* crs_u, mkhe_ct* out);
/* mkhe_mr_finish in two halves, so that the all-reduce of x can still be in flight while the part that needs y alone runs:
 * head = t_i = <h(c0_i), y>_P for every party of op0 (keyswitch_hoisted.go:165-169) and the Decompose of the t_i (:171);
 * tail = out_j += <h(c1_j), x>_P (:146-154), then out_0 += <h(t_i), v_i>_P and out_i += <h(t_i), u>_P (:173-177).
 * mkhe_mr_finish == head; tail.  Both take the operands of the preceding mkhe_mr_partial. */
int  mkhe_mr_finish_head(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, const mkhe_swk* y, mkhe_ct* out);
int  mkhe_mr_finish_tail(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, const mkhe_swk* x,
                         const mkhe_swk* const* rlk_v0, const mkhe_swk* crs_u, mkhe_ct* out);
int  mkhe_ct_fold(mkhe_ctx* ctx, mkhe_ct* ct);

/* ---- limb-sharded multi-GPU MulAndRelin (no reference counterpart; mkhe_kklss_amd/dist.py LimbShardedMulRelin).
 *      A context may own a subset of the RNS moduli (indices over Q then P; n = 0: all).  With an ownership set the
 *      four phases below evaluate KeySwitcher.MulAndRelin (keyswitch.go:122-230) on full operands (all parties, every
 *      rank) for the owned moduli only: the per-party sums x, y are local; after each phase *words_out words of the
 *      caller's device buffer dev_stage (phase 4: of `out` itself) are all-reduced (sums of disjoint slices):
 *        1: tensor, hoisting, x, y, <h(c0_i), y> + InvNTT        -> P limbs of those products
 *        2: their ModDown = t_i on the owned limbs               -> t_i
 *        3: Decompose(t_i), <h(c1_j), x>, <h(t_i), v_i>, <h(t_i), u> + InvNTT  -> P limbs
 *        4: ModDown accumulated into out (owned limbs, zeros elsewhere)        -> out
 *      Keys: rlk_b1 / rlk_d0 for phase 1, rlk_v0 / crs_u for phase 3 (NULL otherwise). */
int  mkhe_ctx_set_owned(mkhe_ctx* ctx, const int* mod_idx, int n);
int  mkhe_lsh_phase(mkhe_ctx* ctx, int phase, const mkhe_ct* op0, const mkhe_ct* op1,
                    const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_d0, const mkhe_swk* const* rlk_v0,
                    const mkhe_swk* crs_u, mkhe_ct* out, void* dev_stage, size_t* words_out);

/* ---- KeySwitcher.Rotate keyswitch.go:234-298 / RotateHoisted keyswitch_hoisted.go:183-247.
 *      galEl = 5^rotidx mod 2N; rk aligned with ct ids (rkSet[id][rotidx]); crs = params.CRS[rotidx]. */
int  mkhe_rotate(mkhe_ctx* ctx, uint64_t galEl, const mkhe_ct* in, const mkhe_swk* const* hoist,
                 const mkhe_swk* const* rk, const mkhe_swk* crs, mkhe_ct* out);
/* ---- Rotate in two phases for a party-sharded evaluation: the key-switch part of keyswitch.go:251-265
 *      without the permutation (with_c0 = 0 leaves c_0 out of out_0: exactly one rank adds it), and the signed
 *      permutation :267-296 on its own.  mkhe_rotate == mkhe_rotate_partial(with_c0 = 1) + mkhe_ct_automorphism. */
int  mkhe_rotate_partial(mkhe_ctx* ctx, const mkhe_ct* in, const mkhe_swk* const* hoist,
                         const mkhe_swk* const* rk, const mkhe_swk* crs, int with_c0, mkhe_ct* out);
int  mkhe_ct_automorphism(mkhe_ctx* ctx, uint64_t galEl, const mkhe_ct* in, mkhe_ct* out);
/* ---- KeySwitcher.Conjugate keyswitch.go:302-332; galEl = 2N-1, crs = params.CRS[-2] */
int  mkhe_conjugate(mkhe_ctx* ctx, uint64_t galEl, const mkhe_ct* in, const mkhe_swk* const* ck,
                    const mkhe_swk* crs, mkhe_ct* out);

/* ---- body of mkckks.Evaluator.Rescale evaluator.go:385-391 = lattigo
 *      ring.DivRoundByLastModulusManyLvl on every poly; out has limbs(in)-nb limbs. */
int  mkhe_rescale(mkhe_ctx* ctx, const mkhe_ct* in, int nb, mkhe_ct* out);

/* ---- elementwise evaluator ops on resident ciphertexts ("next" row of SURVEY.md 8f):
 *      mkckks/evaluator.go:41-104 (evaluateInPlace, add, sub) and mkbfv/evaluator.go:27-76.  out carries the
 *      union of the id sets; components only one operand has are copied (Sub: negated when they come from op1). */
int  mkhe_ct_add(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, mkhe_ct* out);
int  mkhe_ct_sub(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, mkhe_ct* out);

/* mkckks.Evaluator.MultByConst body (mkckks/evaluator.go:150-196): per limb l < min(limbs(in), limbs(out)) the coefficients
 * [0, N/2) are multiplied by c_first[l], [N/2, N) by c_second[l] (host arrays, Montgomery form = the reference's
 * ring.MForm(scaledConst); the float64 getConstAndScale / scaleUpExact logic :40-94 stays on the host). */
int  mkhe_ct_mul_const(mkhe_ctx* ctx, const mkhe_ct* in, const uint64_t* c_first, const uint64_t* c_second, mkhe_ct* out);
/* out = in[0] + in[1] + ... + in[n-1] for n ciphertexts of one shape (same ids; out at its own level <= theirs, may be one of them when at
 * their level): the chain `out = eval.AddNew(out, temp)` over the products of a layer (cnn/cnn.go:19-30,58-62; equal scales: ring.Add per
 * component, mkckks/evaluator.go:316-327) as one launch.  Every partial sum is canonical, so the result is the chain's bit for bit. */
int  mkhe_ct_sum(mkhe_ctx* ctx, int n, const mkhe_ct* const* in, mkhe_ct* out);
/* mkckks.Evaluator.MulPtxtNew body (evaluator.go:465-478) without its Rescale: every component times the plaintext
 * polynomial dev_pt = uint64[limbs][N] (coefficient domain, device), via NTT / MForm / InvNTT. */
int  mkhe_ct_mul_ptxt(mkhe_ctx* ctx, const mkhe_ct* in, const void* dev_pt, mkhe_ct* out);

/* ---- B independent operations of ONE shape per call (round 4; no reference counterpart: the reference evaluates one ciphertext at a time).
 *      On the small rings (PN14QP439, mkckks/mkckks_benchmark_test.go:13; cnn's PN14QP433, cnn/cnn_test.go:80-96) an operation is a chain of
 *      launches of a few dozen limbs each; B inputs in lock step are the same launches with B times the items.  All ciphertexts of one list
 *      have the same ids and limb count; keys and CRS are per party and shared by the inputs; an operand that is the same ciphertext for
 *      every input (cnn's model) is passed nbatch times; outputs are distinct handles, and output k must not be an INPUT of another item j != k
 *      (an error: the items of a batch run as one launch set, in no order; output k may be input k where the single operation may run in place).
 *      Hoisted forms are flat lists [b * n + a] (input b,
 *      party component a) or NULL (the engine hoists).  Each output equals the single-operation entry point's, bit for bit.
 *        mkhe_hoisted_form_batch : mkckks.Evaluator.HoistedForm            mkckks/evaluator.go:543-553
 *        mkhe_rotate_batch       : KeySwitcher.RotateHoisted / Rotate      mkrlwe/keyswitch_hoisted.go:183-247, keyswitch.go:234-298
 *        mkhe_mul_relin_batch    : KeySwitcher.MulAndRelin[Hoisted] (+ the single Rescale of mkckks.Evaluator.mulRelinHoisted when rescale != 0:
 *                                  out is then one level below the product)     keyswitch_hoisted.go:44-179, mkckks/evaluator.go:558-581
 *        mkhe_ct_binary_batch    : op 0 = AddNew, 1 = SubNew               mkckks/evaluator.go:316-356 */
/*      Handles for a batch in one call: nbatch ciphertexts of one shape (contents undefined, like mkhe_ct_create_uninit) / count switching keys,
 *      views into ONE pooled block that is returned when the last of them has been destroyed (one by one with mkhe_ct_destroy / mkhe_swk_destroy, or
 *      all of them with the *_destroy_batch calls). */
int  mkhe_ct_create_batch(mkhe_ctx* ctx, int nbatch, int n, const int* ids, int limbs, mkhe_ct** out);
void mkhe_ct_destroy_batch(mkhe_ctx* ctx, int nbatch, mkhe_ct* const* cts);
int  mkhe_swk_create_batch(mkhe_ctx* ctx, int count, mkhe_swk** out);
void mkhe_swk_destroy_batch(mkhe_ctx* ctx, int count, mkhe_swk* const* swks);
int  mkhe_hoisted_form_batch(mkhe_ctx* ctx, int level, int nbatch, const mkhe_ct* const* cts, mkhe_swk* const* out);
int  mkhe_rotate_batch(mkhe_ctx* ctx, uint64_t galEl, int nbatch, const mkhe_ct* const* in, const mkhe_swk* const* hoist,
                       const mkhe_swk* const* rk, const mkhe_swk* crs, mkhe_ct* const* out);
/*      nbatch rotations of nbatch ciphertexts of one shape, EACH by its own Galois element galEl[b] with its own keys -- rk flat [b * n + a]
 *      (rkSet.GetRotationKey(ids[a], rotidx_b)), crs[b] = params.CRS[rotidx_b], hoist flat [b * n + a] or NULL -- and, when post_add != NULL,
 *      out[b] = post_add[b] + Rotate(in[b]) with the addition of mkckks.Evaluator.AddNew (equal scales: ring.Add) on the store of the rotation:
 *      the independent rotate -> hoist -> MulRelin chains of cnn.Convolution / FC1Layer (cnn/cnn.go:16-30,51-62) as lanes of one launch set, and
 *      the "temp = RotateNew(x, r); x = AddNew(x, temp)" steps of its log-sums (:33-37,64-67,83-86,90-93) as one pass (nbatch = 1).  Bit for bit
 *      what mkhe_rotate followed by mkhe_ct_add(post_add[b], .) gives.  post_add[b] has the ids of in[b], at out's level or above, and is not an output. */
int  mkhe_rotate_multi(mkhe_ctx* ctx, int nbatch, const uint64_t* galEl, const mkhe_ct* const* in, const mkhe_swk* const* hoist,
                       const mkhe_swk* const* rk, const mkhe_swk* const* crs, const mkhe_ct* const* post_add, mkhe_ct* const* out);
int  mkhe_mul_relin_batch(mkhe_ctx* ctx, int nbatch, const mkhe_ct* const* op0, const mkhe_ct* const* op1,
                          const mkhe_swk* const* hoist0, const mkhe_swk* const* hoist1,
                          const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_d0, const mkhe_swk* const* rlk_v0,
                          const mkhe_swk* crs_u, int rescale, mkhe_ct* const* out);
int  mkhe_ct_binary_batch(mkhe_ctx* ctx, int op, int nbatch, const mkhe_ct* const* op0, const mkhe_ct* const* op1, mkhe_ct* const* out);
/*      mkckks.Evaluator.MulPtxtNew (mkckks/evaluator.go:465-481): mkhe_ct_mul_ptxt followed by nb_rescale >= 0 DivRoundByLastModulus steps (the
 *      Rescale of :480; the host decides nb_rescale from the scales, :376-384); out has limbs(in) - nb_rescale limbs */
int  mkhe_ct_mul_ptxt_batch(mkhe_ctx* ctx, int nbatch, const mkhe_ct* const* in, const void* dev_pt, int nb_rescale, mkhe_ct* const* out);

/* ==== mkbfv ========================================================================================
 * Context for mkbfv.NewParametersFromLiteral (mkbfv/params.go:28-76): rings Q, QMul (same length), R = Q||QMul,
 * P and the plaintext modulus T; replaces mkbfv.NewKeySwitcher (keyswitch.go:31-65) + NewFastBasisExtender
 * (basis_extension.go:20-47).  PCount/gamma must be 1 (one prime per gadget digit), as in both reference
 * parameter sets.  Every mkrlwe-level call above works on such a context too (BFV Rotate / Conjugate use
 * mkhe_rotate / mkhe_conjugate directly: mkbfv/evaluator.go:142-207).
 * PolyR device layout: uint64[2*nQ][N], limbs 0..nQ-1 under Q, nQ..2nQ-1 under QMul. */
int  mkhe_ctx_create_bfv(mkhe_ctx** out, int logN, const uint64_t* Q, const uint64_t* QMul, int nQ,
                         const uint64_t* P, int nP, int gamma, uint64_t T, int device);
/* FastBasisExtender.ModUpQtoR / Rescale / Quantize (mkbfv/basis_extension.go:49-96) on raw device buffers:
 * npolys polynomials, polyq = [npolys][nQ][N], polyr = [npolys][2nQ][N] (Quantize: polyr in the NTT domain). */
int  mkhe_bfv_modup_q_to_r(mkhe_ctx* ctx, const void* dev_polyq, void* dev_polyr, int npolys);
int  mkhe_bfv_rescale(mkhe_ctx* ctx, const void* dev_polyq, void* dev_polyr, int npolys);
int  mkhe_bfv_quantize(mkhe_ctx* ctx, const void* dev_polyr_ntt, void* dev_polyq, int npolys);
/* ringR.NTT / InvNTT (keyswitch_hoisted.go:128-129) on [count][2nQ][N] */
int  mkhe_bfv_ntt_r(mkhe_ctx* ctx, const void* dev_src, void* dev_dst, int count, int inverse);
/* KeySwitcher.DecomposeBFV (mkbfv/keyswitch.go:67-90): one PolyR (coefficient domain) -> ad1 (Q digits), ad2 (QMul digits) */
int  mkhe_bfv_decompose(mkhe_ctx* ctx, const void* dev_polyr, mkhe_swk* ad1, mkhe_swk* ad2);
/* KeySwitcher.ExternalProductBFV (mkbfv/keyswitch.go:83-113): the non-hoisted form -- DecomposeBFV of one PolyR (coefficient
 * domain) into the engine's own pool, then the product below; c = [nQ][N], coefficient domain, canonical */
int  mkhe_bfv_external_product(mkhe_ctx* ctx, const void* dev_polyr, const mkhe_swk* bg1, const mkhe_swk* bg2, void* dev_c);
/* KeySwitcher.ExternalProductBFVHoisted (keyswitch_hoisted.go:6-34): c = ModDown_P(sum bg1.ah1 + bg2.ah2), [nQ][N] */
int  mkhe_bfv_external_product_hoisted(mkhe_ctx* ctx, const mkhe_swk* ah1, const mkhe_swk* ah2,
                                       const mkhe_swk* bg1, const mkhe_swk* bg2, void* dev_c);
/* Evaluator.MulRelinNew (mkbfv/evaluator.go:78-82) = mulRelinHoisted (:118-140) + MulAndRelinBFVHoisted
 * (keyswitch_hoisted.go:36-206).  The reference's non-hoisted twin (Evaluator.mulRelin evaluator.go:95-113 ->
 * KeySwitcher.MulAndRelinBFV keyswitch.go:115-251) decomposes the same polynomials inside its loops and yields the same
 * ciphertext bit for bit (tests/test_bfv_oracle.py); it has its own device path below (mkhe_bfv_mul_relin_unhoisted).
 * Ciphertexts at the maximum level, coefficient domain.  Key lists aligned with
 * the operand ids: rlk_b1/b2[j] = rlkSet[ids1[j]].Value[0/1].Value[0], rlk_d1/d2[i] = rlkSet[ids0[i]].Value[0/1].Value[1],
 * rlk_v[i] = rlkSet[ids0[i]].Value[0].Value[2]; crs_u = params.CRS[-1]. */
int  mkhe_bfv_mul_relin(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                        const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_b2,
                        const mkhe_swk* const* rlk_d1, const mkhe_swk* const* rlk_d2,
                        const mkhe_swk* const* rlk_v, const mkhe_swk* crs_u, mkhe_ct* out);
/* Evaluator.mulRelin (mkbfv/evaluator.go:95-113) -> KeySwitcher.MulAndRelinBFV (mkbfv/keyswitch.go:115-251): the reference's NON-hoisted
 * twin as its own device path, in the reference's order and with its pool discipline -- one pair of digit vectors that every
 * DecomposeBFV overwrites (each party component is decomposed twice), x / y accumulated party by party, every ExternalProductBFV /
 * ExternalProduct on its own.  Same arguments and the same ciphertext, bit for bit, as mkhe_bfv_mul_relin (tests/test_gpu_bfv.py); 2 digit
 * vectors of scratch instead of 4k, about twice the forward NTTs. */
int  mkhe_bfv_mul_relin_unhoisted(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                                  const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_b2,
                                  const mkhe_swk* const* rlk_d1, const mkhe_swk* const* rlk_d2,
                                  const mkhe_swk* const* rlk_v, const mkhe_swk* crs_u, mkhe_ct* out);
/* Party-sharded BFV MulRelinNew (no reference counterpart; mkhe_kklss_amd/dist.py ShardedBfvMulRelin; keyswitch_hoisted.go:76-206
 * is the structure being cut): every rank holds c_0 and BOTH components of the parties it owns (Quantize rounds, so the two tensor
 * terms of an output slot must meet on one rank).  mkhe_bfv_mr_partial: conversions, tensor + Quantize into `out` (out_0 only where
 * with_c0), DecomposeBFV, and the rank's canonical partial sums of x1, x2, y1, y2 -- to be summed over the ranks and folded with
 * mkhe_swk_fold(mform = 1); mkhe_bfv_mr_finish: steps E and F with the complete sums; out_0 and out_i are then partial
 * sums / owner slots to be exchanged and folded with mkhe_ct_fold.  mkhe_bfv_mul_relin == partial(with_c0 = 1) + finish on one rank. */
int  mkhe_bfv_mr_partial(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                         const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_b2,
                         const mkhe_swk* const* rlk_d1, const mkhe_swk* const* rlk_d2, int with_c0, mkhe_ct* out,
                         mkhe_swk* x1, mkhe_swk* x2, mkhe_swk* y1, mkhe_swk* y2);
int  mkhe_bfv_mr_finish(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, const mkhe_swk* x1, const mkhe_swk* x2,
                        const mkhe_swk* y1, const mkhe_swk* y2, const mkhe_swk* const* rlk_v, const mkhe_swk* crs_u, mkhe_ct* out);

/* ==== key generation and CRS expansion =========================================
 * mkrlwe/keygen.go, mkbfv/keygen.go, mkrlwe/params.go:16-61,77-99.  The reference draws secrets, errors and CRS from
 * lattigo's crypto PRNG (utils.NewPRNG: keygen.go:26, params.go:28,79), so no output of it can be reproduced bit for
 * bit; what is reproduced is the ring arithmetic applied to the samples:
 *  - secrets / errors are SAMPLES supplied by the caller as host int32 arrays, N small signed coefficients per
 *    polynomial (what ring.TernarySampler / ring.GaussianSampler draw; the Go shim copies them out of its own samplers,
 *    so the secret randomness never comes from the GPU),
 *  - a SecretKey is a device PolyQP buffer uint64[nQ+nP][N] (mkhe_buf_alloc), NTT domain, Montgomery form
 *    (SecretKey.Value, keygen.go:44-55),
 *  - keys are written into SwitchingKey handles in the layout every other entry point expects. */
/* genSecretKeyFromSampler keygen.go:44-55 */
int  mkhe_keygen_secret(mkhe_ctx* ctx, const int32_t* s, void* dev_sk);
/* GenSwitchingKey keygen.go:270-327: g*sk + e; e = int32[betaMax][N] */
int  mkhe_keygen_switching_key(mkhe_ctx* ctx, const void* dev_sk, const int32_t* e, mkhe_swk* out);
/* GenPublicKey keygen.go:88-109: dev_pk = uint64[2][nQ+nP][N], pk[0] = NTT(e) - sk*a, pk[1] = a = CRS[0].Value[0]; e = int32[N] */
int  mkhe_keygen_public_key(mkhe_ctx* ctx, const void* dev_sk, const int32_t* e, const mkhe_swk* crs_a, void* dev_pk);
/* GenRelinearizationKey keygen.go:137-187: (b, d, v) from sk, the auxiliary secret r, a = CRS[0], u = CRS[-1];
 * e = int32[3][betaMax][N], the errors of b, d, v in this order */
int  mkhe_keygen_relin_key(mkhe_ctx* ctx, const void* dev_sk, const void* dev_r, const int32_t* e,
                           const mkhe_swk* crs_a, const mkhe_swk* crs_u, mkhe_swk* b, mkhe_swk* d, mkhe_swk* v);
/* GenRotationKey keygen.go:190-229: galEl = 5^rotidx mod 2N, crs = CRS[rotidx]; e = int32[betaMax][N] */
int  mkhe_keygen_rotation_key(mkhe_ctx* ctx, uint64_t galEl, const void* dev_sk, const int32_t* e,
                              const mkhe_swk* crs, mkhe_swk* out);
/* GenConjugationKey keygen.go:240-268: crs = CRS[-2] */
int  mkhe_keygen_conjugation_key(mkhe_ctx* ctx, const void* dev_sk, const int32_t* e, const mkhe_swk* crs, mkhe_swk* out);
/* mkbfv GenBFVSwitchingKey (mkbfv/keygen.go:91-162), one of its two loops: g = uint64[betaMax][nQ+nP], the residues
 * (plain, < modulus) of the big-integer gadget scalars Gi (:104-116 resp. :137-149), computed by the caller */
int  mkhe_bfv_keygen_switching_key(mkhe_ctx* ctx, const void* dev_sk, const uint64_t* g, const int32_t* e, mkhe_swk* out);
/* mkbfv GenRelinearizationKey (mkbfv/keygen.go:24-88): a1 = CRS[0], a2 = CRS[-3], u = CRS[-1];
 * e = int32[5][betaMax][N] for b1, b2, d1, d2, v */
int  mkhe_bfv_keygen_relin_key(mkhe_ctx* ctx, const void* dev_sk, const void* dev_r, const uint64_t* g1, const uint64_t* g2,
                               const int32_t* e, const mkhe_swk* crs_a1, const mkhe_swk* crs_a2, const mkhe_swk* crs_u,
                               mkhe_swk* b1, mkhe_swk* b2, mkhe_swk* d1, mkhe_swk* d2, mkhe_swk* v);
/* CRS[idx] (params.go:47-59, AddCRS :77-99) expanded on the device from a public seed instead of uploaded (56 MiB each
 * at PN15QP880): limb (digit i, modulus j) coefficient w = MForm(first of the 64-bit words of
 * Philox4x32-10(key = seed, counter = {w, i*(nQ+nP)+j, idx, block}), block = 0, 1, ..., two words per block, masked to
 * bitlen(q_j) bits, that is < q_j) -- the mask-and-reject shape of lattigo's ring.UniformSampler.  Parties that share
 * the seed derive the same CRS. */
int  mkhe_crs_expand(mkhe_ctx* ctx, uint64_t seed, int32_t idx, mkhe_swk* out);

/* ---- measurement support (no reference counterpart): HIP-event timing per kernel class on the
 *      context stream, one record per kernel launch.  Classes (mkhe_prof_name gives the kernel symbol
 *      each class corresponds to in a rocprofv3 kernel trace). */
int  mkhe_prof_enable(mkhe_ctx* ctx, int on);
/* on = 0: no side-stream overlap, every kernel runs alone on the main stream (per-kernel timings that can be
 * compared with a rocprofv3 kernel trace); default on = 1 */
int  mkhe_set_overlap(mkhe_ctx* ctx, int on);
/* diagnostic: forward-NTT workgroups write {start, end (100 MHz ticks), HW_ID, XCC_ID} per job into dev_buf
 * (4 words per job of the NEXT launches; NULL switches it off) */
int  mkhe_ntt_trace(mkhe_ctx* ctx, void* dev_buf);
/* N = 2^15, where two forward kernels apply (same bits): by default (MKHE_NTT32=2) the context times a block of launches of each inside the caller's
 * workload and keeps the faster one per launch shape.  The kernel it has settled on for Decompose launches of `limbs` limb-NTTs: 1 = single-pass
 * (ntt32_fwd_kernel), 0 = two-pass (ntt16_fwd_kernel), -1 = still measuring, never launched, or fixed by MKHE_NTT32 = 0 / 1. */
int  mkhe_ntt_choice(mkhe_ctx* ctx, long limbs, int decompose);      /* -2: error (mkhe_last_error) */
/* Pins that choice instead of measuring it, so that a run can be repeated with the kernels of an earlier one (a bench line records
 * config.ntt_kernel_choice): choice 1 = single-pass, 0 = two-pass, -1 = forget and measure again.  limbs > 0: launches of that many limb-NTTs
 * (decompose = 1: Decompose launches, 0: plain forward transforms); limbs <= 0: every shape of the context, met so far or not.  Same bits either way. */
int  mkhe_ctx_set_ntt_choice(mkhe_ctx* ctx, long limbs, int decompose, int choice);
/* mkhe_mul_relin_batch at N = 2^15: inputs whose hoisting is at least `min_limbs` limb-NTTs ((n0 + n1) * beta * (level + 1 + nP); default 1536: four
 * parties at the top level of PN15QP880) are evaluated IN FLIGHT -- the single-operation path on this context and two internal ones, round robin, joined
 * before the call returns to the stream -- instead of in lock step: one such evaluation fills the chip with every big kernel, what a second one can use
 * is the first one's latency-bound stretches.  0: every batch of this ring in flight; < 0: always lock step.  Same results either way
 * (mkrlwe/keyswitch_hoisted.go:44-179 per input).  The internal contexts follow the caller's overlap setting and its pin for every shape
 * (mkhe_ctx_set_ntt_choice with limbs <= 0); pins of single launch shapes are the caller context's own. */
int  mkhe_ctx_set_batch_lanes(mkhe_ctx* ctx, long min_limbs);
/* The stream-ordered buffer pools (freed ciphertext / key handles are kept for reuse, bounded per DEVICE by MKHE_POOL_GB): bytes this context's
 * pool holds, and "hand everything the pools of this context's device hold back to the driver" (one device-wide synchronisation; the engine does
 * this by itself when an allocation fails for lack of memory and retries once). */
long long mkhe_pool_held_bytes(mkhe_ctx* ctx);                          /* -1: error */
int  mkhe_pool_trim(mkhe_ctx* ctx);
/* diagnostic, no device call: the schedule of ntt16_f2_kernel (N = 2^15: step F2 of MulAndRelin, mkrlwe/keyswitch_hoisted.go:161-178, computed inside
 * the Decompose NTT of the t_i) for `parties` parties of op0, `nb` gadget digits, `nslots` limb slots with the relative cost weights[slot] of a pass,
 * on `grid` workgroups (grid < 0: on the cheapest grid of at most -grid workgroups, as the engine plans it for a device of -grid CUs).  segs: |grid| * 3
 * records of 8 bytes {party, slot, half, first digit, digits, part, parts this run zeroes after its own, 0} (digits = 0: no run).  Returns the workgroups
 * used (0: no schedule within the kernel's limits) and the parts per product in *parts. */
int  mkhe_f2_schedule_probe(int parties, int nb, int nslots, const long* weights, int grid, unsigned char* segs, int* parts);
int  mkhe_prof_nclass(void);
const char* mkhe_prof_name(int cls);
int  mkhe_prof_collect(mkhe_ctx* ctx, double* ms, long* launches, double* alg_bytes);

#ifdef __cplusplus
}
#endif
#endif
