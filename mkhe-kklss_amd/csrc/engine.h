// engine.h -- device-resident multi-key RLWE key-switch engine (host-side C++ over HIP).
//
// One Context = one mkrlwe.KeySwitcher (mkrlwe/keyswitch.go:8-47): ring tables for Q and P in
// HBM, the ModDown / rescale constants, and the scratch pools the reference keeps in
// ks.Pool / swkPool1-3 / polyQPool (engine-internal here).  One HIP stream per context; calls
// on a context are serialized by the caller, exactly like the non-reentrant reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <atomic>
#include <mutex>
#include <map>
#include <memory>
#include <vector>
#include <stdexcept>
#include "modarith.h"
#include "ntt_kernels.h"
#include "poly_kernels.h"
#include "keygen_kernels.h"

namespace mkhe {

// A/B switches that several translation units ask for (csrc/switches.h: constants in the product library): x, y and step E inside the F1 kernel
int ab_fuse_x();
int ab_fuse_y();
int ab_fuse_e();

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

#define MKHE_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw ::mkhe::Error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// device SwitchingKey / hoisted digit vector: uint64[betaMax][nQ+nP][N]
struct Swk { u64* d = nullptr; bool owned = true; };
// device ciphertext: uint64[1+n][limbs][N], slot 0 = c_0, slot 1+i = party ids[i]
struct Ct { int n = 0; int limbs = 0; std::vector<int> ids; u64* d = nullptr; };
// one external product of a batch: dst (+)= ModDown_P( sum_i bg[i] (.) ah[i] )
struct ExtItem { const u64* ah; const u64* bg; u64* dst; bool accumulate; const u64* ah2 = nullptr; const u64* bg2 = nullptr;
                 const u64* xkey = nullptr;   /* F1 only: this party's d_i, for the x by-product (ExtInnerArgs::xkey) */
                 bool pre = false;            /* the inner products of this item exist already (step E computed by the F1 kernel): in its c1 slot, or ... */
                 const u64* pre_src = nullptr; /* ... at pre_src [mtot][N] (a batch: the inner kernel copies them into the slot instead of computing) */
                 const u64* xkey2 = nullptr;  /* mkbfv F1: this party's d2_i (second gadget) */
                 const u64* addend = nullptr; /* accumulate onto this polynomial instead of onto dst (Rotate: c_0 of the input) */
                 const u64* qadd = nullptr;   /* first product of a destination, not accumulating: an NTT-domain polynomial [L][N] (canonical, already
                                                 times P) that joins the summed Q parts of the merged batch -- dst = that term + the products */
                 unsigned gal = 0;            /* rotate_multi: this destination's own Galois element (0: the ext_batch call's); same for every product of a destination */
                 const u64* post = nullptr;   /* rotate_multi: polynomial [L][N] added to the finished destination at the STORED (permuted) position --
                                                 ring.Add(post, Rotate(..)) of AddNew(ct, RotateNew(ct, r)), cnn/cnn.go:33-37 -- by the last product of the destination */
                 int f2_party = -1;           /* >= 0: step F2 computed by ntt16_f2_kernel (ntt_kernels.h F2FusedArgs) from the party's t (Context::ext_f2_src_): `ah` is not read */
                 int f2_key = 0;              /* ... 0: the product with v_i, 1: with the CRS u */ };

typedef unsigned long long seq_t;
// per handle: (uid of a context, that context's call counter at its latest use of the buffer); `exposed` once the raw device
// pointer was handed out (uses can no longer be seen by the C ABI)
struct HandleUsers { std::vector<std::pair<seq_t, seq_t>> v; bool exposed = false; };

class Context {
  public:
    // QMul / T: optional mkbfv extension (mkbfv/params.go:30-76): nqm = nq extra primes and the plaintext modulus
    Context(int logN, const u64* Q, int nq, const u64* P, int np, int gamma,
            const u64* psiQ, const u64* psiP, int device, const u64* QMul = nullptr, int nqm = 0, u64 T = 0);
    ~Context();

    int logN, N, nq, np, mtot, gamma, alpha, beta_max, device;
    int nqm = 0, mall = 0;          // QMul primes (BFV) and the total number of moduli with tables
    u64 bfv_t = 0;
    std::vector<u64> moduli;        // Q then P (then QMul)
    std::vector<u64> psi_plain;     // 2N-th roots actually used
    hipStream_t stream = nullptr;      // main stream: everything the caller observes is ordered on it
    hipStream_t stream2 = nullptr;     // side stream for independent sub-chains (always joined back into `stream`)

    int beta(int level) const { return (level + 1 + alpha - 1) / alpha; }
    size_t swk_words() const { return (size_t)beta_max * mtot * N; }
    size_t poly_words(int limbs) const { return (size_t)limbs * N; }

    // ---- ring-level entry points (tests / bench / callers that hold raw device buffers)
    // NTT of `count` polys of `limbs` limbs each; modulus of limb l is mod_base + l.
    void ntt(const u64* src, u64* dst, int count, int limbs, int mod_base, bool inverse, bool lazy);

    // ---- KeySwitcher method set (mkrlwe/keyswitch.go, keyswitch_hoisted.go)
    void decompose(int level, bool is_ntt, const u64* a /*[level+1.. ][N] device*/, u64* out_swk);
    void external_product_hoisted(int level, const u64* ah, const u64* bg, u64* c, bool accumulate);
    void external_product(int level, bool is_ntt, const u64* a, const u64* bg, u64* c, bool accumulate);
    void mul_and_relin(const Ct& op0, const Ct& op1, const Swk* const* hoist0, const Swk* const* hoist1,
                       const Swk* const* rlk_b1, const Swk* const* rlk_d0, const Swk* const* rlk_v0,
                       const Swk& crs_u, Ct& out);
    void rotate(u64 galEl, const Ct& in, const Swk* const* hoist, const Swk* const* rk, const Swk& crs, Ct& out);
    void conjugate(u64 galEl, const Ct& in, const Swk* const* ck, const Swk& crs, Ct& out);
    void rotate_partial(const Ct& in, const Swk* const* hoist, const Swk* const* rk, const Swk& crs, bool with_c0, Ct& out);
    void rotate_core(const Ct& in, const Swk* const* hoist, const Swk* const* rk, const Swk& crs, bool with_c0, Ct& out, u64 galEl);
    void automorphism(u64 galEl, const Ct& in, Ct& out);
    // ---- mkbfv (mkbfv/basis_extension.go, keyswitch.go, keyswitch_hoisted.go, evaluator.go); PolyR = [2nq][N]
    bool is_bfv() const { return nqm > 0; }
    void bfv_modup_q_to_r(const u64* polyq, u64* polyr, int npolys);          // conv.ModUpQtoR
    void bfv_rescale(const u64* polyq, u64* polyr, int npolys);               // conv.Rescale
    void bfv_quantize(const u64* polyr_ntt, u64* polyq, int npolys);          // conv.Quantize (polyr_ntt is consumed)
    void bfv_decompose_batch(const std::vector<const u64*>& srcr, const std::vector<u64*>& ad1, const std::vector<u64*>& ad2, bool internal = false);
    void bfv_external_product_hoisted(const u64* ah1, const u64* ah2, const u64* bg1, const u64* bg2, u64* c);
    void bfv_external_product(const u64* polyr, const u64* bg1, const u64* bg2, u64* c);
    void bfv_mr_partial(const Ct& op0, const Ct& op1, const Swk* const* rlk_b1, const Swk* const* rlk_b2,
                        const Swk* const* rlk_d1, const Swk* const* rlk_d2, bool with_c0, bool mform, Ct& out,
                        u64* x1, u64* x2, u64* y1, u64* y2, bool fuse_x = false, bool fuse_y = false);
    void bfv_mr_finish(const Ct& op0, const Ct& op1, const u64* x1, const u64* x2, const u64* y1, const u64* y2,
                       const Swk* const* rlk_v, const Swk& crs_u, Ct& out);
    void bfv_slots(const Ct& op0, const Ct& op1, const Ct& out, std::vector<int>& slot0, std::vector<int>& slot1) const;
    bool bfv_plan_valid_ = false;
    void mul_relin_rescale(const Ct& op0, const Ct& op1, const Swk* const* hoist0, const Swk* const* hoist1, const Swk* const* rlk_b1,
                           const Swk* const* rlk_d0, const Swk* const* rlk_v0, const Swk& crs_u, Ct& out);
    void bfv_mul_relin_unhoisted(const Ct& op0, const Ct& op1, const Swk* const* rlk_b1, const Swk* const* rlk_b2,
                                 const Swk* const* rlk_d1, const Swk* const* rlk_d2, const Swk* const* rlk_v, const Swk& crs_u, Ct& out);
    void bfv_mul_relin(const Ct& op0, const Ct& op1, const Swk* const* rlk_b1, const Swk* const* rlk_b2,
                       const Swk* const* rlk_d1, const Swk* const* rlk_d2, const Swk* const* rlk_v,
                       const Swk& crs_u, Ct& out);                            // Evaluator.MulRelinNew
    // elementwise evaluator ops (mkckks/evaluator.go:41-300, mkbfv/evaluator.go:27-76): op 0 add, 1 sub
    void ct_binary(int op, const Ct& a, const Ct& b, Ct& out);
    // mkckks MultByConst body (evaluator.go:150-196) and MulPtxtNew body without its Rescale (:471-478)
    void ct_mul_const(const Ct& in, const u64* c_first, const u64* c_second, Ct& out);
    void ct_mul_ptxt(const Ct& in, const u64* dev_pt, Ct& out);
    void ntt_r(const u64* src, u64* dst, int count, bool inverse);            // ringR.NTT / InvNTT
    // mkckks Rescale body: nb successive DivRoundByLastModulus on every poly (evaluator.go:385-391)
    void rescale(const Ct& in, int nb, Ct& out);
    // MulAndRelin split in phases (party-sharded multi-GPU evaluation: the x / y partial sums and out_0
    // are reduced across devices between the phases)
    void mr_prepare(const Ct& op0, const Ct& op1, const Swk* const* hoist0, const Swk* const* hoist1, bool with_c0, Ct& out);
    // defer_x: x is accumulated on the side stream and joined by mr_finish just before its first use (step E), so that it
    // overlaps the latency-bound part of step F; false (split-phase ABI): x and y are both complete on the main stream
    // fuse_x: x is not computed here but by the F1 kernel of mr_finish_head (one pass over h(c0_i) less); needs 1 <= n0 <= 4
    // fuse_y (with fuse_x, n1 == n0 <= 4): y is not computed here either but inside the same F1 kernel, from the b_j and h(c1_j), and never stored
    void mr_xy(const Swk* const* rlk_b1, const Swk* const* rlk_d0, u64* x, u64* y, bool mform, bool defer_x = false, bool fuse_x = false, bool fuse_y = false);
    void mr_finish_head(const Ct& op0, const Ct& op1, const u64* y, Ct& out);          // F1 + Decompose(t_i): needs y only
    void mr_finish_tail(const Ct& op0, const Ct& op1, const u64* x, const Swk* const* rlk_v0, const Swk& crs_u, Ct& out);
    void mr_finish(const Ct& op0, const Ct& op1, const u64* x, const u64* y, const Swk* const* rlk_v0,
                   const Swk& crs_u, Ct& out);
    void fold(u64* buf, bool qp_shaped, int level, int npolys, long poly_stride, bool mform);
    // the reduce-scatter half of a mesh exchange of x / y (dist.py): limbs [first_limb, first_limb + nlimbs) of a switching-key buffer, summed over
    // npieces pieces, folded and MForm'ed into dst (limb i of the range at dst + i * N)
    void fold_pieces(const u64* pieces, int npieces, long piece_stride, long first_limb, long nlimbs, int level, bool mform, u64* dst);
    // ---- B independent operations of one shape as one launch set (batch.hip; small rings: DESIGN.md section 8).  Flat lists: hoisted forms are
    // [b * n + a] (input b, party component a) or empty (the engine hoists); keys are per party, shared by the inputs.
    void hoisted_form_batch(int level, const std::vector<const Ct*>& cts, const std::vector<Swk*>& outs);
    void rotate_batch(u64 galEl, const std::vector<const Ct*>& ins, const std::vector<const Swk*>& hoists, const Swk* const* rk, const Swk& crs,
                      const std::vector<Ct*>& outs);
    void mul_relin_batch(const std::vector<const Ct*>& op0, const std::vector<const Ct*>& op1, const std::vector<const Swk*>& hoist0,
                         const std::vector<const Swk*>& hoist1, const Swk* const* rlk_b1, const Swk* const* rlk_d0, const Swk* const* rlk_v0,
                         const Swk& crs_u, bool rescale_out, const std::vector<Ct*>& outs);
    // B rotations in one launch set, each with its own Galois element and keys (rk flat [b * n + a], crs per input), optionally
    // out[b] = post_add[b] + Rotate(in[b]) (batch.hip; nbatch = 1: the fused rotate-and-add of one ciphertext)
    void rotate_multi(const std::vector<u64>& galEl, const std::vector<const Ct*>& ins, const std::vector<const Swk*>& hoists, const std::vector<const Swk*>& rk,
                      const std::vector<const Swk*>& crs, const std::vector<const Ct*>& post_add, const std::vector<Ct*>& outs);
    // out = ins[0] + ... + ins[n-1] (one shape; out at its own level <= theirs): the AddNew chain over the lanes' products as one launch
    void ct_sum(const std::vector<const Ct*>& ins, Ct& out);
    void ct_binary_batch(int op, const std::vector<const Ct*>& a, const std::vector<const Ct*>& b, const std::vector<Ct*>& outs);
    // MulPtxtNew body (ct_mul_ptxt) for the batch, followed by nb >= 0 divisions by the last modulus (outs have limbs(in) - nb limbs)
    void ct_mul_ptxt_batch(const std::vector<const Ct*>& ins, const u64* dev_pt, int nb, const std::vector<Ct*>& outs);
    u64* pool_x() { return x_; }
    u64* pool_y() { return y_; }
    // batched building blocks (all parties in one launch)
    // internal = true: the digits stay inside the engine (hoist pools): their forward NTT skips the final
    // normalisation (values < 34q with the same residues; every consumer is a Montgomery product)
    // stage_only (alpha = 1, N = 2^14, engine-internal): digit spread + the three cross stages only -- the digits are then consumed by ext_batch through
    // ext_fused_lds_kernel (ext_staged_ names them), which finishes the transform in LDS and multiplies in the same kernel (ntt_kernels.h, ExtFusedArgs)
    void decompose_batch(int level, const std::vector<const u64*>& src, const std::vector<u64*>& dst, bool internal = false, bool stage_only = false);
    bool ext_fused_ok(int level, int nvec) const;
    // stage 0: whole external products; 1: front half only (inner products + inverse NTT into the c1 pool);
    // 2: back half only (ModDown of the c1 pool filled by the preceding stage-1 call with the same items)
    void ext_batch(int level, const std::vector<ExtItem>& items, int join_before_moddown = -1, int stage = 0, u64 galEl = 0);
    u64* ext_xout_ = nullptr;             // set around the one ext_batch call that carries the x by-product
    std::vector<const u64*> staged_open_; // digit vectors left after the cross stages that no product kernel has finished yet: an ext_batch that reads one of them as a
                                          // full transform is an engine bug and throws (round 5: mkhe_rotate_batch did, for one afternoon, on launches only a fuzz run reached)
    std::vector<const u64*> ext_staged_;  // set around the one ext_batch call whose items read digit vectors that decompose_batch left staged (stage_only)
    // N = 2^15 (round 6): the tail batch of a MulAndRelin whose F2 products come out of the Decompose NTT of the t_i itself (ntt16_f2_kernel): the t_i by
    // party, set around that ext_batch call; the items carry ExtItem::f2_party
    std::vector<const u64*> ext_f2_src_;
    struct F2Sched { F2Seg* d_segs = nullptr; int nwg = 0, parts = 1; };
    std::map<long, F2Sched> f2_sched_;                   // by (parties, level): the runs of every workgroup, in device memory
    const F2Sched& f2_schedule(int nparties, int level);
    bool f2_fused_ok(int level, int n0, int n1) const;
    u64* ext_xout2_ = nullptr;            // ... and the second gadget's x (mkbfv)
    std::vector<u64*> ext_eouts_;                           // a batch's F1 call that computes step E too: where input b's E products go ([n1][mtot][N])
    int ext_e_slot_ = -1;                                  // >= 0 around the F1 call that computes step E too: first c1 slot of the E products
    std::vector<const u64*> ext_ykeys_, ext_yh_;          // set around the F1 call whose kernel computes y itself (ExtXyArgs::ykey / yh); with ext_xmap_ (a
                                                         // batch): ext_yh_ holds every input's digits in turn, ext_ykeys_.size() per input
    std::vector<std::pair<const u64*, u64*>> ext_xmap_;   // batch.hip: (shared key y_b, x_b) per input around the F1 call of a batch: one x per group
    std::vector<const u64*> bfv_xk1_, bfv_xk2_;   // mkbfv single-device MulRelinNew: d1_i, d2_i for the fused x1, x2
    std::vector<const u64*> bfv_yk1_, bfv_yk2_;   // ... and b1_j, b2_j when y1, y2 (and step E) are computed inside the F1 kernel too
    std::vector<const u64*> ext_ykeys2_, ext_yh2_;  // second gadget of ext_ykeys_ / ext_yh_
    // External products that ModDown adds into ONE destination are merged (ModDown is linear in the Q part, see NttBatch::vi and
    // ModDownMergedArgs): virtual item v = up to VI_MAX items of the batch with the same destination; their Q limbs are summed in the
    // NTT domain at the load of ONE inverse NTT, their P limbs are transformed and lifted one by one.  MKHE_EXT_MERGE=0 switches it off.
    struct ExtMerge {
        int nvi = 0, members_max = 0;
        unsigned char cnt[64] = {}, mem[64][4] = {}, accumulate[64] = {};
        u64* dst[64] = {};
        const u64* addend[64] = {};
        const u64* qadd[64] = {};
        unsigned gal[64] = {};
        const u64* post[64] = {};
    };
    int ext_merge_members(int level) const;      // members a virtual item may have at this level (< 2: no merging)
    bool ext_plan_merge(int level, const ExtItem* items, int n, ExtMerge& mp) const;
    void ext_front(int level, const ExtItem* items, int n, u64* c1, const ExtMerge* mp = nullptr);   // inner products + lazy inverse NTT into c1 [n][mtot][N]
    bool ext_front_f2(int level, const ExtItem* items, int n, u64* c1, ExtInnerArgs& ia, unsigned short* f2_parts);
    void ext_back(int level, const ExtItem* items, int n, const u64* c1, u64 galEl = 0, const ExtMerge* mp = nullptr);   // ModDown of c1 into / onto the destinations [signed-permuted]

    // ---- limb-sharded multi-GPU evaluation (mkhe_kklss_amd/dist.py LimbShardedMulRelin): this context owns a subset of
    // the RNS moduli ("slots"); NTTs, inner products and ModDown outputs are computed for the owned slots only, all
    // parties.  The per-party sums x, y are then LOCAL; what crosses the links are the P limbs of the external
    // products, t_i and the output ciphertext (about 40 MB per step instead of 130 MB for party sharding).
    void set_owned(const int* mod_idx, int n);          // n == 0: everything (single-device behaviour)
    bool masked() const { return masked_; }
    // MulAndRelin in four phases around three exchanges through `stage` (device, caller-owned); each returns the number
    // of words of `stage` to all-reduce (sum of disjoint slices) before the next phase.  After phase 4 `out` holds the
    // owned limbs (zeros elsewhere) and is all-reduced itself.
    size_t lsh_phase(int phase, const Ct& op0, const Ct& op1, const Swk* const* rlk_b1, const Swk* const* rlk_d0,
                     const Swk* const* rlk_v0, const Swk* crs_u, Ct& out, u64* stage);

    // ---- key generation and CRS expansion (mkrlwe/keygen.go, mkbfv/keygen.go, params.go:16-99).
    // Secrets and errors are SAMPLES supplied by the caller (host int32, N per polynomial: what lattigo's ternary /
    // Gaussian samplers draw); secret keys are device PolyQP buffers [nq+np][N] (NTT, Montgomery form).
    void keygen_secret(const int32_t* s, u64* dev_sk);                                         // keygen.go:44-55
    void keygen_switching_key(const u64* sk, const int32_t* e, u64* out);                      // keygen.go:270-327
    void keygen_public_key(const u64* sk, const int32_t* e, const u64* crs_a, u64* dev_pk);    // keygen.go:88-109
    void keygen_relin_key(const u64* sk, const u64* r, const int32_t* e, const u64* crs_a, const u64* crs_u,
                          u64* b, u64* d, u64* v);                                             // keygen.go:137-187
    void keygen_rotation_key(u64 galEl, const u64* sk, const int32_t* e, const u64* crs, u64* out);   // keygen.go:190-229
    void keygen_conjugation_key(const u64* sk, const int32_t* e, const u64* crs, u64* out);    // keygen.go:240-268
    // mkbfv/keygen.go:91-162 (one gadget) and :24-88; g1 / g2: host residues [beta][nq+np] of the big-integer scalars Gi
    void bfv_keygen_switching_key(const u64* sk, const u64* g, const int32_t* e, u64* out);
    void bfv_keygen_relin_key(const u64* sk, const u64* r, const u64* g1, const u64* g2, const int32_t* e,
                              const u64* crs_a1, const u64* crs_a2, const u64* crs_u, u64* b1, u64* b2, u64* d1, u64* d2, u64* v);
    void crs_expand(u64 seed, int32_t idx, u64* out);                                          // params.go:47-59,91-98

    bool overlap = true;               // false: everything on the main stream (clean per-kernel timings)
    u64* ntt_trace = nullptr;          // diagnostic buffer handed to the forward NTT kernels (mkhe_ntt_trace)
    // stream-ordered buffer cache for ciphertext / switching-key handles: freeing a handle does not
    // synchronise or hipFree, the words go back to a size-keyed free list and are reused by the next
    // create (all work of a context is ordered on its main stream, so reuse is safe).
    u64* pool_alloc(size_t words);
    size_t pool_take_all(std::vector<u64*>& out);      // empties the free list into `out` (the caller synchronises the device, then frees); words taken
    size_t pool_held_words();                          // what this context's free list holds
    size_t pool_trim_device();                         // every context's free list of this device back to the driver; words released
    void pool_free(u64* p, size_t words, const HandleUsers* users = nullptr);   // users == nullptr / exposed: every live context counts
    void sync() { const unsigned long long s0 = seq_.load(); MKHE_HIP(hipStreamSynchronize(stream)); if (completed_.load() < s0) completed_.store(s0); }
    // cross-context ordering on one device: everything enqueued on this context from now on starts after everything
    // enqueued on `other` so far has finished (event on other's stream; no host synchronisation).  Contexts over the same
    // ring share keys / CRS / ciphertext handles freely (handles are plain device memory): independent operations issued
    // through different contexts overlap on the GPU.
    void wait_for(Context& other);

    // ---- per-kernel-class timing with HIP events on the context stream (bench.py roofline leg)
    enum { PROF_NTT_DECOMP = 0, PROF_NTT_DECOMP_BIGQ, PROF_NTT_DECOMP_MIXED, PROF_NTT16_DECOMP, PROF_NTT16_FWD, PROF_NTT32_DECOMP, PROF_NTT32_FWD, PROF_NTT14_SPLIT, PROF_NTT_FWD, PROF_NTT_FWD_BIGQ, PROF_NTT_INV, PROF_INNER, PROF_EXT_INNER,
           PROF_MODDOWN, PROF_TENSOR, PROF_BASISCONV, PROF_SPREAD, PROF_NTT_F2, PROF_OTHER, PROF_NCLASS };
    void prof_enable(bool on);
    void prof_collect(double* ms, long* launches, double* alg_bytes);
    void recover();             // after an exception: active stream back to the main stream, plans dropped, both streams drained   // arrays of PROF_NCLASS; syncs and resets

    // device tables (public for the C ABI accessors / tests)
    Mod* d_mods = nullptr;
    u64 *d_psi = nullptr, *d_psiinv = nullptr, *d_inv_aux = nullptr;
    u64* d_psi31b = nullptr;                     // [mall][16][64][2]: the pairs of stages 5..9 per wave of ntt32_kernels.hip (NttBatch::psi31b)
    u64* d_psi31c = nullptr;                     // [mall][16][31][64][2]: the pairs of stages 10..14 in the load order of ntt32_kernels.hip (NttBatch::psi31c)
    u64* d_psi31n = nullptr;                     // [mall][4][2]: pairs of -psi[1..3] (NttBatch::psi31n)
    u64 *d_psiinv31 = nullptr, *d_inv31c = nullptr;   // inverse twiddles as pairs [mall][N][2]; last-stage constants [mall][8][6] (NttBatch::inv31c)
    unsigned long long small_mods_ = 0;          // bit m set = modulus m has 31 q < 2^62
    u64* d_psif = nullptr;                       // [mall][N] forward twiddles as doubles, N = 2^16 only (NttBatch::psif)
    unsigned long long f_mods_ = 0;              // bit m set = modulus m is of the F class (80 q < 2^52)
    std::vector<unsigned char> h16_sched_;       // per modulus: NttBatch::sched
    unsigned long long u_mods_ = 0;              // bit m: modulus m is of the H16 kernel's U class (NttBatch::u_mods)
    u64* d_psi31 = nullptr;                      // logN >= 15: twiddle pairs of the H16 kernel's one-round product (NttBatch::psi31)
    bool h16_gap_ = false;                       // some modulus has 31q < 2^62 <= 48q: H16's signed ranges do not cover it
    int *d_map_qp = nullptr, *d_map_id = nullptr;
    u64 *d_md_qoverqiinvqi = nullptr, *d_md_qoverqimodp = nullptr, *d_md_vtimes = nullptr, *d_md_down = nullptr;
    u64* d_pmodq = nullptr;                      // [nq] MForm(P mod q_j)
    u64* d_rescale = nullptr;
    u64 *d_dec_a = nullptr, *d_dec_b = nullptr, *d_dec_c = nullptr;     // Decomposer tables (alpha >= 2)
    u64 *d_tb30 = nullptr, *d_tw30 = nullptr;                           // ... and those of the radix-4 spread (alpha = 2, N = 2^16, moduli < 2^57)
    // mkbfv tables: convQQMul in both directions, ModDown constants, mFormQMul, MForm(t) per limb of R
    int* d_map_r = nullptr;
    u64 *d_bq_qoverqiinvqi = nullptr, *d_bq_qoverqimodp = nullptr, *d_bq_vtimes = nullptr;     // Q -> QMul
    u64 *d_bm_qoverqiinvqi = nullptr, *d_bm_qoverqimodp = nullptr, *d_bm_vtimes = nullptr;     // QMul -> Q
    u64 *d_down_q_in_m = nullptr, *d_down_m_in_q = nullptr, *d_mform_qmul = nullptr, *d_t_mont = nullptr;

  private:
    // scratch pools
    u64 *x_ = nullptr, *y_ = nullptr, *swk3_ = nullptr;      // swkPool1..3
    u64* c1_ = nullptr;                                      // ks.Pool[1]  (PolyQP)
    u64* polyq_[3] = {nullptr, nullptr, nullptr};            // polyQPool
    u64* invntt_ = nullptr;                                  // ks.PoolInvNTT
    u64* nttbuf_ = nullptr; size_t nttbuf_words_ = 0;        // tensor inputs in NTT form
    u64* ctbuf_ = nullptr;  size_t ctbuf_words_ = 0;         // rotate / rescale staging
    std::vector<Swk> hoist_pool_[5];                         // rlkSet.HoistPool[0/1] + h(t_i) of step F; [3],[4]: BFV HoistPool2[0/1]
    u64 *x2_ = nullptr, *y2_ = nullptr;                      // BFV swkPool4 / swkPool6
    u64* rbuf_ = nullptr;  size_t rbuf_words_ = 0;          // BFV: operands over R, their NTTs, tensor output
    u64* c1b_ = nullptr;   size_t c1b_words_ = 0;           // batched ks.Pool[1]
    u64* tbuf_ = nullptr;  size_t tbuf_words_ = 0;          // t_i of step F
    // mul_relin_rescale / mul_relin_batch: Rescale folded into the store of the last merged ModDown.  One entry per product that is not to be
    // written: its (unwritten) base, the rescaled output one level down, its polynomials; `done` once a launch has covered all of them
    struct RsMap { const u64* full; u64* out; int npolys; int out_limbs; bool done; };
    std::vector<RsMap> rs_maps_;
    u64* spreadbuf_ = nullptr; size_t spreadbuf_words_ = 0;  // N = 2^16: staging of the spread digits (decompose_batch), so that the sub-transforms run out of place
    u64* tens_ = nullptr;  size_t tens_words_ = 0;          // tensor term kept in the NTT domain (times P) for the merged E / F2 batch
    // key generation scratch: uploaded samples, gadget constants (slot 0: mkrlwe gadget, 1: caller's), permuted secret
    int32_t* kg_small_ = nullptr; u64 *kg_g_ = nullptr, *kg_sk_ = nullptr;
    void wipe_samples(size_t count);
    bool kg_ready_ = false;
    void kg_init();
    // out <- NTT(e_i) for beta_max error polynomials, then the combine pass (keygen_kernels.h); gadget: 0 none, 1 mkrlwe
    // (P on the limbs of digit i, keygen.go:288-323), 2 the constants last uploaded with kg_upload_g
    void kg_key(const int32_t* e, int gadget, const u64* skA, const u64* crs, const u64* skB, int sign, bool neg, u64* out);
    void kg_upload_g(const u64* g_plain);

    u64* scratch(u64*& p, size_t& have, size_t want);
    Swk& hoist_slot(int which, int idx);
    const int* map_qp(int level) const { return (masked_ ? d_map_own : d_map_qp) + (size_t)level * mtot; }
    int nslots_qp(int level) const { return masked_ ? own_cnt_[level] : level + 1 + np; }
    // slot ownership (limb sharding)
    bool masked_ = false;
    std::vector<char> own_;                                  // per modulus index < mtot
    std::vector<std::vector<int>> own_list_;                 // per level: owned active modulus indices (Q limbs first)
    std::vector<int> own_cnt_, ownq_;                        // per level: size of own_list_; owned Q limbs, ascending
    int *d_map_own = nullptr, *d_ownq = nullptr;
    int nq_owned(int level) const { int c = 0; for (int l : ownq_) if (l <= level) ++c; return c; }
    void slots_q_owned(NttBatch& b, int L) const;            // plain polynomials: owned Q limbs < L
    void zero_unowned(u64* base, int npolys, long poly_stride, int first_mod, int nlimbs);
    std::vector<ExtItem> lsh_items_;
    void check_level(int level) const;
    void slots_qp(NttBatch& b, int level) const;
    void slots_range(NttBatch& b, int mod_base, int limbs) const;
    void ntt_fwd_launch(const NttBatch& b, bool decompose);
    void ntt_inv_launch(NttBatch& b);       // fills in the tables of the H16-class inverse kernel, then launch_ntt_inv
    std::vector<unsigned char> small_q_;                     // per modulus: 34q < 2^63
    std::vector<unsigned char> small16_;                     // per modulus: 48q < 2^62 -- the short class of the H16 / H32 kernels (ntt16_kernels.hip)
    void ext_core(int level, const u64* ah, const u64* bg, u64* c, bool accumulate);

    struct MrPlan {
        bool valid = false;
        int level = 0, L = 0, n0 = 0, n1 = 0, nout = 0;
        std::vector<int> slot0, slot1;
        std::vector<const u64*> h0, h1;
        bool own0 = false, own1 = false;     // hoisted digits computed by the engine itself
        bool x_pending = false;              // x still running on the side stream (chain 2)
        const u64* tens = nullptr;           // the tensor term stays in the NTT domain (times P) and joins the E / F2 batch
        bool head_done = false;              // mr_finish_head ran, mr_finish_tail still to come
        std::vector<const u64*> xkeys;       // non-empty: x is produced by the F1 kernel of mr_finish_head (into xfused) instead of by mr_xy
        u64* xfused = nullptr;
        std::vector<const u64*> ykeys;       // non-empty: y is computed inside the F1 kernel from these keys (b_j) and h1 (never stored)
        bool f2_staged = false;              // the digits of the t_i were left after the cross stages (decompose_batch stage_only): the tail batch finishes them
        bool f2_fused = false;               // N = 2^15: no Decompose launch for the t_i at all -- the tail batch's product kernel is ntt16_f2_kernel
        const u64* f2_tbuf = nullptr;
        bool e_done = false;                 // ... and so was step E: its products sit in the c1 slots 2 n0 .. 2 n0 + n1 - 1 of the scratch, for the tail batch
    } plan_;

    // Stream-ordered buffer pool.  A buffer freed through this context may still be in use by kernels that ANOTHER context of the
    // same device enqueued (handles are shared freely between forked contexts).  Ordering between contexts is tracked with one
    // counter per context: seq_ counts the C-ABI calls that named the context (anything it may have enqueued), completed_ is seq_ at
    // its last host-side drain, and synced_ holds, per other context, the seq_ value this context's stream is known to be ordered
    // after (wait_for, fences).  A free records for which contexts that knowledge is behind at that moment; pool_alloc makes the
    // stream wait for exactly those (one event each) unless a wait_for / drain has caught up in the meantime.  The joins an
    // application needs for its own data flow therefore make the pool free of charge; single-context processes never fence.
    // Which contexts enqueued work on a handle's buffer is recorded in the handle (HandleUsers, filled by the C ABI on every call
    // that names the handle), so a temporary that never left its context costs nothing and never orders independent forks.
    struct FreeEntry { size_t words; u64* p; std::vector<std::pair<seq_t, seq_t>> behind; };      // (uid of a context, its seq_ at the free)
    std::vector<FreeEntry> free_list_;
    std::mutex pool_mu_;                                      // guards free_list_ (a trim may come from another context's thread); taken AFTER the registry mutex, never before
    hipEvent_t fence_ev_ = nullptr;
    void registry_add();
    void init(const u64* Q, const u64* P, const u64* psiQ, const u64* psiP, const u64* QMul, int nqm_, u64 T);     // the constructor's body
    void release_all() noexcept;                              // frees everything the context owns (destructor; constructor that throws)
    void registry_remove();
    seq_t uid_ = 0;
    std::atomic<seq_t> seq_{1}, completed_{0};
    std::atomic<seq_t> enqueued_{0};                         // seq_ as of the last C-ABI call that has FINISHED enqueuing its work: the clock an event
                                                             // recorded on this context's stream by ANOTHER thread may claim to cover (seq_ runs ahead of it
                                                             // while a call is in flight)
    std::atomic<bool> external_{false};                      // stream handed out (mkhe_ctx_stream): work may arrive without a C-ABI call
    std::vector<std::pair<seq_t, seq_t>> synced_;            // (uid, seq_) pairs, guarded by the registry mutex
    seq_t synced_with(seq_t uid) const { for (auto& e : synced_) if (e.first == uid) return e.second; return 0; }
    void set_synced(seq_t uid, seq_t v) { for (auto& e : synced_) if (e.first == uid) { if (e.second < v) e.second = v; return; } synced_.push_back({uid, v}); }
  public:
    void note_use(HandleUsers& u);                            // this context is about to enqueue work on the handle's buffer
    void touch() { seq_.fetch_add(1, std::memory_order_relaxed); }
    void mark_enqueued() { const seq_t s = seq_.load(); if (enqueued_.load() < s) enqueued_.store(s); }
    void mark_external() { external_.store(true); }
    seq_t now_seq() { return external_.load() ? seq_.fetch_add(1) + 1 : seq_.load(); }
  private:

    struct ProfRec { hipEvent_t e0, e1; int cls; double bytes; };
  public:
    // MKHE_NTT32=2 (default): which forward kernel a launch shape takes when both apply (H32 and H16 give the same bits), measured in the caller's own
    // workload.  The single-pass kernel wins by 2-5 % inside a MulRelin on most parts and loses 6 % on some (same library, same call:
    // profiles/README.md) -- back to back it is ahead on all of them, so a calibration outside the workload would pick wrongly; so did a first version
    // that alternated the kernels over the first dozen launches (the first milliseconds run on ramping clocks with power to spare), and a second that
    // timed the kernel alone (with the side stream on, a kernel's duration says little about the operation's).  What is measured is the PERIOD of the
    // shape -- the time from one of its launches to the next on the GPU's clock, i.e. the whole operation around it in a running service.  Per shape:
    // WARM launches on H16, then a block on H32 and a block on H16 (SETTLE launches, then TIMED periods); the medians decide, the choice stays.
    // A smaller shape follows the largest one unless the other kernel is 3 % ahead.
    struct NttTune { static constexpr int WARM = 64, SETTLE = 16, TIMED = 32, RING = 2 * (TIMED + 1); int n[2] = {0, 0}, req[2] = {0, 0}, blk[2] = {0, 0}; float t[2][TIMED];
                     int decided = -1, seen = 0;
                     // start events of the timed launches that have not been turned into periods yet (the host runs ahead of the GPU): ring[head .. head + inflight)
                     hipEvent_t e0[RING] = {}; int which[RING] = {}; int head = 0, inflight = 0; int slot = -1; };
    std::map<long, NttTune> ntt_tune_;
    int ntt_forced_ = -1;                            // mkhe_ctx_set_ntt_choice(limbs <= 0): the answer for every shape that has no entry of its own
    long batch_lanes_min_ = 1536;      // mkhe_ctx_set_batch_lanes: hoisted limb-NTTs per input from which mul_relin_batch runs its inputs in flight (< 0: never)
    void ntt_reset(NttTune& t, int choice);          // pins a shape (0 / 1) or makes it measure again (-1); samples in flight are dropped
    int ntt_pick(long key, NttTune*& sampling);      // 1 = H32, 0 = H16; sampling != nullptr: this launch is timed (record e0[slot] in front of it)
  private:
    bool prof_on_ = false;
    std::vector<ProfRec> prof_recs_;
    std::vector<hipEvent_t> prof_pool_;
    hipEvent_t prof_event();
    hipStream_t s_ = nullptr;                                // active stream of the launch helpers
    hipEvent_t ev_[8] = {};                                  // fork/join events of the side stream
    hipEvent_t xev_ = nullptr;                               // wait_for
    // mul_relin_batch at N >= 2^15: the B evaluations go through the single-operation path on this context and two internal ones, round robin -- in
    // flight side by side (the latency-bound stretches of one beside the other's kernels) instead of in lock step (batch.hip)
    std::vector<std::unique_ptr<Context>> lanes_;
    Context* lane(int i);
    bool batch_lanes_ok(size_t B, long limbs) const;
    void fork_side(int k);      // side stream waits for everything enqueued so far on the active stream
    void side_done(int k);      // marks the end of side chain k
    void join_side(int k);      // active stream waits for side chain k
  public:
    struct ProfScope {
        Context* c; size_t idx; bool on; hipStream_t st;
        ProfScope(Context* c_, int cls, double bytes);
        ~ProfScope();
    };
};

}  // namespace mkhe
