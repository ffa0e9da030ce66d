// batch.hip -- B independent evaluator operations of ONE shape as one launch set (round 4).
//
// The reference's small rings -- PN14QP439, its first benchmark set (mkckks/mkckks_benchmark_test.go:13), and cnn's PN14QP433
// (cnn/cnn_test.go:80-96) -- give an operation a few dozen 2^14-point limbs per launch: a fraction of the 256 CUs, and an encrypted
// inference is about 360 DEPENDENT launches whose issue, not their work, sets its 2.9 ms.  Nothing shortens that chain;
// what a server evaluating the same circuit on many inputs can do is run B of them in lock step: every kernel of the engine already takes
// LISTS -- gadget decompositions (NttBatch items), external products (ExtItem), ModDown destinations, ciphertext components (CtBinArgs) --
// so the B operations of one step are the same launches with B times the items, and the launch count per input falls by almost B.
//
// All B operations of a call have the same shape: the same id lists, limb counts and keys (the relinearization / rotation keys and the CRS
// belong to the parties, not to the inputs).  An operand that is the SAME ciphertext for every input (the model of cnn: kernels, weights,
// biases) is passed B times.  Results are the integers of the single-operation entry points, bit for bit (tests/test_gpu_batch.py): the
// batch path takes no shortcut that changes a representative.  The fusions of the single-operation path carry over per input: the tensor term
// rides on the first product of every output slot, the Rescale on the merged ModDown's store (Context::rs_maps_), the diagonal of engine-hoisted
// digits is the tensor input, and x_b comes out of input b's step F1 (up to four parties per operand: Context::ext_xmap_).
//
// Replaces, B at a time: KeySwitcher.MulAndRelin[Hoisted] (mkrlwe/keyswitch_hoisted.go:44-179) + Rescale (mkckks/evaluator.go:558-581),
// RotateHoisted / Rotate (:183-247, keyswitch.go:234-298), Evaluator.HoistedForm (mkckks/evaluator.go:543-553), AddNew / SubNew (:316-356).
#include "engine.h"
#include <algorithm>

namespace mkhe {

namespace {
// one pooled block carved into the temporaries of a batched call (returned to the context's stream-ordered pool at the end)
struct Arena {
    Context* c; u64* base = nullptr; size_t words = 0, used = 0;
    Arena(Context* c_, size_t w) : c(c_), words(w) { if (w) base = c->pool_alloc(w); }
    ~Arena() { if (base) { const HandleUsers none; c->pool_free(base, words, &none); } }
    u64* take(size_t w) { if (used + w > words) throw Error("mkhe: internal: batch arena overrun"); u64* p = base + used; used += w; return p; }
};
void same_shape(const std::vector<const Ct*>& v, const char* what) {
    for (const Ct* c : v) {
        if (!c) throw Error(std::string("mkhe: null ciphertext in a batch (") + what + ")");
        if (c->n != v[0]->n || c->limbs != v[0]->limbs || c->ids != v[0]->ids) throw Error(std::string("mkhe: the ciphertexts of a batch must have one shape (") + what + ")");
    }
}
// Aliasing rule of the batch entry points (include/mkhe.h): output k may be input k of the same call where the single-operation entry point
// allows in-place evaluation; it must never be an input of ANOTHER item -- the items of a batch run in one launch set, in no order.
void no_cross_alias(const std::vector<const Ct*>& ins, const std::vector<Ct*>& outs, const char* what) {
    for (size_t k = 0; k < outs.size(); ++k)
        for (size_t j = 0; j < ins.size(); ++j)
            if (j != k && ins[j] && outs[k] && ins[j]->d == outs[k]->d)
                throw Error(std::string("mkhe: output ") + std::to_string(k) + " of the batch is input " + std::to_string(j) + " of another item (" + what + ")");
}
}  // namespace

void Context::hoisted_form_batch(int level, const std::vector<const Ct*>& cts, const std::vector<Swk*>& outs) {
    if (cts.empty()) return;
    same_shape(cts, "HoistedForm");
    const int n = cts[0]->n;
    if (cts[0]->limbs < level + 1) throw Error("mkhe: ciphertext level below requested level");
    if (outs.size() != cts.size() * (size_t)n) throw Error("mkhe: hoisted_form_batch: one output per party component");
    std::vector<const u64*> src; std::vector<u64*> dst;
    const size_t PI = (size_t)cts[0]->limbs * N;
    for (size_t b = 0; b < cts.size(); ++b)
        for (int a = 0; a < n; ++a) { src.push_back(cts[b]->d + (1 + a) * PI); dst.push_back(outs[b * n + a]->d); }
    if (!src.empty()) decompose_batch(level, src, dst, false);
}

void Context::rotate_batch(u64 galEl, const std::vector<const Ct*>& ins, const std::vector<const Swk*>& hoists, const Swk* const* rk,
                           const Swk& crs, const std::vector<Ct*>& outs) {
    const size_t B = ins.size();
    if (!B) return;
    if (outs.size() != B) throw Error("mkhe: rotate_batch: one output per input");
    same_shape(ins, "Rotate");
    { std::vector<const Ct*> o(outs.begin(), outs.end()); same_shape(o, "Rotate outputs"); }
    const int n = ins[0]->n, L = outs[0]->limbs, level = L - 1;
    check_level(level);
    if (ins[0]->limbs < L) throw Error("Cannot Rotate: ctIn and ctOut have different levels");
    if (outs[0]->n != n || outs[0]->ids != ins[0]->ids) throw Error("mkhe: ctOut must carry the ids of ctIn");
    if (!hoists.empty() && hoists.size() != B * (size_t)n) throw Error("mkhe: rotate_batch: one hoisted form per party component");
    no_cross_alias(ins, outs, "Rotate");
    bool alias = false;
    for (size_t b = 0; b < B; ++b) alias = alias || ins[b]->d == outs[b]->d;
    if (n == 0 || 2 * n > EXT_MAX_ITEMS || alias || galEl == 0) {
        // (no party, or the fused permutation does not apply: one operation at a time)
        for (size_t b = 0; b < B; ++b) {
            std::vector<const Swk*> h;
            for (int a = 0; a < n && !hoists.empty(); ++a) h.push_back(hoists[b * n + a]);
            rotate(galEl, *ins[b], hoists.empty() ? nullptr : h.data(), rk, crs, *outs[b]);
        }
        return;
    }
    for (int a = 0; a < n; ++a) if (!rk[a]) throw Error("cannot GetRotationKeys: there is no rotation key with given id");
    const size_t PI = (size_t)ins[0]->limbs * N, PO = (size_t)L * N;
    Arena ar(this, hoists.empty() ? B * n * swk_words() : 0);
    std::vector<const u64*> h(B * n);
    bool stage = false;
    if (hoists.empty()) {
        std::vector<const u64*> dsrc; std::vector<u64*> ddst;
        for (size_t b = 0; b < B; ++b)
            for (int a = 0; a < n; ++a) { u64* d = ar.take(swk_words()); dsrc.push_back(ins[b]->d + (1 + a) * PI); ddst.push_back(d); h[b * n + a] = d; }
        // (digits the engine made for itself, read once by the two products of their party: a small launch finishes the transform inside the product kernel)
        stage = ext_fused_ok(level, (int)(B * n));
        decompose_batch(level, dsrc, ddst, true, stage);
    } else {
        for (size_t i = 0; i < B * n; ++i) { if (!hoists[i]) throw Error("mkhe: missing hoisted form"); h[i] = hoists[i]->d; }
    }
    // whole ciphertexts per ext_batch call: the products of one destination must meet in one ModDown launch (the fused permutation)
    const size_t per = std::max<size_t>(1, EXT_MAX_ITEMS / (2 * n));
    for (size_t b0 = 0; b0 < B; b0 += per) {
        std::vector<ExtItem> items;
        for (size_t b = b0; b < std::min(B, b0 + per); ++b)
            for (int a = 0; a < n; ++a) {
                items.push_back(ExtItem{h[b * n + a], rk[a]->d, outs[b]->d, true});
                if (a == 0) items.back().addend = ins[b]->d;
                items.push_back(ExtItem{h[b * n + a], crs.d, outs[b]->d + (size_t)(1 + a) * PO, false});
            }
        if (stage) ext_staged_.assign(h.begin() + b0 * n, h.begin() + std::min(B, b0 + per) * n);
        try { ext_batch(level, items, -1, 0, galEl); } catch (...) { ext_staged_.clear(); staged_open_.clear(); throw; }
        ext_staged_.clear();
    }
    MKHE_HIP(hipGetLastError());
}

// B rotations of B ciphertexts of ONE shape in one launch set, each by its own Galois element with its own rotation keys and CRS -- the independent
// rotate -> hoist -> MulRelin chains of cnn's Convolution and FC1Layer (cnn/cnn.go:16-30,51-62) are lanes of one batch instead of chains on forked
// contexts (round 5: a small kernel does not overlap another one on this chip; 8 chains were 8 times the launches) -- and optionally
// out[b] = post_add[b] + Rotate(in[b]): the AddNew that follows every RotateNew of the log-sums (cnn.go:33-37,64-67,83-86,90-93) rides on the store.
void Context::rotate_multi(const std::vector<u64>& galEl, const std::vector<const Ct*>& ins, const std::vector<const Swk*>& hoists, const std::vector<const Swk*>& rk,
                           const std::vector<const Swk*>& crs, const std::vector<const Ct*>& post_add, const std::vector<Ct*>& outs) {
    const size_t B = ins.size();
    if (!B) return;
    if (outs.size() != B || galEl.size() != B || crs.size() != B) throw Error("mkhe: rotate_multi: one Galois element, one CRS and one output per input");
    if (!post_add.empty() && post_add.size() != B) throw Error("mkhe: rotate_multi: one addend per input (or none)");
    same_shape(ins, "Rotate");
    { std::vector<const Ct*> o(outs.begin(), outs.end()); same_shape(o, "Rotate outputs"); }
    const int n = ins[0]->n, L = outs[0]->limbs, level = L - 1;
    check_level(level);
    if (ins[0]->limbs < L) throw Error("Cannot Rotate: ctIn and ctOut have different levels");
    if (outs[0]->n != n || outs[0]->ids != ins[0]->ids) throw Error("mkhe: ctOut must carry the ids of ctIn");
    if (rk.size() != B * (size_t)n) throw Error("mkhe: rotate_multi: one rotation key per input and party");
    if (!hoists.empty() && hoists.size() != B * (size_t)n) throw Error("mkhe: rotate_multi: one hoisted form per party component");
    no_cross_alias(ins, outs, "Rotate");
    for (size_t b = 0; b < B; ++b) {
        if (galEl[b] == 0 || galEl[b] >= (2ull << logN) || !(galEl[b] & 1)) throw Error("mkhe: rotate_multi: bad Galois element");
        if (!crs[b]) throw Error("mkhe: rotate_multi: missing CRS");
        if (!post_add.empty()) {
            const Ct* p = post_add[b];
            if (!p || p->n != n || p->ids != ins[0]->ids || p->limbs < L) throw Error("mkhe: rotate_multi: the addend must carry the ids of ctIn at ctOut's level or above");
            for (size_t k = 0; k < B; ++k) if (p->d == outs[k]->d) throw Error("mkhe: rotate_multi: an output aliases an addend");
        }
        for (int a = 0; a < n; ++a) if (!rk[b * n + a]) throw Error("cannot GetRotationKeys: there is no rotation key with given id");
    }
    bool alias = false;
    for (size_t b = 0; b < B; ++b) alias = alias || ins[b]->d == outs[b]->d;
    if (n == 0 || 2 * n > EXT_MAX_ITEMS || alias) {
        // (no party component, or in place: one operation at a time, then the addition)
        for (size_t b = 0; b < B; ++b) {
            std::vector<const Swk*> h, r(rk.begin() + b * n, rk.begin() + (b + 1) * n);
            for (int a = 0; a < n && !hoists.empty(); ++a) h.push_back(hoists[b * n + a]);
            if (n == 0) automorphism(galEl[b], *ins[b], *outs[b]);
            else rotate(galEl[b], *ins[b], hoists.empty() ? nullptr : h.data(), r.data(), *crs[b], *outs[b]);
            if (!post_add.empty()) ct_binary(0, *post_add[b], *outs[b], *outs[b]);
        }
        return;
    }
    const size_t PI = (size_t)ins[0]->limbs * N, PO = (size_t)L * N;
    Arena ar(this, hoists.empty() ? B * n * swk_words() : 0);
    std::vector<const u64*> h(B * n);
    bool stage = false;
    if (hoists.empty()) {
        std::vector<const u64*> dsrc; std::vector<u64*> ddst;
        for (size_t b = 0; b < B; ++b)
            for (int a = 0; a < n; ++a) { u64* d = ar.take(swk_words()); dsrc.push_back(ins[b]->d + (1 + a) * PI); ddst.push_back(d); h[b * n + a] = d; }
        // (digits the engine made for itself, read once by the two products of their party: a small launch finishes the transform inside the product kernel)
        stage = ext_fused_ok(level, (int)(B * n));
        decompose_batch(level, dsrc, ddst, true, stage);
    } else {
        for (size_t i = 0; i < B * n; ++i) { if (!hoists[i]) throw Error("mkhe: missing hoisted form"); h[i] = hoists[i]->d; }
    }
    // whole ciphertexts per ext_batch call: the products of one destination meet in one ModDown launch (the fused permutation)
    const size_t per = std::max<size_t>(1, EXT_MAX_ITEMS / (2 * n));
    for (size_t b0 = 0; b0 < B; b0 += per) {
        std::vector<ExtItem> items;
        for (size_t b = b0; b < std::min(B, b0 + per); ++b) {
            const size_t PP = post_add.empty() ? 0 : (size_t)post_add[b]->limbs * N;
            for (int a = 0; a < n; ++a) {
                items.push_back(ExtItem{h[b * n + a], rk[b * n + a]->d, outs[b]->d, true});
                if (a == 0) items.back().addend = ins[b]->d;
                items.back().gal = (unsigned)galEl[b];
                if (PP) items.back().post = post_add[b]->d;
                items.push_back(ExtItem{h[b * n + a], crs[b]->d, outs[b]->d + (size_t)(1 + a) * PO, false});
                items.back().gal = (unsigned)galEl[b];
                if (PP) items.back().post = post_add[b]->d + (size_t)(1 + a) * PP;
            }
        }
        if (stage) ext_staged_.assign(h.begin() + b0 * n, h.begin() + std::min(B, b0 + per) * n);
        try { ext_batch(level, items, -1, 0, 0); } catch (...) { ext_staged_.clear(); staged_open_.clear(); throw; }
        ext_staged_.clear();
    }
    MKHE_HIP(hipGetLastError());
}

#ifndef MKHE_BATCH_LANES
#define MKHE_BATCH_LANES 2                 // internal contexts beside the caller's (0: lock step on every ring, as rounds 4-5)
#endif
Context* Context::lane(int i) {
    while ((int)lanes_.size() <= i) {
        const u64* Q = moduli.data(); const u64* P = Q + nq;
        std::unique_ptr<Context> c(new Context(logN, Q, nq, P, np, gamma, psi_plain.data(), psi_plain.data() + nq, device,
                                               nqm ? Q + nq + np : nullptr, nqm, bfv_t));
        lanes_.push_back(std::move(c));
    }
    return lanes_[i].get();
}
// -- for evaluations whose big kernels fill the chip by themselves.  Measured on PN15QP880 (profiles/r6_batch_lanes.txt): four parties (1792 hoisted
// limb-NTTs per input) in flight 1350-1381 MulRelin/s against 1285-1350 in lock step and 1340 one at a time; two parties (896) and one (448) are
// launch sets that lock step still merges to advantage (2478-2764 against 2418-2514; 4282-5096 against 3973-4371): the threshold sits between.
bool Context::batch_lanes_ok(size_t B, long limbs) const {
    if (MKHE_BATCH_LANES < 1 || logN != 15 || alpha != 1 || B < 2 || masked_ || batch_lanes_min_ < 0 || limbs < batch_lanes_min_) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cs);
    return cs == hipStreamCaptureStatusNone;          // (a capture records the lock-step form: one stream pair)
}
void Context::mul_relin_batch(const std::vector<const Ct*>& op0, const std::vector<const Ct*>& op1, const std::vector<const Swk*>& hoist0,
                              const std::vector<const Swk*>& hoist1, const Swk* const* rlk_b1, const Swk* const* rlk_d0, const Swk* const* rlk_v0,
                              const Swk& crs_u, bool rescale_out, const std::vector<Ct*>& outs) {
    const size_t B = op0.size();
    if (!B) return;
    if (op1.size() != B || outs.size() != B) throw Error("mkhe: mul_relin_batch: one op1 and one output per op0");
    if (masked_) throw Error("mkhe: a limb-sharded context evaluates one operation at a time");
    same_shape(op0, "MulRelin op0"); same_shape(op1, "MulRelin op1");
    no_cross_alias(op0, outs, "MulRelin op0"); no_cross_alias(op1, outs, "MulRelin op1");
    { std::vector<const Ct*> o(outs.begin(), outs.end()); same_shape(o, "MulRelin outputs"); }
    const Ct& o0 = *outs[0];
    const int L = o0.limbs + (rescale_out ? 1 : 0), level = L - 1, n0 = op0[0]->n, n1 = op1[0]->n, nout = o0.n;
    check_level(level);
    if (rescale_out && (o0.limbs < 1 || L > nq)) throw Error("cannot Rescale: input Ciphertext already at level 0");
    if (op0[0]->limbs < L || op1[0]->limbs < L) throw Error("Cannot MulAndRelin: op0 and op1 have different levels");
    if (n0 > 32 || n1 > 32 || nout > 32) throw Error("mkhe: too many parties");
    if (!hoist0.empty() && hoist0.size() != B * (size_t)n0) throw Error("mkhe: mul_relin_batch: one hoisted form per party component of op0");
    if (!hoist1.empty() && hoist1.size() != B * (size_t)n1) throw Error("mkhe: mul_relin_batch: one hoisted form per party component of op1");
    // out ids = the union of the operand id sets (newCiphertextBinary, mkckks/evaluator.go:306-313)
    std::vector<int> slot0(n0), slot1(n1);
    {
        auto find = [&](int id) { for (int o = 0; o < nout; ++o) if (o0.ids[o] == id) return o; return -1; };
        std::vector<char> seen(nout, 0);
        for (int a = 0; a < n0; ++a) { const int o = find(op0[0]->ids[a]); if (o < 0) throw Error("mkhe: ctOut lacks an id of op0"); slot0[a] = o; seen[o] = 1; }
        for (int a = 0; a < n1; ++a) { const int o = find(op1[0]->ids[a]); if (o < 0) throw Error("mkhe: ctOut lacks an id of op1"); slot1[a] = o; seen[o] = 1; }
        for (int o = 0; o < nout; ++o) if (!seen[o]) throw Error("mkhe: ctOut has an id that neither operand has");
    }
    for (int a = 0; a < n0; ++a) if (!rlk_d0[a] || !rlk_v0[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
    for (int a = 0; a < n1; ++a) if (!rlk_b1[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
    // N >= 2^15: one evaluation fills the chip with every big kernel, and B of them in lock step only lengthen those kernels (PN15QP880, four parties:
    // 1266 / 1311 MulRelin/s at B = 2 / 4 against 1340 one at a time, round 5) -- what a second evaluation can use is the latency-bound stretches of the
    // first (small inverse NTTs, ModDowns).  So the B evaluations run through the single-operation path, round robin on this context and two internal
    // ones: in flight side by side, + 4-9 % over one at a time (profiles/r6_cu_partition.txt, r6_batch_lanes.txt).  Same integers: it IS the single path.
    if (batch_lanes_ok(B, (long)(n0 + n1) * beta(level) * nslots_qp(level))) {
        const int NL = (int)std::min<size_t>(B, 1 + MKHE_BATCH_LANES);
        for (int l = 1; l < NL; ++l) { Context* c = lane(l - 1); c->overlap = overlap; c->ntt_forced_ = ntt_forced_; c->wait_for(*this); }      // the inputs (and earlier readers of the outputs) are ordered on this stream
        try {
            for (size_t b = 0; b < B; ++b) {
                Context* c = (b % NL) ? lane((int)(b % NL) - 1) : this;
                const Swk* const* hb0 = hoist0.empty() ? nullptr : hoist0.data() + b * n0;
                const Swk* const* hb1 = hoist1.empty() ? nullptr : ((!hoist0.empty() && op0[b] == op1[b] && std::equal(hoist0.begin() + b * n0, hoist0.begin() + (b + 1) * n0, hoist1.begin() + b * n1)) ? hb0 : hoist1.data() + b * n1);
                if (rescale_out) c->mul_relin_rescale(*op0[b], *op1[b], hb0, hb1, rlk_b1, rlk_d0, rlk_v0, crs_u, *outs[b]);
                else c->mul_and_relin(*op0[b], *op1[b], hb0, hb1, rlk_b1, rlk_d0, rlk_v0, crs_u, *outs[b]);
            }
        } catch (...) { for (auto& l : lanes_) l->recover(); throw; }
        for (int l = 1; l < NL; ++l) wait_for(*lane(l - 1));
        MKHE_HIP(hipGetLastError());
        return;
    }
    bool same = hoist0.size() == hoist1.size();          // squares: op1 IS op0 (and its hoisted forms): hoist once
    for (size_t b = 0; b < B && same; ++b) same = op0[b] == op1[b];
    for (size_t i = 0; i < hoist0.size() && same; ++i) same = hoist0[i] == hoist1[i];
    const bool own0 = hoist0.empty(), own1 = hoist1.empty() && !same;
    const size_t P0 = (size_t)op0[0]->limbs * N, P1 = (size_t)op1[0]->limbs * N, PO = (size_t)L * N, SW = swk_words();
    const bool fold = n0 >= 1 && 2 * n0 + n1 <= EXT_MAX_ITEMS && ext_merge_members(level) >= 2;
    const int fuse_env = ab_fuse_x();
    const bool fuse_x = fuse_env && n0 >= 1 && n0 <= 4;      // x_b = sum_i d_i (.) h(c0_{b,i}) as a by-product of input b's step F1
    const int fuse_y_env = ab_fuse_y();
    const bool fuse_y = fuse_x && fuse_y_env && n1 >= 1 && n1 <= 4;    // ... and y_b computed in the same threads (ext_inner_xy_batch_kernel<G0, G1>), never stored
    const int fuse_e_env = ab_fuse_e();
    const bool fuse_e = fuse_y && fuse_e_env;                          // ... and step E: input b's <h(c1_j), x_b> as precomputed items of the tail batch
    const size_t per_b = (size_t)(2 + n0 + n1) * PO + (size_t)(1 + nout) * PO * ((fold ? 1 : 0) + (rescale_out ? 1 : 0)) + 2 * SW + (size_t)n0 * PO + (size_t)n0 * SW +
                         (own0 ? (size_t)n0 * SW : 0) + (own1 ? (size_t)n1 * SW : 0) + (fuse_e ? (size_t)n1 * mtot * N : 0);
    Arena ar(this, B * per_b);
    std::vector<u64*> nb_(B), tens(B, nullptr), full(B), x(B), y(B), tbuf(B), epre(B, nullptr);
    std::vector<const u64*> h0(B * n0), h1(B * n1);
    std::vector<u64*> h2(B * n0);
    {
        std::vector<const u64*> dsrc; std::vector<u64*> ddst;
        for (size_t b = 0; b < B; ++b) {
            nb_[b] = ar.take((size_t)(2 + n0 + n1) * PO);
            if (fold) tens[b] = ar.take((size_t)(1 + nout) * PO);
            full[b] = rescale_out ? ar.take((size_t)(1 + nout) * PO) : outs[b]->d;
            x[b] = ar.take(SW); y[b] = ar.take(SW); tbuf[b] = ar.take((size_t)n0 * PO);
            if (fuse_e) epre[b] = ar.take((size_t)n1 * mtot * N);
            for (int a = 0; a < n0; ++a) h2[b * n0 + a] = ar.take(SW);
            for (int a = 0; a < n0; ++a) {
                if (own0) { u64* d = ar.take(SW); dsrc.push_back(op0[b]->d + (1 + a) * P0); ddst.push_back(d); h0[b * n0 + a] = d; }
                else { if (!hoist0[b * n0 + a]) throw Error("mkhe: missing hoisted form"); h0[b * n0 + a] = hoist0[b * n0 + a]->d; }
            }
            for (int a = 0; a < n1; ++a) {
                if (same) h1[b * n1 + a] = h0[b * n0 + a];
                else if (own1) { u64* d = ar.take(SW); dsrc.push_back(op1[b]->d + (1 + a) * P1); ddst.push_back(d); h1[b * n1 + a] = d; }
                else { if (!hoist1[b * n1 + a]) throw Error("mkhe: missing hoisted form"); h1[b * n1 + a] = hoist1[b * n1 + a]->d; }
            }
        }
        // hoisting of whatever the caller did not supply (MulRelinNew, mkckks/evaluator.go:416-443): engine-internal digits
        if (!dsrc.empty()) decompose_batch(level, dsrc, ddst, true);
    }
    // ---- step D: tensor product in the NTT domain (keyswitch_hoisted.go:119-144): NTT of every component of both operands, one launch per side
    // (alpha = 1: digit l of an engine-hoisted component under its own modulus l IS NTT_l of its limb l -- the diagonal h[l][l], limb stride
    // (nQ + nP + 1) N -- so that only c_0 of such an operand is transformed; caller-supplied hoisted forms are not trusted with that, as in mr_prepare)
    const bool diag0 = own0 && alpha == 1, diag1 = (own1 || (same && own0)) && alpha == 1;
    {
        const int nn0 = diag0 ? 0 : n0, nn1 = diag1 ? 0 : n1;
        // (both sides as the items of ONE launch when they have one shape: twice the limbs per launch -- on the small rings that is what
        // takes it over the threshold of the H16-class kernel, one launch for both modulus classes instead of four of the round-1 kernel)
        const bool both = !same && P0 == P1 && nn0 == nn1;
        struct It { const u64* s; u64* d; };
        for (int side = 0; side < (same || both ? 1 : 2); ++side) {
            const int n = side ? nn1 : nn0;
            std::vector<It> its;
            for (size_t b = 0; b < B; ++b) {
                if (both || side == 0) its.push_back(It{op0[b]->d, nb_[b]});
                if (both || side == 1) its.push_back(It{op1[b]->d, nb_[b] + (size_t)(1 + n0) * PO});
            }
            for (size_t base = 0; base < its.size(); base += NTT_MAX_ITEMS) {
                const int cnt = (int)std::min<size_t>(NTT_MAX_ITEMS, its.size() - base);
                NttBatch b{};
                b.mods = d_mods; b.psi = d_psi; b.aux = d_inv_aux; slots_q_owned(b, L);
                b.src_inner = b.dst_inner = N; b.src_outer = (long)(side ? P1 : P0); b.dst_outer = (long)PO;
                b.nitems = cnt; b.outers_per_item = 1 + n; b.nouter = cnt * (1 + n);
                for (int i = 0; i < cnt; ++i) { b.src_items[i] = its[base + i].s; b.dst_items[i] = its[base + i].d; }
                ntt_fwd_launch(b, false);
            }
        }
    }
    {
        // (the temporaries of the inputs sit at one stride in the arena: ONE tensor launch for the batch when the term stays in the NTT domain)
        const long stride = B > 1 ? (long)(nb_[1] - nb_[0]) : 0;
        const bool one = fold && B > 1 && (rescale_out || true);
        for (size_t b = 0; b < (one ? 1 : B); ++b) {
            TensorArgs ta{};
            ta.a0 = nb_[b]; ta.b0 = same ? nb_[b] : nb_[b] + (size_t)(1 + n0) * PO; ta.out = fold ? tens[b] : full[b]; ta.mods = d_mods;
            if (fold) ta.scale = d_pmodq;
            ta.nout = nout; ta.L = L; ta.N = N; ta.with_c0 = 1;
            if (one) { ta.nbatch = (int)B; ta.in_batch = stride; ta.out_batch = (long)(tens[1] - tens[0]); }
            const long dstride = (long)(mtot + 1) * N;
            for (int a = 0; a < n0; ++a) {
                const int o = 1 + slot0[a];
                if (diag0) { ta.a[o] = h0[b * n0 + a]; ta.a_ls[o] = dstride; } else { ta.a[o] = nb_[b] + (size_t)(1 + a) * PO; ta.a_ls[o] = N; }
            }
            for (int a = 0; a < n1; ++a) {
                const int o = 1 + slot1[a];
                if (diag1) { ta.b[o] = h1[b * n1 + a]; ta.b_ls[o] = dstride; }
                else { ta.b[o] = (same ? nb_[b] : nb_[b] + (size_t)(1 + n0) * PO) + (size_t)(1 + a) * PO; ta.b_ls[o] = N; }
            }
            { ProfScope ps(this, PROF_TENSOR, 8.0 * N * L * (2.0 + n0 + n1 + 1 + nout) * (one ? B : 1)); launch_tensor(ta, s_); }
            if (!fold) ntt(full[b], full[b], 1 + nout, L, 0, true, false);
        }
    }
    // ---- steps B, C: x = MForm(sum_i d_i (.) h(c0_i)), y = MForm(sum_j b_j (.) h(c1_j))   (:79-117): one launch per side for up to sixteen inputs
    {
        const int nbt = beta(level), nslots = nslots_qp(level);
        if (n0 > MAX_TERMS || n1 > MAX_TERMS) throw Error("mkhe: too many parties");
        for (int side = fuse_y ? 0 : 1; side >= (fuse_x ? 1 : 0); --side) {
            const int n = side ? n1 : n0;
            if (n == 0) { for (size_t b = 0; b < B; ++b) MKHE_HIP(hipMemsetAsync(side ? y[b] : x[b], 0, SW * sizeof(u64), s_)); continue; }
            if (n <= IPB_MAX_TERMS) {
                for (size_t b0 = 0; b0 < B; b0 += IPB_MAX_BATCH) {
                    const int cnt = (int)std::min<size_t>(IPB_MAX_BATCH, B - b0);
                    InnerProductBatchArgs ip{};
                    for (int a = 0; a < n; ++a) ip.a[a] = (side ? rlk_b1[a] : rlk_d0[a])->d;
                    for (int k = 0; k < cnt; ++k) {
                        for (int a = 0; a < n; ++a) ip.b[k][a] = side ? h1[(b0 + k) * n1 + a] : h0[(b0 + k) * n0 + a];
                        ip.out[k] = side ? y[b0 + k] : x[b0 + k];
                    }
                    ip.mods = d_mods; ip.map = map_qp(level); ip.term_outer = ip.out_outer = (long)mtot * N;
                    ip.nterms = n; ip.nslots = nslots; ip.nouter = nbt; ip.N = N; ip.nbatch = cnt; ip.mform_out = 1;
                    { ProfScope ps(this, PROF_INNER, 8.0 * N * nslots * nbt * (2.0 * n + 1) * cnt); launch_inner_product_batch(ip, s_); }
                }
                continue;
            }
            for (size_t b = 0; b < B; ++b) {
                InnerProductArgs ip{};
                for (int a = 0; a < n; ++a) { ip.a[a] = (side ? rlk_b1[a] : rlk_d0[a])->d; ip.b[a] = side ? h1[b * n1 + a] : h0[b * n0 + a]; }
                ip.out = side ? y[b] : x[b]; ip.mods = d_mods; ip.map = map_qp(level);
                ip.term_outer = ip.out_outer = (long)mtot * N; ip.nterms = n; ip.nslots = nslots; ip.nouter = nbt; ip.N = N; ip.mform_out = 1;
                { ProfScope ps(this, PROF_INNER, 8.0 * N * nslots * nbt * (2.0 * n + 1)); launch_inner_product(ip, s_); }
            }
        }
    }
    // ---- step F1: t_i = <h(c0_i), y>_P (:165-169), every input in one batch (whole inputs per launch: with fuse_x the thread that holds input
    // b's digits h(c0_i) multiplies them by the d_i as well and stores x_b); then h(t_i)
    if (n0) {
        const size_t per = std::max<size_t>(1, EXT_MAX_ITEMS / (size_t)n0);
        for (size_t b0 = 0; b0 < B; b0 += per) {
            std::vector<ExtItem> items;
            ext_xmap_.clear(); ext_ykeys_.clear(); ext_yh_.clear(); ext_eouts_.clear();
            for (size_t b = b0; b < std::min(B, b0 + per); ++b) {
                if (fuse_e) ext_eouts_.push_back(epre[b]);
                for (int a = 0; a < n0; ++a) {
                    items.push_back(ExtItem{h0[b * n0 + a], y[b], tbuf[b] + (size_t)a * PO, false});
                    if (fuse_x) items.back().xkey = rlk_d0[a]->d;
                }
                if (fuse_y) for (int a = 0; a < n1; ++a) ext_yh_.push_back(h1[b * n1 + a]);
                if (fuse_x) ext_xmap_.push_back({y[b], x[b]});
            }
            if (fuse_y) for (int a = 0; a < n1; ++a) ext_ykeys_.push_back(rlk_b1[a]->d);
            try { ext_batch(level, items); } catch (...) { ext_xmap_.clear(); ext_ykeys_.clear(); ext_yh_.clear(); ext_eouts_.clear(); throw; }
            ext_xmap_.clear(); ext_ykeys_.clear(); ext_yh_.clear(); ext_eouts_.clear();
        }
        std::vector<const u64*> dsrc; std::vector<u64*> ddst;
        for (size_t b = 0; b < B; ++b)
            for (int a = 0; a < n0; ++a) { dsrc.push_back(tbuf[b] + (size_t)a * PO); ddst.push_back(h2[b * n0 + a]); }
        decompose_batch(level, dsrc, ddst, true);
    }
    // ---- steps E, F2: out_j += <h(c1_j), x>_P ; out_0 += <h(t_i), v_i>_P ; out_i += <h(t_i), u>_P   (:146-154,173-177); whole operations per call.
    // With the tensor term folded in, every polynomial of a product is written exactly once by the merged ModDown of its call: the single
    // Rescale of mkckks.Evaluator.mulRelinHoisted rides on that store (Context::rs_maps_) and the product itself is never written.
    rs_maps_.clear();
    if (rescale_out && fold)
        for (size_t b = 0; b < B; ++b) rs_maps_.push_back(RsMap{full[b], outs[b]->d, 1 + nout, o0.limbs, false});
    if (2 * n0 + n1 > 0) {
        const size_t m = (size_t)(2 * n0 + n1), per = std::max<size_t>(1, EXT_MAX_ITEMS / m);
        for (size_t b0 = 0; b0 < B; b0 += per) {
            std::vector<ExtItem> items;
            for (size_t b = b0; b < std::min(B, b0 + per); ++b) {
                const size_t first = items.size();
                for (int a = 0; a < n0; ++a) {
                    items.push_back(ExtItem{h2[b * n0 + a], rlk_v0[a]->d, full[b], true});
                    items.push_back(ExtItem{h2[b * n0 + a], crs_u.d, full[b] + (size_t)(1 + slot0[a]) * PO, true});
                }
                for (int a = 0; a < n1; ++a) {
                    items.push_back(ExtItem{h1[b * n1 + a], x[b], full[b] + (size_t)(1 + slot1[a]) * PO, true});
                    if (fuse_e) { items.back().pre = true; items.back().pre_src = epre[b] + (size_t)a * mtot * N; }
                }
                if (fold) {
                    // the tensor term of every output slot (NTT domain, times P) rides on the first product that goes there: ModDown returns it as itself
                    std::vector<const u64*> seen;
                    for (size_t i = first; i < items.size(); ++i) {
                        if (std::find(seen.begin(), seen.end(), items[i].dst) != seen.end()) continue;
                        seen.push_back(items[i].dst);
                        items[i].accumulate = false; items[i].qadd = tens[b] + (items[i].dst - full[b]);
                    }
                    if ((int)seen.size() != 1 + nout) throw Error("mkhe: internal: an output slot without an external product");
                }
            }
            ext_batch(level, items);
        }
    }
    // ---- the single Rescale of mkckks.Evaluator.mulRelinHoisted (mkckks/evaluator.go:558-581)
    std::vector<char> rescaled(B, 0);
    for (size_t b = 0; b < rs_maps_.size(); ++b) rescaled[b] = rs_maps_[b].done ? 1 : 0;
    rs_maps_.clear();
    if (rescale_out)
        for (size_t b = 0; b < B; ++b)
            if (!rescaled[b]) launch_div_round_last(outs[b]->d, full[b], d_mods, d_rescale + (size_t)(level - 1) * nq, level, N, 1 + nout, (long)PO, (long)(level * (size_t)N), s_);
    MKHE_HIP(hipGetLastError());
}

void Context::ct_sum(const std::vector<const Ct*>& ins, Ct& out) {
    if (ins.empty()) throw Error("mkhe: ct_sum: nothing to add");
    same_shape(ins, "Sum");
    if (out.n != ins[0]->n || out.ids != ins[0]->ids) throw Error("mkhe: ctOut must carry the ids of the summands");
    if (out.limbs < 1 || out.limbs > ins[0]->limbs) throw Error("mkhe: operand level below ctOut level");
    const bool same_limbs = out.limbs == ins[0]->limbs;
    for (const Ct* c : ins) if (c->d == out.d && !same_limbs) throw Error("mkhe: ct_sum in place needs ctOut at the summands' level");
    const int L = out.limbs, npolys = 1 + out.n;
    const size_t PI = (size_t)ins[0]->limbs * N, PO = (size_t)L * N;
    // more than CTSUM_MAX summands: partial sums into out, which then joins the next group as its first summand
    size_t k0 = 0;
    bool have = false;
    while (k0 < ins.size()) {
        // (summands at a higher level than out: polynomial stride differs -> one launch per polynomial)
        const size_t take = std::min<size_t>(ins.size() - k0, (size_t)CTSUM_MAX - (have ? 1 : 0));
        for (int p = 0; p < (same_limbs ? 1 : npolys); ++p) {
            CtSumArgs a{};
            a.mods = d_mods; a.L = L; a.N = N; a.npolys = same_limbs ? npolys : 1;
            int c = 0;
            if (have) a.in[c++] = out.d + (size_t)p * PO;
            for (size_t k = k0; k < k0 + take; ++k) a.in[c++] = ins[k]->d + (size_t)p * PI;
            a.n = c; a.dst = out.d + (size_t)p * PO;
            ProfScope ps(this, PROF_OTHER, 8.0 * N * L * a.npolys * (c + 1));
            launch_ct_sum(a, s_);
        }
        have = true; k0 += take;
    }
    MKHE_HIP(hipGetLastError());
}

void Context::ct_binary_batch(int op, const std::vector<const Ct*>& a, const std::vector<const Ct*>& b, const std::vector<Ct*>& outs) {
    const size_t B = a.size();
    if (!B) return;
    if (b.size() != B || outs.size() != B) throw Error("mkhe: ct_binary_batch: one op1 and one output per op0");
    same_shape(a, "Add op0"); same_shape(b, "Add op1");
    no_cross_alias(a, outs, "Add op0"); no_cross_alias(b, outs, "Add op1");
    { std::vector<const Ct*> o(outs.begin(), outs.end()); same_shape(o, "Add outputs"); }
    const Ct& A = *a[0]; const Ct& Bc = *b[0]; const Ct& O = *outs[0];
    const int L = O.limbs;
    if (A.limbs < L || Bc.limbs < L) throw Error("mkhe: operand level below ctOut level");
    if (1 + O.n > CTBIN_MAX) throw Error("mkhe: too many parties in one ciphertext");
    const size_t PA = (size_t)A.limbs * N, PB = (size_t)Bc.limbs * N, PO = (size_t)L * N;
    auto find = [](const Ct& c, int id) { for (int i = 0; i < c.n; ++i) if (c.ids[i] == id) return i; return -1; };
    std::vector<int> ia(1 + O.n), ib(1 + O.n);
    for (int o = -1; o < O.n; ++o) {
        ia[1 + o] = o < 0 ? 0 : 1 + find(A, O.ids[o]); ib[1 + o] = o < 0 ? 0 : 1 + find(Bc, O.ids[o]);
        if (o >= 0 && ia[1 + o] == 0 && ib[1 + o] == 0) throw Error("mkhe: ctOut has an id that neither operand has");
    }
    const size_t per = std::max<size_t>(1, CTBIN_MAX / (1 + O.n));
    for (size_t b0 = 0; b0 < B; b0 += per) {
        CtBinArgs ba{};
        ba.mods = d_mods; ba.L = L; ba.N = N;
        double bytes = 0;
        int c = 0;
        for (size_t k = b0; k < std::min(B, b0 + per); ++k)
            for (int o = -1; o < O.n; ++o, ++c) {
                const bool ha = o < 0 || ia[1 + o] > 0, hb = o < 0 || ib[1 + o] > 0;
                ba.dst[c] = outs[k]->d + (size_t)(1 + o) * PO;
                ba.a[c] = ha ? a[k]->d + ia[1 + o] * PA : nullptr;
                ba.b[c] = hb ? b[k]->d + ib[1 + o] * PB : nullptr;
                ba.mode[c] = (ha && hb) ? (op == 0 ? 0 : 1) : ha ? 2 : (op == 0 ? 3 : 4);
                bytes += 8.0 * N * L * ((ha && hb) ? 3 : 2);
            }
        ba.ncomp = c;
        { ProfScope ps(this, PROF_OTHER, bytes); launch_ct_binary(ba, s_); }
    }
    MKHE_HIP(hipGetLastError());
}

// mkckks.Evaluator.MulPtxtNew (mkckks/evaluator.go:465-481) on B inputs: NTT of the plaintext once, of every component of every input in one
// launch, one product launch, one inverse launch, and the nb divisions of the Rescale that follows (:480) with one launch each for the batch.
void Context::ct_mul_ptxt_batch(const std::vector<const Ct*>& ins, const u64* dev_pt, int nb, const std::vector<Ct*>& outs) {
    const size_t B = ins.size();
    if (!B) return;
    if (outs.size() != B) throw Error("mkhe: ct_mul_ptxt_batch: one output per input");
    same_shape(ins, "MulPtxt");
    no_cross_alias(ins, outs, "MulPtxt");
    { std::vector<const Ct*> o(outs.begin(), outs.end()); same_shape(o, "MulPtxt outputs"); }
    const int L = ins[0]->limbs, np_ = 1 + ins[0]->n, level = L - 1;
    if (nb < 0 || nb > level) throw Error("cannot Rescale: input Ciphertext already at level 0");
    if (outs[0]->limbs != L - nb || outs[0]->n != ins[0]->n || outs[0]->ids != ins[0]->ids) throw Error("mkhe: ctOut shape does not match ct");
    const size_t PO = (size_t)L * N;
    Arena ar(this, (1 + 2 * B * np_) * PO);
    u64* pt = ar.take(PO); u64* w = ar.take(B * np_ * PO); u64* w2 = ar.take(B * np_ * PO);
    ntt(dev_pt, pt, 1, L, 0, false, false);
    for (size_t base = 0; base < B; base += NTT_MAX_ITEMS) {
        const int cnt = (int)std::min<size_t>(NTT_MAX_ITEMS, B - base);
        NttBatch b{};
        b.mods = d_mods; b.psi = d_psi; b.aux = d_inv_aux; slots_range(b, 0, L);
        b.src_inner = b.dst_inner = N; b.src_outer = b.dst_outer = (long)PO;
        b.nitems = cnt; b.outers_per_item = np_; b.nouter = cnt * np_;
        for (int i = 0; i < cnt; ++i) { b.src_items[i] = ins[base + i]->d; b.dst_items[i] = w + (base + i) * np_ * PO; }
        ntt_fwd_launch(b, false);
    }
    { ProfScope ps(this, PROF_OTHER, 8.0 * N * L * (2.0 * B * np_ + 1)); launch_mul_by_poly(w, w, pt, d_mods, L, N, (int)(B * np_), s_); }
    if (nb == 0) {
        for (size_t base = 0; base < B; base += NTT_MAX_ITEMS) {
            const int cnt = (int)std::min<size_t>(NTT_MAX_ITEMS, B - base);
            NttBatch b{};
            b.mods = d_mods; b.psi = d_psiinv; b.aux = d_inv_aux; slots_range(b, 0, L);
            b.src_inner = b.dst_inner = N; b.src_outer = b.dst_outer = (long)PO;
            b.nitems = cnt; b.outers_per_item = np_; b.nouter = cnt * np_;
            for (int i = 0; i < cnt; ++i) { b.src_items[i] = w + (base + i) * np_ * PO; b.dst_items[i] = outs[base + i]->d; }
            { ProfScope ps(this, PROF_NTT_INV, 16.0 * N * b.nouter * L); ntt_inv_launch(b); }
        }
        return;
    }
    ntt(w, w, (int)(B * np_), L, 0, true, false);
    // DivRoundByLastModulusMany (mkckks/evaluator.go:385-391): nb steps, the last one into the output ciphertexts
    u64* cur = w; u64* nxt = w2;
    for (int k = 0; k < nb; ++k) {
        const int lv = level - k;
        if (k < nb - 1) { launch_div_round_last(nxt, cur, d_mods, d_rescale + (size_t)(lv - 1) * nq, lv, N, (int)(B * np_), (long)PO, (long)PO, s_); std::swap(cur, nxt); continue; }
        for (size_t b0 = 0; b0 < B; b0 += DRL_MAX) {
            DivRoundListArgs da{};
            da.ngroups = (int)std::min<size_t>(DRL_MAX, B - b0);
            for (int g = 0; g < da.ngroups; ++g) da.dst[g] = outs[b0 + g]->d;
            da.src = cur + b0 * np_ * PO; da.mods = d_mods; da.rescale_row = d_rescale + (size_t)(lv - 1) * nq;
            da.src_poly = (long)PO; da.dst_poly = (long)(L - nb) * N; da.level = lv; da.N = N; da.per_group = np_;
            launch_div_round_last_list(da, s_);
        }
    }
    MKHE_HIP(hipGetLastError());
}

}  // namespace mkhe
