// engine_mulrelin.hip -- Context: MulAndRelin[Hoisted] (steps A-F), its split-phase and limb-sharded forms (engine.hip has the tables, pools and the external-product batch)
#include "engine.h"
#include <exception>
#include <algorithm>
#include <cstring>
#include <cstdlib>

namespace mkhe {

// ------------------------------------------------------------------ MulAndRelin[Hoisted]
// keyswitch_hoisted.go:44-179 (hoist == nullptr: keyswitch.go:122-230, same values).
void Context::mul_and_relin(const Ct& op0, const Ct& op1, const Swk* const* hoist0, const Swk* const* hoist1,
                            const Swk* const* rlk_b1, const Swk* const* rlk_d0, const Swk* const* rlk_v0,
                            const Swk& crs_u, Ct& out) {
    mr_prepare(op0, op1, hoist0, hoist1, true, out);
    const int fuse_env = ab_fuse_x();
    const bool fuse = fuse_env && plan_.n0 >= 1 && plan_.n0 <= 16 && !masked_;      // (five to sixteen parties: the wide forms of the kernel)
    // y inside the F1 kernel as well (round 4): the thread that forms <h(c0_i), y> at a coefficient needs y there and nowhere else, so that y is
    // neither a launch nor 2 x 59 MB of traffic -- when op1 has as many parties as op0 (at most four: the group form of the kernel)
    const int fuse_y_env = ab_fuse_y();
    const bool fuse_y = fuse && fuse_y_env && plan_.n1 >= 1 && (plan_.n0 <= 4 ? plan_.n1 <= 4 : (plan_.n0 <= 8 && plan_.n1 == plan_.n0));      // (one to four parties per operand: ext_inner_xy_kernel<G0, G1>; five to eight in both: ext_inner_xy_wide_kernel)
    mr_xy(rlk_b1, rlk_d0, x_, y_, true, true, fuse, fuse_y);
    mr_finish(op0, op1, x_, y_, rlk_v0, crs_u, out);
}

// mkckks.Evaluator.mulRelinHoisted (evaluator.go:558-581) = MulAndRelinHoisted + one Rescale, as ONE engine call: the DivRoundByLastModulus
// is applied by the merged ModDown of the last batch as it stores (ModDownMergedArgs::rescale_row), so the level-L product is never written
// and the rescale is neither a launch nor a pass of its own.  That needs every output slot to be written exactly once by that launch (up to
// four products per destination: at most four parties per operand, single device, tensor term folded in); otherwise the product goes to a
// pooled temporary and Context::rescale follows -- the same integers either way.
void Context::mul_relin_rescale(const Ct& op0, const Ct& op1, const Swk* const* hoist0, const Swk* const* hoist1,
                                const Swk* const* rlk_b1, const Swk* const* rlk_d0, const Swk* const* rlk_v0,
                                const Swk& crs_u, Ct& out) {
    const int L = out.limbs + 1;                           // limbs of the product
    if (out.limbs < 1 || L > nq) throw Error("cannot Rescale: input Ciphertext already at level 0");
    Ct full; full.n = out.n; full.limbs = L; full.ids = out.ids;
    const size_t words = (size_t)(1 + out.n) * L * N;
    full.d = pool_alloc(words);
    static const int fuse_env = MKHE_AB_INT("MKHE_FUSE_RESCALE", 1);
    rs_maps_.clear();
    if (fuse_env && !masked_) rs_maps_.push_back(RsMap{full.d, out.d, 1 + out.n, out.limbs, false});
    try {
        mul_and_relin(op0, op1, hoist0, hoist1, rlk_b1, rlk_d0, rlk_v0, crs_u, full);
        const bool done = !rs_maps_.empty() && rs_maps_[0].done;
        rs_maps_.clear();
        if (!done) rescale(full, 1, out);
    } catch (...) { rs_maps_.clear(); const HandleUsers none; pool_free(full.d, words, &none); throw; }
    // the temporary never left this context (no handle, no other context can have work queued on it): an EMPTY user list, so that the pool does
    // not order its next user behind every live context (users == nullptr means "unknown": the forks of the cnn evaluation would serialise)
    const HandleUsers none;
    pool_free(full.d, words, &none);
}

// -- step 0: validate, map ids, hoist the operands when the caller did not (MulRelinNew, evaluator.go:416-443)
// and start step D (tensor).  with_c0 = false leaves c0_0*c1_0 out of out_0 (another rank of a party-sharded
// evaluation adds it).
void Context::mr_prepare(const Ct& op0, const Ct& op1, const Swk* const* hoist0, const Swk* const* hoist1, bool with_c0, Ct& out) {
    MrPlan& p = plan_;
    p = MrPlan{};
    p.level = out.limbs - 1; p.L = p.level + 1;
    check_level(p.level);
    if (op0.limbs < p.L || op1.limbs < p.L) throw Error("Cannot MulAndRelin: op0 and op1 have different levels");
    p.n0 = op0.n; p.n1 = op1.n; p.nout = out.n;
    if (p.n0 > 32 || p.n1 > 32 || out.n > 32) throw Error("mkhe: too many parties");
    // out ids must be the union of the operand id sets (newCiphertextBinary, mkckks/evaluator.go:306-313)
    p.slot0.assign(p.n0, 0); p.slot1.assign(p.n1, 0);
    auto find = [&](int id) { for (int o = 0; o < out.n; ++o) if (out.ids[o] == id) return o; return -1; };
    std::vector<char> seen(out.n, 0);
    for (int a = 0; a < p.n0; ++a) { int o = find(op0.ids[a]); if (o < 0) throw Error("mkhe: ctOut lacks an id of op0"); p.slot0[a] = o; seen[o] = 1; }
    for (int a = 0; a < p.n1; ++a) { int o = find(op1.ids[a]); if (o < 0) throw Error("mkhe: ctOut lacks an id of op1"); p.slot1[a] = o; seen[o] = 1; }
    for (int o = 0; o < out.n; ++o) if (!seen[o]) throw Error("mkhe: ctOut has an id that neither operand has");
    const size_t P0 = (size_t)op0.limbs * N, P1 = (size_t)op1.limbs * N;
    p.h0.assign(p.n0, nullptr); p.h1.assign(p.n1, nullptr);
    const bool same = (&op0 == &op1) && hoist0 == hoist1;
    p.own0 = (hoist0 == nullptr) && alpha == 1; p.own1 = (hoist1 == nullptr) && alpha == 1;
    if (masked_ && !(p.own0 && p.own1)) throw Error("mkhe: a limb-sharded evaluation hoists its operands itself");
    std::vector<const u64*> dsrc; std::vector<u64*> ddst;
    for (int a = 0; a < p.n0; ++a) {
        if (hoist0) { if (!hoist0[a]) throw Error("mkhe: missing hoisted form"); p.h0[a] = hoist0[a]->d; }
        else { Swk& s = hoist_slot(0, a); dsrc.push_back(op0.d + (1 + a) * P0); ddst.push_back(s.d); p.h0[a] = s.d; }
    }
    for (int a = 0; a < p.n1; ++a) {
        if (hoist1) { if (!hoist1[a]) throw Error("mkhe: missing hoisted form"); p.h1[a] = hoist1[a]->d; }
        else if (same) p.h1[a] = p.h0[a];
        else { Swk& s = hoist_slot(1, a); dsrc.push_back(op1.d + (1 + a) * P1); ddst.push_back(s.d); p.h1[a] = s.d; }
    }
    if (!dsrc.empty()) decompose_batch(p.level, dsrc, ddst, true);
    // D: tensor product in the NTT domain, back to coefficients -- started here on the side stream: it only
    // needs the operands and the engine's own hoisted digits, runs beside the x / y accumulation and meets
    // the main chain again at the first ModDown of mr_finish.
    {
        const int level = p.level, L = p.L, n0 = p.n0, n1 = p.n1;
        const size_t PO = (size_t)L * N;
        u64* nb_ = scratch(nttbuf_, nttbuf_words_, (size_t)(2 + n0 + n1) * PO);
        (void)level;
        fork_side(1);
        s_ = overlap ? stream2 : stream;
        {
            // NTT(c0_0), NTT(c1_0) always; party components only when the caller supplied the hoisted forms
            // (the engine's own hoisted digits already contain NTT(c_i) on their diagonal, alpha = 1)
            NttBatch b{};
            b.mods = d_mods; b.psi = d_psi; b.aux = d_inv_aux; slots_q_owned(b, L);
            b.src_inner = b.dst_inner = N; b.dst_outer = (long)PO;
            if (p.own0 && p.own1) {
                b.nitems = 2; b.outers_per_item = 1; b.nouter = 2;
                b.src_items[0] = op0.d; b.src_items[1] = op1.d;
                b.dst_items[0] = nb_; b.dst_items[1] = nb_ + (size_t)(1 + n0) * PO;
                ntt_fwd_launch(b, false);
            } else {
                b.src = op0.d; b.src_outer = (long)P0; b.dst = nb_; b.nouter = p.own0 ? 1 : 1 + n0;
                ntt_fwd_launch(b, false);
                b.src = op1.d; b.src_outer = (long)P1; b.dst = nb_ + (size_t)(1 + n0) * PO; b.nouter = p.own1 ? 1 : 1 + n1;
                ntt_fwd_launch(b, false);
            }
        }
        // With at least one party in op0 every output slot receives an external product in steps E / F2.  The tensor term then
        // stays in the NTT domain, times P, and joins the summed Q parts of that (merged) batch: ModDown's (x - lift) * P^-1
        // returns it as itself, canonical like everything else -- no inverse NTT for step D.
        const bool fold = n0 >= 1 && !masked_ && 2 * n0 + n1 <= EXT_MAX_ITEMS && ext_merge_members(level) >= 2;
        u64* tout = out.d;
        if (fold) { tout = scratch(tens_, tens_words_, (size_t)(1 + out.n) * PO); p.tens = tout; }
        TensorArgs ta{};
        ta.a0 = nb_; ta.b0 = nb_ + (size_t)(1 + n0) * PO; ta.out = tout; ta.mods = d_mods;
        if (fold) ta.scale = d_pmodq;
        ta.nout = out.n; ta.L = L; ta.N = N; ta.with_c0 = with_c0 ? 1 : 0;
        if (masked_) { ta.limbs = d_ownq; ta.nlimbs = nq_owned(level); }
        const long diag = (long)(mtot + 1) * N;
        for (int a = 0; a < n0; ++a) {
            const int o = 1 + p.slot0[a];
            if (p.own0) { ta.a[o] = p.h0[a]; ta.a_ls[o] = diag; } else { ta.a[o] = nb_ + (size_t)(1 + a) * PO; ta.a_ls[o] = N; }
        }
        for (int a = 0; a < n1; ++a) {
            const int o = 1 + p.slot1[a];
            if (p.own1) { ta.b[o] = p.h1[a]; ta.b_ls[o] = diag; } else { ta.b[o] = nb_ + (size_t)(2 + n0 + a) * PO; ta.b_ls[o] = N; }
        }
        { ProfScope ps(this, PROF_TENSOR, 8.0 * N * L * (2.0 + n0 + n1 + 1 + out.n)); launch_tensor(ta, s_); }
        if (fold) { /* no inverse NTT: see above */ }
        else if (!masked_) ntt(out.d, out.d, 1 + out.n, L, 0, true, false);
        else {
            NttBatch ib{};
            ib.src = out.d; ib.dst = out.d; ib.mods = d_mods; ib.psi = d_psiinv; ib.aux = d_inv_aux; slots_q_owned(ib, L);
            ib.src_outer = ib.dst_outer = (long)PO; ib.src_inner = ib.dst_inner = N; ib.nouter = 1 + out.n;
            if (ib.nslots > 0) { ProfScope ps(this, PROF_NTT_INV, 16.0 * N * ib.nouter * ib.nslots); ntt_inv_launch(ib); }
        }
        side_done(1);
        s_ = stream;
    }
    p.valid = true; p.head_done = false;
}

// -- steps B, C: x = [MForm] sum_i d_i (.) h(c0_i),  y = [MForm] sum_j b_j (.) h(c1_j)   (keyswitch_hoisted.go:79-117)
// mform = false leaves the canonical partial sums for a cross-device reduction (party sharding).
void Context::mr_xy(const Swk* const* rlk_b1, const Swk* const* rlk_d0, u64* x, u64* y, bool mform, bool defer_x, bool fuse_x, bool fuse_y) {
    MrPlan& p = plan_;
    if (!p.valid) throw Error("mkhe: mr_xy without mr_prepare");
    const int nb = beta(p.level), nslots = nslots_qp(p.level);
    p.xkeys.clear(); p.xfused = nullptr;
    if (fuse_x) {
        if (!mform) throw Error("mkhe: internal: the fused x is produced in Montgomery form");
        for (int a = 0; a < p.n0; ++a) {
            if (!rlk_d0[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
            p.xkeys.push_back(rlk_d0[a]->d);
        }
        p.xfused = x;
    }
    p.ykeys.clear();
    if (fuse_y) {
        if (!fuse_x || p.n1 < 1 || p.n0 > 8 || (p.n0 > 4 ? p.n1 != p.n0 : p.n1 > 4)) throw Error("mkhe: internal: y inside the F1 kernel needs the x by-product and one to four parties per operand (or five to eight in both)");
        for (int a = 0; a < p.n1; ++a) {
            if (!rlk_b1[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
            p.ykeys.push_back(rlk_b1[a]->d);
        }
    }
    // y first: it feeds step F, the long chain (F1 -> Decompose -> F2); x only feeds step E
    for (int side = fuse_y ? 0 : 1; side >= (fuse_x ? 1 : 0); --side) {
        const int n = side ? p.n1 : p.n0;
        if (n > MAX_TERMS) throw Error("mkhe: too many parties");
        InnerProductArgs ip{};
        for (int a = 0; a < n; ++a) {
            const Swk* key = side ? rlk_b1[a] : rlk_d0[a];
            if (!key) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
            ip.a[a] = key->d; ip.b[a] = side ? p.h1[a] : p.h0[a];
        }
        ip.out = side ? y : x; ip.mods = d_mods; ip.map = map_qp(p.level);
        ip.term_outer = ip.out_outer = (long)mtot * N; ip.nterms = n; ip.nslots = nslots; ip.nouter = nb; ip.N = N; ip.mform_out = mform ? 1 : 0;
        const bool on_side = side == 0 && defer_x && overlap;
        if (on_side) { fork_side(2); s_ = stream2; }
        { ProfScope ps(this, PROF_INNER, 8.0 * N * nslots * nb * (2.0 * n + 1)); launch_inner_product(ip, s_); }
        if (on_side) { side_done(2); s_ = stream; p.x_pending = true; }
    }
    MKHE_HIP(hipGetLastError());
}

// -- steps D, E, F (keyswitch_hoisted.go:119-178) with x, y in Montgomery form.
// mr_finish = head (F1 and the Decompose of its results: needs y only) + tail (E and F2: needs x).  A party-sharded caller
// runs the head while the all-reduce of x is still in flight (mkhe-kklss_amd/dist.py).
void Context::mr_finish_head(const Ct& op0, const Ct& op1, const u64* y, Ct& out) {
    MrPlan& p = plan_;
    if (!p.valid) throw Error("mkhe: mr_finish without mr_prepare");
    if (out.limbs != p.L || out.n != p.nout || op0.n != p.n0 || op1.n != p.n1) throw Error("mkhe: mr_finish arguments do not match mr_prepare");
    const int level = p.level, L = p.L, n0 = p.n0;
    const size_t PO = (size_t)L * N;
    u64* tbuf = scratch(tbuf_, tbuf_words_, (size_t)n0 * PO);
    std::vector<ExtItem> items;
    // F1: t_i = <h(c0_i), y>_P -- the head of the long chain; E (needs x, which may still be accumulating on the side
    // stream) joins the last batch below
    for (int a = 0; a < n0; ++a) {
        ExtItem it{p.h0[a], y, tbuf + (size_t)a * PO, false};
        if (!p.xkeys.empty()) it.xkey = p.xkeys[a];
        items.push_back(it);
    }
    if (!p.xkeys.empty()) ext_xout_ = p.xfused;          // x = sum_i d_i (.) h(c0_i) comes out of the same pass over h(c0_i)
    // N = 2^15 (round 6): the digits of the t_i never reach HBM -- no Decompose launch below, the tail batch's product kernel transforms them itself
    // (ntt16_f2_kernel) -- when that batch is a merged one (the parts of its products meet at the load of the inverse NTT); step E is either done by
    // then (inside the F1 kernel) or computed by the inner-product kernel in front of it (the sharded finish: x arrives from the other ranks)
    const int fuse_e_env = ab_fuse_e();
    const bool will_e = !p.ykeys.empty() && fuse_e_env && 2 * n0 + p.n1 <= EXT_MAX_ITEMS;
    p.f2_fused = n0 > 0 && p.tens != nullptr && f2_fused_ok(level, n0, p.n1);
    const int f2_extra = p.f2_fused ? 2 * n0 * (f2_schedule(n0, level).parts - 1) : 0;
    if (p.f2_fused && p.ykeys.empty()) scratch(c1b_, c1b_words_, (size_t)(2 * n0 + p.n1 + f2_extra) * mtot * N);
    if (!p.ykeys.empty()) {
        ext_ykeys_ = p.ykeys; ext_yh_ = p.h1;            // ... and y is computed in it
        // ... and step E: the thread holds x[d] and h(c1_j)[d], so <h(c1_j), x> costs it G more accumulators, and x is never stored nor the h(c1_j) read
        // again by the tail batch -- whose c1 slots 2 n0 .. 2 n0 + n1 - 1 (the E items) are filled here: the scratch is sized for the tail now, so
        // that it is the same allocation then (nothing else of a MulAndRelin touches it in between)
        if (will_e) {
            scratch(c1b_, c1b_words_, (size_t)(2 * n0 + p.n1 + f2_extra) * mtot * N);
            ext_e_slot_ = 2 * n0;
        }
    }
    try { ext_batch(level, items); } catch (...) { ext_xout_ = nullptr; ext_ykeys_.clear(); ext_yh_.clear(); ext_e_slot_ = -1; throw; }
    p.e_done = ext_e_slot_ >= 0;
    ext_xout_ = nullptr; ext_ykeys_.clear(); ext_yh_.clear(); ext_e_slot_ = -1;
    // F2: h(t_i) ; out_0 += <h(t_i), v_i>_P ; out_i += <h(t_i), u>_P
    p.f2_tbuf = tbuf;
    if (!p.f2_fused) {
        std::vector<const u64*> dsrc; std::vector<u64*> ddst;
        for (int a = 0; a < n0; ++a) { dsrc.push_back(tbuf + (size_t)a * PO); ddst.push_back(hoist_slot(2, a).d); }
        // (the digits of the t_i are read once, by the two F2 products of their party: a small launch of the small ring leaves them after the cross
        // stages and the tail's product kernel finishes the transform -- when step E is not an item that reads other digits in the same launch)
        p.f2_staged = n0 > 0 && (p.e_done || p.n1 == 0) && ext_fused_ok(level, n0);
        if (n0) decompose_batch(level, dsrc, ddst, true, p.f2_staged);
    }
    p.head_done = true;
    MKHE_HIP(hipGetLastError());
}
void Context::mr_finish_tail(const Ct& op0, const Ct& op1, const u64* x, const Swk* const* rlk_v0, const Swk& crs_u, Ct& out) {
    MrPlan& p = plan_;
    if (!p.valid || !p.head_done) throw Error("mkhe: mr_finish_tail without mr_finish_head");
    if (out.limbs != p.L || out.n != p.nout || op0.n != p.n0 || op1.n != p.n1) throw Error("mkhe: mr_finish arguments do not match mr_prepare");
    const int level = p.level, L = p.L, n0 = p.n0, n1 = p.n1;
    const size_t PO = (size_t)L * N;
    // E: out_j += <h(c1_j), x>_P ; F2: out_0 += <h(t_i), v_i>_P, out_i += <h(t_i), u>_P   (one batch; items that share a
    // destination are accumulated one after the other by the same thread of the ModDown kernel)
    std::vector<ExtItem> items;
    // (the F2 pairs first: grouped four at a time they are the longest blocks of the launch, and the sums are order independent)
    for (int a = 0; a < n0; ++a) {
        if (!rlk_v0[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
        const u64* ht = p.f2_fused ? p.f2_tbuf + (size_t)a * PO : hoist_slot(2, a).d;      // (fused: the digits exist in registers only; `ah` just tells the items apart)
        items.push_back(ExtItem{ht, rlk_v0[a]->d, out.d, true});
        if (p.f2_fused) { items.back().f2_party = a; items.back().f2_key = 0; }
        items.push_back(ExtItem{ht, crs_u.d, out.d + (size_t)(1 + p.slot0[a]) * PO, true});
        if (p.f2_fused) { items.back().f2_party = a; items.back().f2_key = 1; }
    }
    for (int a = 0; a < n1; ++a) { items.push_back(ExtItem{p.h1[a], x, out.d + (size_t)(1 + p.slot1[a]) * PO, true}); items.back().pre = p.e_done; }
    if (p.f2_fused) { ext_f2_src_.clear(); for (int a = 0; a < n0; ++a) ext_f2_src_.push_back(p.f2_tbuf + (size_t)a * PO); }
    if (p.x_pending) { join_side(2); p.x_pending = false; }
    if (p.f2_staged) { ext_staged_.clear(); for (int a = 0; a < n0; ++a) ext_staged_.push_back(hoist_slot(2, a).d); }
    struct Unstage { Context* c; ~Unstage() { c->ext_staged_.clear(); c->ext_f2_src_.clear(); if (std::uncaught_exceptions()) c->staged_open_.clear(); } } unstage{this};
    if (p.tens) {
        // the tensor term of every output slot rides on the first product that goes there (see mr_prepare)
        std::vector<const u64*> seen;
        for (auto& it : items) {
            if (std::find(seen.begin(), seen.end(), it.dst) != seen.end()) continue;
            seen.push_back(it.dst);
            it.accumulate = false; it.qadd = p.tens + (it.dst - out.d);
        }
        if ((int)seen.size() != 1 + out.n) throw Error("mkhe: internal: an output slot without an external product");
        join_side(1);                  // the tensor chain, before the inverse NTT that sums it in
        ext_batch(level, items);
    } else
    ext_batch(level, items, 1);        // joins the tensor chain before the ModDown accumulates into out
    p.valid = false; p.head_done = false;
    MKHE_HIP(hipGetLastError());
}
void Context::mr_finish(const Ct& op0, const Ct& op1, const u64* x, const u64* y, const Swk* const* rlk_v0,
                        const Swk& crs_u, Ct& out) {
    mr_finish_head(op0, op1, y, out);
    mr_finish_tail(op0, op1, x, rlk_v0, crs_u, out);
}

// ------------------------------------------------------------------ the schedule of ntt16_f2_kernel (ntt_kernels.h F2FusedArgs)
// One workgroup per CU, one round: the kernel lasts as long as its longest workgroup, so the passes -- (party, limb slot, half limb, digit), parties
// outermost, digits innermost -- are dealt by WEIGHT: a pass under a 59/60-bit modulus carries its partial reductions (MKHE_F2_BALANCE percent per
// reduction point of the modulus's schedule; 0, the default: every pass the same, the cuts then fall on whole and half groups).  A workgroup's
// passes of one (party, slot, half) are a run = one part of that group's two products; parts = the most any group is cut into (the inverse NTT adds
// them at its load: VI_SUMS), a group cut into fewer has its last run zero the others.  parts = 0: no schedule within the kernel's limits.
bool Context::f2_fused_ok(int level, int n0, int n1) const {
    (void)n1;
    if (logN != 15 || alpha != 1 || masked_ || !d_psi31 || !d_psi31n || mall > NTT_MAX_SLOTS || h16_gap_) return false;
    if (!ntt16_f2_ok(logN, n0, beta(level), nslots_qp(level))) return false;
    if (ext_merge_members(level) < 2) return false;
    return const_cast<Context*>(this)->f2_schedule(n0, level).parts >= 1;
}
// the cut itself, a pure function of the shape (also behind mkhe_f2_schedule_probe: tests/test_f2_schedule.py checks on the CPU that every pass of
// every shape is dealt exactly once).  weights: per limb slot; segs: G * F2_SEGS entries.  Returns the number of workgroups, parts in *parts_out;
// 0 = the shape has no schedule within the kernel's limits.
int f2_build_schedule(int np0, int nb, int nslots, const long* weights, int G, F2Seg* segs, int* parts_out, int max_parts, long* cost_out, long* worst_out) {
    *parts_out = 0;
    if (max_parts <= 0) max_parts = f2_max_parts(np0);
    if (np0 < 1 || np0 > F2_MAX_P || nb < 1 || nb > 255 || nslots < 1 || nslots > NTT_MAX_SLOTS || G < 1) return 0;
    const int ngroups = np0 * nslots * 2;
    long W = 0;
    for (int g = 0; g < ngroups; ++g) { const long wi = weights[(g / 2) % nslots]; if (wi < 1) return 0; W += wi * nb; }
    // pass -> workgroup by the midpoint of its weight interval; runs = maximal stretches of one group inside one workgroup
    struct Run { int wg, g, d0, nd; };
    std::vector<Run> runs;
    long cum = 0;
    for (int g = 0; g < ngroups; ++g)
        for (int d = 0; d < nb; ++d) {
            const long wi = weights[(g / 2) % nslots];
            int wg = (int)(((2 * cum + wi) * (long)G) / (2 * W));
            if (wg >= G) wg = G - 1;
            cum += wi;
            if (!runs.empty() && runs.back().wg == wg && runs.back().g == g) ++runs.back().nd;
            else runs.push_back(Run{wg, g, d, 1});
        }
    std::vector<int> per_wg(G, 0), per_g(ngroups, 0);
    for (const Run& r : runs) { if (++per_wg[r.wg] > F2_SEGS) return 0; ++per_g[r.g]; }
    int parts = 0;
    for (int g = 0; g < ngroups; ++g) parts = std::max(parts, per_g[g]);
    // the members of a merged destination (out_0: one per party, four at most) x their parts + the tensor term at the load of one inverse job
    if (parts < 1 || parts > max_parts) return 0;
    if (cost_out || worst_out) {
        // what the launch lasts = its longest workgroup: a pass = its weight, a run = a quarter of a pass on top (twiddles of its modulus into LDS,
        // accumulators out); the parts are read again by the inverse NTT
        std::vector<long> wcost(G, 0), wpass(G, 0);
        for (const Run& r : runs) { const long w = (long)r.nd * weights[(r.g / 2) % nslots]; wpass[r.wg] += w; wcost[r.wg] += w + F2_RUN_COST; }
        long worst = 0, worst_p = 0;
        for (int w = 0; w < G; ++w) { worst = std::max(worst, wcost[w]); worst_p = std::max(worst_p, wpass[w]); }
        if (cost_out) *cost_out = worst + (long)F2_PART_COST * parts;
        if (worst_out) *worst_out = worst_p;
    }
    for (size_t i = 0; i < (size_t)G * F2_SEGS; ++i) segs[i] = F2Seg{};
    std::vector<int> nseg(G, 0), seen(ngroups, 0);
    int nwg = 0;
    for (const Run& r : runs) {
        F2Seg& sg = segs[(size_t)r.wg * F2_SEGS + nseg[r.wg]++];
        sg.party = (unsigned char)(r.g / (2 * nslots)); sg.slot = (unsigned char)((r.g / 2) % nslots); sg.half = (unsigned char)(r.g & 1);
        sg.d0 = (unsigned char)r.d0; sg.nd = (unsigned char)r.nd; sg.part = (unsigned char)seen[r.g]++;
        sg.pad0 = (unsigned char)(seen[r.g] == per_g[r.g] ? parts - per_g[r.g] : 0);      // the group's last run zeroes the parts the group does not have
        nwg = std::max(nwg, r.wg + 1);
    }
    *parts_out = parts;
    return nwg;
}
// the grid: the cheapest feasible cut on at most Gmax workgroups (one per CU; ties: the smaller grid).  Shapes with fewer passes than two per CU --
// one or two parties, low levels -- are cut into more parts on fewer workgroups than CUs rather than not at all.
int f2_plan_schedule(int np0, int nb, int nslots, const long* weights, int Gmax, F2Seg* segs, int* parts_out, int max_parts) {
    *parts_out = 0;
    if (Gmax < 1 || nb < 2) return 0;          // (level 0: one digit, three limb slots -- the launches it would replace are a few microseconds each)
    std::vector<F2Seg> trial((size_t)Gmax * F2_SEGS);
    long best = -1; int best_g = 0;
    for (int G = 1; G <= Gmax; ++G) {
        int parts = 0; long cost = 0;
        if (f2_build_schedule(np0, nb, nslots, weights, G, trial.data(), &parts, max_parts, &cost) < 1) continue;
        if (best < 0 || cost < best) { best = cost; best_g = G; }
    }
    if (!best_g) return 0;
    // Worth it?  The kernel lasts as long as its longest workgroup -- whole passes, a run's set-up, the parts read again -- where the launches it replaces
    // (Decompose NTT two workgroups per CU, streaming products) cost by the total.  Measured over parties x levels (profiles/r6_f2_plan_sweep.txt): the
    // fused launch wins while that cost stays within ~1.4 passes of an even deal over the chip, and loses beyond (three parties at level 13: six passes in
    // two runs where 5.25 would be even, - 3 %).
    {
        long W = 0;
        for (int g = 0; g < np0 * nslots * 2; ++g) W += weights[(g / 2) % nslots] * nb;
        if (best - W / Gmax > F2_SLACK) return 0;
    }
    for (size_t i = 0; i < (size_t)Gmax * F2_SEGS; ++i) segs[i] = F2Seg{};
    return f2_build_schedule(np0, nb, nslots, weights, best_g, segs, parts_out, max_parts, nullptr);
}
const Context::F2Sched& Context::f2_schedule(int np0, int level) {
    const long key = ((long)np0 << 8) | level;
    auto found = f2_sched_.find(key);
    if (found != f2_sched_.end()) return found->second;
    F2Sched sc;
    const int nb = beta(level), G = ntt16_f2_grid();
    NttBatch q{};
    slots_qp(q, level);
    const int nslots = q.nslots;
    static const int balance = MKHE_AB_INT("MKHE_F2_BALANCE", 0);      // percent of a pass per reduction point of a 59/60-bit modulus (0: every pass the same)
    const int wred = balance;
    std::vector<long> w(nslots, 100);
    for (int s2 = 0; s2 < nslots && balance; ++s2) {
        const int m = q.mod[s2];
        if (!small16_[m]) w[s2] = 100 + (long)wred * __builtin_popcount((h16_sched_.empty() ? 15u : (unsigned)h16_sched_[m]) & 15u);
    }
    std::vector<F2Seg> segs((size_t)G * F2_SEGS);
    int parts = 0;
    const bool plan = ntt16_f2_mode() != 2;            // (MKHE_F2_FUSED=2: the whole chip or nothing, as the first version of the kernel was scheduled)
    const int nwg = plan ? f2_plan_schedule(np0, nb, nslots, w.data(), G, segs.data(), &parts, 0)
                         : f2_build_schedule(np0, nb, nslots, w.data(), G, segs.data(), &parts, std::min(3, f2_max_parts(np0)));
    if (nwg > 0) {
        MKHE_HIP(hipSetDevice(device));
        MKHE_HIP(hipMalloc(&sc.d_segs, segs.size() * sizeof(F2Seg)));
        const hipError_t e = hipMemcpy(sc.d_segs, segs.data(), segs.size() * sizeof(F2Seg), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(sc.d_segs); sc.d_segs = nullptr; MKHE_HIP(e); }
        sc.nwg = nwg; sc.parts = parts;
    } else { sc.parts = 0; sc.nwg = 0; }
    return f2_sched_.emplace(key, sc).first->second;
}

// ------------------------------------------------------------------ limb-sharded MulAndRelin (see engine.h)
void Context::zero_unowned(u64* base, int npolys, long poly_stride, int first_mod, int nlimbs) {
    for (int l = 0; l < nlimbs; ++l)
        if (!own_[first_mod + l])
            MKHE_HIP(hipMemset2DAsync(base + (size_t)l * N, (size_t)poly_stride * sizeof(u64), 0, (size_t)N * sizeof(u64), npolys, s_));
}

size_t Context::lsh_phase(int phase, const Ct& op0, const Ct& op1, const Swk* const* rlk_b1, const Swk* const* rlk_d0,
                          const Swk* const* rlk_v0, const Swk* crs_u, Ct& out, u64* stage) {
    if (!masked_) throw Error("mkhe: lsh_phase needs mkhe_ctx_set_owned first");
    if (!stage) throw Error("mkhe: lsh_phase needs a staging buffer");
    MrPlan& p = plan_;
    const size_t item_words = (size_t)mtot * N, prow = (size_t)np * N * sizeof(u64);
    // P limbs of the c1 pool <-> contiguous staging [item][np][N]; limbs this rank does not own travel as zeros
    auto pack = [&](int n) -> size_t {
        zero_unowned(c1b_ + (size_t)nq * N, n, (long)item_words, nq, np);
        if (n) MKHE_HIP(hipMemcpy2DAsync(stage, prow, c1b_ + (size_t)nq * N, item_words * sizeof(u64), prow, n, hipMemcpyDeviceToDevice, s_));
        return (size_t)n * np * N;
    };
    auto unpack = [&](int n) {
        if (n) MKHE_HIP(hipMemcpy2DAsync(c1b_ + (size_t)nq * N, item_words * sizeof(u64), stage, prow, prow, n, hipMemcpyDeviceToDevice, s_));
    };
    if (phase == 1) {
        // every rank computes its limbs of the tensor product (c0_0*c1_0 included: the limbs are disjoint), hoists all
        // parties under its moduli and accumulates x, y there: complete sums, no exchange
        mr_prepare(op0, op1, nullptr, nullptr, true, out);
        mr_xy(rlk_b1, rlk_d0, x_, y_, true, false);
        const size_t PO = (size_t)p.L * N;
        u64* tbuf = scratch(tbuf_, tbuf_words_, (size_t)std::max(p.n0, 1) * PO);
        lsh_items_.clear();
        for (int a = 0; a < p.n0; ++a) lsh_items_.push_back(ExtItem{p.h0[a], y_, tbuf + (size_t)a * PO, false});
        scratch(c1b_, c1b_words_, (size_t)std::max<size_t>(lsh_items_.size(), 1) * item_words);
        ext_batch(p.level, lsh_items_, -1, 1);
        return pack((int)lsh_items_.size());
    }
    if (!p.valid) throw Error("mkhe: lsh_phase out of order");
    const int level = p.level, L = p.L;
    const size_t PO = (size_t)L * N;
    if (phase == 2) {
        unpack((int)lsh_items_.size());
        ext_batch(level, lsh_items_, -1, 2);                       // t_i, owned limbs
        zero_unowned(tbuf_, p.n0, (long)PO, 0, L);
        if (p.n0) MKHE_HIP(hipMemcpyAsync(stage, tbuf_, (size_t)p.n0 * PO * sizeof(u64), hipMemcpyDeviceToDevice, s_));
        return (size_t)p.n0 * PO;
    }
    if (phase == 3) {
        if (!crs_u || !rlk_v0) throw Error("mkhe: lsh_phase 3 needs the v keys and the CRS");
        if (p.n0) MKHE_HIP(hipMemcpyAsync(tbuf_, stage, (size_t)p.n0 * PO * sizeof(u64), hipMemcpyDeviceToDevice, s_));
        std::vector<const u64*> dsrc; std::vector<u64*> ddst;
        for (int a = 0; a < p.n0; ++a) { dsrc.push_back(tbuf_ + (size_t)a * PO); ddst.push_back(hoist_slot(2, a).d); }
        if (p.n0) decompose_batch(level, dsrc, ddst, true);
        lsh_items_.clear();
        for (int a = 0; a < p.n1; ++a) lsh_items_.push_back(ExtItem{p.h1[a], x_, out.d + (size_t)(1 + p.slot1[a]) * PO, true});
        for (int a = 0; a < p.n0; ++a) {
            if (!rlk_v0[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
            lsh_items_.push_back(ExtItem{hoist_slot(2, a).d, rlk_v0[a]->d, out.d, true});
            lsh_items_.push_back(ExtItem{hoist_slot(2, a).d, crs_u->d, out.d + (size_t)(1 + p.slot0[a]) * PO, true});
        }
        scratch(c1b_, c1b_words_, (size_t)std::max<size_t>(lsh_items_.size(), 1) * item_words);
        ext_batch(level, lsh_items_, -1, 1);
        return pack((int)lsh_items_.size());
    }
    if (phase == 4) {
        unpack((int)lsh_items_.size());
        ext_batch(level, lsh_items_, 1, 2);                        // joins the tensor chain, then accumulates into out
        zero_unowned(out.d, 1 + out.n, (long)PO, 0, L);
        p.valid = false;
        MKHE_HIP(hipGetLastError());
        return (size_t)(1 + out.n) * PO;
    }
    throw Error("mkhe: lsh_phase 1..4");
}

// reduction epilogue of the party-sharded path: words hold sums of canonical residues of several ranks
// (each < q, total < 2^63); bring them back to [0,q) and optionally to Montgomery form (MFormLvl).
void Context::fold(u64* buf, bool qp_shaped, int level, int npolys, long poly_stride, bool mform) {
    check_level(level);
    FoldArgs fa{};
    fa.buf = buf; fa.mods = d_mods; fa.map = qp_shaped ? map_qp(level) : d_map_id;
    fa.nslots = qp_shaped ? nslots_qp(level) : level + 1; fa.npolys = npolys; fa.poly_stride = poly_stride; fa.N = N; fa.mform = mform ? 1 : 0;
    { ProfScope ps(this, PROF_OTHER, 16.0 * N * fa.nslots * npolys); launch_fold(fa, s_); }
    MKHE_HIP(hipGetLastError());
}

void Context::fold_pieces(const u64* pieces, int npieces, long piece_stride, long first_limb, long nlimbs, int level, bool mform, u64* dst) {
    check_level(level);
    if (!pieces || !dst || npieces < 1 || npieces > 64 || first_limb < 0 || nlimbs < 0 || first_limb + nlimbs > (long)beta_max * mtot || mtot > 64)
        throw Error("mkhe: fold_pieces outside a switching key");
    FoldPiecesArgs fa{};
    fa.pieces = pieces; fa.dst = dst; fa.mods = d_mods; fa.piece_stride = piece_stride; fa.first_limb = first_limb;
    for (int j = 0; j <= level; ++j) fa.active |= 1ull << j;
    for (int j = 0; j < np; ++j) fa.active |= 1ull << (nq + j);
    fa.npieces = npieces; fa.nlimbs = (int)nlimbs; fa.mtot = mtot; fa.ndigits = beta(level); fa.N = N; fa.mform = mform ? 1 : 0;
    { ProfScope ps(this, PROF_OTHER, 8.0 * N * nlimbs * (npieces + 1.0)); launch_fold_pieces(fa, s_); }
    MKHE_HIP(hipGetLastError());
}

}  // namespace mkhe
