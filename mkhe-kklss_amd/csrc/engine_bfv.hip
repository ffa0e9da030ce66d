// engine_bfv.hip -- Context: mkbfv -- the ring R = Q || QMul, basis conversions, two-gadget external products, MulAndRelinBFV[Hoisted]
#include "engine.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>

namespace mkhe {

// ------------------------------------------------------------------ mkbfv
// ring R = Q || QMul (mkbfv/params.go:36-38): limb j of a PolyR uses modulus j (j < nq) or nq+np+(j-nq)
void Context::ntt_r(const u64* src, u64* dst, int count, bool inverse) {
    if (!is_bfv()) throw Error("mkhe: not a BFV context");
    NttBatch b{};
    b.src = src; b.dst = dst; b.mods = d_mods; b.psi = inverse ? d_psiinv : d_psi; b.aux = d_inv_aux;
    b.nslots = 2 * nq;
    for (int j = 0; j < 2 * nq; ++j) { b.mod[j] = j < nq ? j : mtot + (j - nq); b.pos[j] = j; }
    b.src_outer = b.dst_outer = 2L * nq * N; b.src_inner = b.dst_inner = N;
    b.nouter = count;
    if (inverse) { ProfScope ps(this, PROF_NTT_INV, 16.0 * N * count * 2 * nq); ntt_inv_launch(b); }
    else {
        // ring-R polynomials out of ModUpQtoR / Rescale are lazy multSum representatives (< 3q, mkbfv/basis_extension.go:54-62,91-96): the
        // 59/60-bit reduction schedule of the H16 kernel must not assume inputs below 2^60 (NttBatch::src_lazy only selects that schedule here)
        b.src_lazy = 1;
        ntt_fwd_launch(b, false);
    }
    MKHE_HIP(hipGetLastError());
}

// conv.ModUpQtoR (mkbfv/basis_extension.go:49-64): Q part copied, QMul part = lazy ModUpQtoP
void Context::bfv_modup_q_to_r(const u64* polyq, u64* polyr, int npolys) {
    if (!is_bfv()) throw Error("mkhe: not a BFV context");
    BasisConvArgs a{};
    a.src = polyq; a.src_poly = (long)nq * N;
    a.dst = polyr + (size_t)nq * N; a.dst_poly = 2L * nq * N;
    a.copy_dst = polyr; a.copy_poly = 2L * nq * N;
    a.mods_s = d_mods; a.mods_t = d_mods + mtot;
    a.t = BasisConvTables{d_bq_qoverqiinvqi, d_bq_qoverqimodp, d_bq_vtimes};
    a.ns = nq; a.nt = nq; a.N = N; a.npolys = npolys;
    { ProfScope ps(this, PROF_BASISCONV, 8.0 * N * npolys * 3.0 * nq); launch_basis_conv(a, s_); }
    MKHE_HIP(hipGetLastError());
}

// conv.Rescale (basis_extension.go:82-96): QMul part = ModDownQPtoP(x * QMul mod Q, 0), Q part = lazy ModUpPtoQ of it
void Context::bfv_rescale(const u64* polyq, u64* polyr, int npolys) {
    if (!is_bfv()) throw Error("mkhe: not a BFV context");
    BasisConvArgs a{};
    a.src = polyq; a.src_poly = (long)nq * N;
    a.dst = polyr + (size_t)nq * N; a.dst_poly = 2L * nq * N;
    a.mods_s = d_mods; a.mods_t = d_mods + mtot;
    a.prescale = d_mform_qmul; a.downparam = d_down_q_in_m;
    a.t = BasisConvTables{d_bq_qoverqiinvqi, d_bq_qoverqimodp, d_bq_vtimes};
    a.ns = nq; a.nt = nq; a.N = N; a.npolys = npolys;
    { ProfScope ps(this, PROF_BASISCONV, 8.0 * N * npolys * 2.0 * nq); launch_basis_conv(a, s_); }
    BasisConvArgs c{};
    c.src = polyr + (size_t)nq * N; c.src_poly = 2L * nq * N;
    c.dst = polyr; c.dst_poly = 2L * nq * N;
    c.mods_s = d_mods + mtot; c.mods_t = d_mods;
    c.t = BasisConvTables{d_bm_qoverqiinvqi, d_bm_qoverqimodp, d_bm_vtimes};
    c.ns = nq; c.nt = nq; c.N = N; c.npolys = npolys;
    { ProfScope ps(this, PROF_BASISCONV, 8.0 * N * npolys * 2.0 * nq); launch_basis_conv(c, s_); }
    MKHE_HIP(hipGetLastError());
}

// conv.Quantize (basis_extension.go:66-80) AFTER the MulScalar(t): InvNTT over R (in place), ModDownQPtoQ with QMul as "P".
// The MulScalar itself is folded into the producer (tensor kernel) or done by the caller (bfv_quantize_full).
static void quantize_tail_args(BasisConvArgs& a, const Context& c, u64* polyr, u64* polyq, int npolys) {
    a.src = polyr + (size_t)c.nq * c.N; a.src_poly = 2L * c.nq * c.N;
    a.xsub = polyr; a.xsub_poly = 2L * c.nq * c.N;
    a.dst = polyq; a.dst_poly = (long)c.nq * c.N;
    a.mods_s = c.d_mods + c.mtot; a.mods_t = c.d_mods;
    a.downparam = c.d_down_m_in_q;
    a.t = BasisConvTables{c.d_bm_qoverqiinvqi, c.d_bm_qoverqimodp, c.d_bm_vtimes};
    a.ns = c.nq; a.nt = c.nq; a.N = c.N; a.npolys = npolys;
}
void Context::bfv_quantize(const u64* polyr_ntt, u64* polyq, int npolys) {
    if (!is_bfv()) throw Error("mkhe: not a BFV context");
    const size_t PR = 2 * (size_t)nq * N;
    u64* tmp = scratch(rbuf_, rbuf_words_, (size_t)npolys * PR);
    // scalar multiplication limb by limb: z = x * t  (MRed(x, MForm(t)))
    { ProfScope ps(this, PROF_OTHER, 16.0 * N * npolys * 2 * nq); launch_mul_const(tmp, polyr_ntt, d_mods, d_map_r, d_t_mont, 2 * nq, N, npolys, (long)PR, s_); }
    ntt_r(tmp, tmp, npolys, true);
    BasisConvArgs a{};
    quantize_tail_args(a, *this, tmp, polyq, npolys);
    { ProfScope ps(this, PROF_BASISCONV, 8.0 * N * npolys * 3.0 * nq); launch_basis_conv(a, s_); }
    MKHE_HIP(hipGetLastError());
}

// DecomposeBFV (mkbfv/keyswitch.go:67-90), alpha = 1: digit d = limb d of aR spread under Q and P and NTT'd
// (DecomposeSingleNTT); Q digits -> ad1, QMul digits -> ad2.  The QMul limbs of ModUpQtoR outputs and the Q limbs
// of Rescale outputs are lazy (< 3x their modulus): src_lazy.
void Context::bfv_decompose_batch(const std::vector<const u64*>& srcr, const std::vector<u64*>& ad1, const std::vector<u64*>& ad2, bool internal) {
    if (!is_bfv()) throw Error("mkhe: not a BFV context");
    const int level = nq - 1;
    for (int half = 0; half < 2; ++half) {
        for (size_t base = 0; base < srcr.size(); base += NTT_MAX_ITEMS) {
            const int n = (int)std::min<size_t>(NTT_MAX_ITEMS, srcr.size() - base);
            NttBatch b{};
            b.mods = d_mods; b.psi = d_psi; b.aux = d_inv_aux; slots_qp(b, level);
            b.src_outer = N; b.src_inner = 0; b.src_mapped = 0;
            b.dst_outer = (long)mtot * N; b.dst_inner = N; b.dst_mapped = 1;
            b.reduce_in = 1; b.reduce_src_mod_is_outer = 2; b.src_lazy = 1; b.skip_norm = internal ? 1 : 0;
            for (int d = 0; d < nq; ++d) b.outer_mod[d] = half ? mtot + d : d;
            b.nitems = n; b.outers_per_item = nq;
            for (int i = 0; i < n; ++i) {
                b.src_items[i] = srcr[base + i] + (half ? (size_t)nq * N : 0);
                b.dst_items[i] = half ? ad2[base + i] : ad1[base + i];
            }
            b.nouter = n * nq;
            ntt_fwd_launch(b, true);
        }
    }
    MKHE_HIP(hipGetLastError());
}

// ExternalProductBFV (mkbfv/keyswitch.go:83-113): DecomposeBFV into the engine's own pool (ks.swkPool1 / swkPool2 there), then the
// same sum over digits, InvNTTLazy and ModDownQPtoQ as the hoisted form
void Context::bfv_external_product(const u64* polyr, const u64* bg1, const u64* bg2, u64* c) {
    if (!is_bfv()) throw Error("mkhe: not a BFV context");
    u64* a1 = hoist_slot(3, 0).d; u64* a2 = hoist_slot(3, 1).d;
    bfv_decompose_batch({polyr}, {a1}, {a2}, true);
    bfv_external_product_hoisted(a1, a2, bg1, bg2, c);
}

// ExternalProductBFVHoisted (keyswitch_hoisted.go:6-34)
void Context::bfv_external_product_hoisted(const u64* ah1, const u64* ah2, const u64* bg1, const u64* bg2, u64* c) {
    if (!is_bfv()) throw Error("mkhe: not a BFV context");
    std::vector<ExtItem> items;
    ExtItem it{ah1, bg1, c, false}; it.ah2 = ah2; it.bg2 = bg2;
    items.push_back(it);
    ext_batch(nq - 1, items);
}

// Evaluator.MulRelinNew = mulRelinHoisted (mkbfv/evaluator.go:78-82,118-140) followed by
// KeySwitcher.MulAndRelinBFVHoisted (keyswitch_hoisted.go:36-206), in two phases so that the parties can be sharded over GPUs
// (mkhe_kklss_amd/dist.py ShardedBfvMulRelin):
//   bfv_mr_partial: ModUpQtoR / Rescale, tensor over R + Quantize (out_0 only where with_c0), DecomposeBFV of the party
//                   components, the partial sums x1, x2, y1, y2 (MForm'ed when mform, canonical partial sums otherwise);
//   bfv_mr_finish:  steps E and F with the complete x, y.
// Quantize rounds, so both tensor terms of an output slot (op0_0 * op1_j + op0_j * op1_0) have to be added before it: a rank
// must own whole parties (both components of every id it holds).
void Context::bfv_slots(const Ct& op0, const Ct& op1, const Ct& out, std::vector<int>& slot0, std::vector<int>& slot1) const {
    const int n0 = op0.n, n1 = op1.n;
    if (op0.limbs != nq || op1.limbs != nq || out.limbs != nq) throw Error("mkhe: BFV ciphertexts live at the maximum level");
    if (n0 > 32 || n1 > 32 || out.n > 32) throw Error("mkhe: too many parties");
    slot0.assign(n0, 0); slot1.assign(n1, 0);
    auto find = [&](int id) { for (int o = 0; o < out.n; ++o) if (out.ids[o] == id) return o; return -1; };
    std::vector<char> seen(out.n, 0);
    for (int a = 0; a < n0; ++a) { int o = find(op0.ids[a]); if (o < 0) throw Error("mkhe: ctOut lacks an id of op0"); slot0[a] = o; seen[o] = 1; }
    for (int a = 0; a < n1; ++a) { int o = find(op1.ids[a]); if (o < 0) throw Error("mkhe: ctOut lacks an id of op1"); slot1[a] = o; seen[o] = 1; }
    for (int o = 0; o < out.n; ++o) if (!seen[o]) throw Error("mkhe: ctOut has an id that neither operand has");
}

void Context::bfv_mr_partial(const Ct& op0, const Ct& op1, const Swk* const* rlk_b1, const Swk* const* rlk_b2,
                             const Swk* const* rlk_d1, const Swk* const* rlk_d2, bool with_c0, bool mform, Ct& out,
                             u64* x1, u64* x2, u64* y1, u64* y2, bool fuse_x, bool fuse_y) {
    if (!is_bfv()) throw Error("mkhe: not a BFV context");
    const int level = nq - 1, L = nq, n0 = op0.n, n1 = op1.n;
    bfv_xk1_.clear(); bfv_xk2_.clear(); bfv_yk1_.clear(); bfv_yk2_.clear();
    std::vector<int> slot0, slot1;
    bfv_slots(op0, op1, out, slot0, slot1);
    for (int a = 0; a < n0; ++a) if (!rlk_d1[a] || !rlk_d2[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
    for (int a = 0; a < n1; ++a) if (!rlk_b1[a] || !rlk_b2[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");

    const size_t PR = 2 * (size_t)nq * N;
    const int np0 = 1 + n0, np1 = 1 + n1, npo = 1 + out.n;
    // rbuf: [ct0R | ct1R | NTT(ct0R) | NTT(ct1R) | tensor out]
    u64* rb = scratch(rbuf_, rbuf_words_, (size_t)(2 * (np0 + np1) + npo) * PR);
    u64 *r0 = rb, *r1 = rb + (size_t)np0 * PR, *f0 = r1 + (size_t)np1 * PR, *f1 = f0 + (size_t)np0 * PR, *tz = f1 + (size_t)np1 * PR;
    bfv_modup_q_to_r(op0.d, r0, np0);
    bfv_rescale(op1.d, r1, np1);

    // tensor over R + Quantize on the side stream (needs only ct0R / ct1R)
    fork_side(1);
    s_ = overlap ? stream2 : stream;
    {
        ntt_r(r0, f0, np0 + np1, false);        // f0, f1 are contiguous like r0, r1
        TensorArgs ta{};
        ta.a0 = f0; ta.b0 = f1; ta.out = tz; ta.mods = d_mods; ta.map = d_map_r; ta.scale = d_t_mont;
        ta.nout = out.n; ta.L = 2 * nq; ta.N = N; ta.with_c0 = with_c0 ? 1 : 0;
        for (int a = 0; a < n0; ++a) { ta.a[1 + slot0[a]] = f0 + (size_t)(1 + a) * PR; ta.a_ls[1 + slot0[a]] = N; }
        for (int a = 0; a < n1; ++a) { ta.b[1 + slot1[a]] = f1 + (size_t)(1 + a) * PR; ta.b_ls[1 + slot1[a]] = N; }
        { ProfScope ps(this, PROF_TENSOR, 8.0 * N * 2 * nq * (2.0 + n0 + n1 + npo)); launch_tensor(ta, s_); }
        ntt_r(tz, tz, npo, true);
        BasisConvArgs qa{};
        quantize_tail_args(qa, *this, tz, out.d, npo);
        { ProfScope ps(this, PROF_BASISCONV, 8.0 * N * npo * 3.0 * nq); launch_basis_conv(qa, s_); }
    }
    side_done(1);
    s_ = stream;

    // hoisted forms of the party components (DecomposeBFV of every id of both operands, evaluator.go:126-133)
    std::vector<const u64*> h0a(n0), h0b(n0), h1a(n1), h1b(n1);
    {
        std::vector<const u64*> src; std::vector<u64*> d1, d2;
        for (int a = 0; a < n0; ++a) {
            Swk& s1 = hoist_slot(0, a); Swk& s2 = hoist_slot(3, a);
            src.push_back(r0 + (size_t)(1 + a) * PR); d1.push_back(s1.d); d2.push_back(s2.d); h0a[a] = s1.d; h0b[a] = s2.d;
        }
        for (int a = 0; a < n1; ++a) {
            Swk& s1 = hoist_slot(1, a); Swk& s2 = hoist_slot(4, a);
            src.push_back(r1 + (size_t)(1 + a) * PR); d1.push_back(s1.d); d2.push_back(s2.d); h1a[a] = s1.d; h1b[a] = s2.d;
        }
        if (!src.empty()) bfv_decompose_batch(src, d1, d2, true);
    }
    // y1, y2 on the main stream (they feed step F, the long chain), x1, x2 on the side stream (step E joins the last
    // batch)   (keyswitch_hoisted.go:76-126)
    const int nslots = L + np;
    // single-device evaluation with 1..4 parties in op0: x1, x2 come out of step F1 as by-products of the digits it holds anyway
    // (Context::mul_and_relin does the same for mkckks) -- two inner-product launches and one pass over h1(c0_i), h2(c0_i) less
    if (fuse_x) {
        if (!mform) throw Error("mkhe: internal: the fused x is produced in Montgomery form");
        for (int a = 0; a < n0; ++a) { bfv_xk1_.push_back(rlk_d1[a]->d); bfv_xk2_.push_back(rlk_d2[a]->d); }
    }
    // ... and y1, y2 (and step E) inside it as well, when op1 has as many parties (Context::mul_and_relin, round 4)
    if (fuse_y) {
        if (!fuse_x || n1 < 1 || n1 > 4 || n0 > 4) throw Error("mkhe: internal: y inside the F1 kernel needs the x by-product and one to four parties per operand");
        for (int a = 0; a < n1; ++a) { bfv_yk1_.push_back(rlk_b1[a]->d); bfv_yk2_.push_back(rlk_b2[a]->d); }
    }
    for (int which = fuse_y ? 1 : 3; which >= (fuse_x ? 2 : 0); --which) {
        const int side = which >> 1, half = which & 1;
        const int n = side ? n1 : n0;
        InnerProductArgs ip{};
        for (int a = 0; a < n; ++a) {
            const Swk* key = side ? (half ? rlk_b2[a] : rlk_b1[a]) : (half ? rlk_d2[a] : rlk_d1[a]);
            ip.a[a] = key->d;
            ip.b[a] = side ? (half ? h1b[a] : h1a[a]) : (half ? h0b[a] : h0a[a]);
        }
        ip.out = side ? (half ? y2 : y1) : (half ? x2 : x1);
        ip.mods = d_mods; ip.map = map_qp(level);
        ip.term_outer = ip.out_outer = (long)mtot * N; ip.nterms = n; ip.nslots = nslots; ip.nouter = beta_max; ip.N = N; ip.mform_out = mform ? 1 : 0;
        const bool on_side = side == 0 && overlap;
        if (which == 1 && on_side) fork_side(2);
        if (on_side) s_ = stream2;
        { ProfScope ps(this, PROF_INNER, 8.0 * N * nslots * beta_max * (2.0 * n + 1)); launch_inner_product(ip, s_); }
        if (on_side) s_ = stream;
        if (which == 0 && on_side) side_done(2);
    }
    // split-phase callers read x1, x2 between the phases (cross-device reduction): the side chain joins the main stream here
    if (!mform) join_side(2);
    bfv_plan_valid_ = true;
    MKHE_HIP(hipGetLastError());
}

void Context::bfv_mr_finish(const Ct& op0, const Ct& op1, const u64* x1, const u64* x2, const u64* y1, const u64* y2,
                            const Swk* const* rlk_v, const Swk& crs_u, Ct& out) {
    if (!is_bfv()) throw Error("mkhe: not a BFV context");
    if (!bfv_plan_valid_) throw Error("mkhe: bfv_mr_finish without bfv_mr_partial");
    const int level = nq - 1, n0 = op0.n, n1 = op1.n;
    std::vector<int> slot0, slot1;
    bfv_slots(op0, op1, out, slot0, slot1);
    for (int a = 0; a < n0; ++a) if (!rlk_v[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
    const size_t PQ = (size_t)nq * N;
    // F1: t_i = <h(c0_i), (y1,y2)>
    u64* tbuf = scratch(tbuf_, tbuf_words_, (size_t)n0 * PQ);
    std::vector<ExtItem> items;
    const bool fused = !bfv_xk1_.empty();
    for (int a = 0; a < n0; ++a) {
        ExtItem it{hoist_slot(0, a).d, y1, tbuf + (size_t)a * PQ, false}; it.ah2 = hoist_slot(3, a).d; it.bg2 = y2;
        if (fused) { it.xkey = bfv_xk1_[a]; it.xkey2 = bfv_xk2_[a]; }
        items.push_back(it);
    }
    if (fused) { ext_xout_ = const_cast<u64*>(x1); ext_xout2_ = const_cast<u64*>(x2); }
    bool e_done = false, f2 = false;
    if (fused && !bfv_yk1_.empty()) {
        ext_ykeys_ = bfv_yk1_; ext_ykeys2_ = bfv_yk2_;
        for (int a = 0; a < n1; ++a) { ext_yh_.push_back(hoist_slot(1, a).d); ext_yh2_.push_back(hoist_slot(4, a).d); }
        const int fuse_e_env = ab_fuse_e();
        if (fuse_e_env && 2 * n0 + n1 <= EXT_MAX_ITEMS) {
            // (step F2 is the plain Q gadget here too, keyswitch_hoisted.go:199-204: with step E done by the F1 kernel the digits of the t_i stay in the
            // registers of ntt16_f2_kernel, as in Context::mr_finish_head; its parts live behind the tail batch's items in this same allocation)
            f2 = n0 >= 2 && f2_fused_ok(level, n0, n1);
            const int f2_extra = f2 ? 2 * n0 * (f2_schedule(n0, level).parts - 1) : 0;
            scratch(c1b_, c1b_words_, (size_t)(2 * n0 + n1 + f2_extra) * mtot * N);
            ext_e_slot_ = 2 * n0;
        }
    }
    auto clear_xy = [&] { ext_xout_ = ext_xout2_ = nullptr; ext_ykeys_.clear(); ext_ykeys2_.clear(); ext_yh_.clear(); ext_yh2_.clear(); ext_e_slot_ = -1; };
    try { ext_batch(level, items); } catch (...) { clear_xy(); throw; }
    e_done = ext_e_slot_ >= 0;
    clear_xy();
    bfv_xk1_.clear(); bfv_xk2_.clear(); bfv_yk1_.clear(); bfv_yk2_.clear();
    // F2: ks.Decompose(t_i) ; out_0 += <h(t_i), v_i> ; out_i += <h(t_i), u>
    {
        std::vector<const u64*> dsrc; std::vector<u64*> ddst;
        f2 = f2 && e_done;
        for (int a = 0; a < n0; ++a) { dsrc.push_back(tbuf + (size_t)a * PQ); if (!f2) ddst.push_back(hoist_slot(2, a).d); }
        if (f2) ext_f2_src_.assign(dsrc.begin(), dsrc.end());
        else if (n0) decompose_batch(level, dsrc, ddst, true);
    }
    // E: out_j += <h(c1_j), (x1,x2)> together with F2
    items.clear();
    // (the F2 pairs first, as in Context::mr_finish_tail: out_0 is the longest ModDown group)
    for (int a = 0; a < n0; ++a) {
        const u64* ht = f2 ? tbuf + (size_t)a * PQ : hoist_slot(2, a).d;
        items.push_back(ExtItem{ht, rlk_v[a]->d, out.d, true});
        if (f2) { items.back().f2_party = a; items.back().f2_key = 0; }
        items.push_back(ExtItem{ht, crs_u.d, out.d + (size_t)(1 + slot0[a]) * PQ, true});
        if (f2) { items.back().f2_party = a; items.back().f2_key = 1; }
    }
    for (int a = 0; a < n1; ++a) {
        ExtItem it{hoist_slot(1, a).d, x1, out.d + (size_t)(1 + slot1[a]) * PQ, true}; it.ah2 = hoist_slot(4, a).d; it.bg2 = x2; it.pre = e_done; items.push_back(it);
    }
    join_side(2);
    try { ext_batch(level, items, 1); } catch (...) { ext_f2_src_.clear(); throw; }        // joins the tensor / Quantize chain before the ModDown accumulates into out
    ext_f2_src_.clear();
    bfv_plan_valid_ = false;
    MKHE_HIP(hipGetLastError());
}

void Context::bfv_mul_relin(const Ct& op0, const Ct& op1, const Swk* const* rlk_b1, const Swk* const* rlk_b2,
                            const Swk* const* rlk_d1, const Swk* const* rlk_d2, const Swk* const* rlk_v,
                            const Swk& crs_u, Ct& out) {
    for (int a = 0; a < op0.n; ++a) if (!rlk_v[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
    const int fuse_env = ab_fuse_x();
    const bool fuse = fuse_env && op0.n >= 1 && op0.n <= 4 && !masked_;
    const int fuse_y_env = ab_fuse_y();
    const bool fuse_y = fuse && fuse_y_env && op1.n >= 1 && op1.n <= 4;
    bfv_mr_partial(op0, op1, rlk_b1, rlk_b2, rlk_d1, rlk_d2, true, true, out, x_, x2_, y_, y2_, fuse, fuse_y);
    bfv_mr_finish(op0, op1, x_, x2_, y_, y2_, rlk_v, crs_u, out);
}

// Evaluator.mulRelin (mkbfv/evaluator.go:95-113) -> KeySwitcher.MulAndRelinBFV (mkbfv/keyswitch.go:115-251): the NON-hoisted twin, restated
// in the reference's own order with its pool discipline -- ONE pair of digit vectors (ks.swkPool1 / swkPool2) that every DecomposeBFV
// overwrites, so every party component is decomposed twice (once for its term of x or y, once inside ExternalProductBFV) and the digit
// scratch is 2 vectors instead of 4k; x1, x2, y1, y2 grow party by party through MulCoeffsMontgomeryAndAdd (InnerProductArgs::addend) and
// are MForm'ed by the call of the last party; every ExternalProductBFV / ExternalProduct is its own Decompose + inner product + InvNTTLazy +
// ModDown + AddLvl (no batching over parties, no merged ModDown, no x by-product).  Same integers as bfv_mul_relin (the products are exact
// residues and every sum is canonical), checked bit for bit on the device in tests/test_gpu_bfv.py.
void Context::bfv_mul_relin_unhoisted(const Ct& op0, const Ct& op1, const Swk* const* rlk_b1, const Swk* const* rlk_b2,
                                      const Swk* const* rlk_d1, const Swk* const* rlk_d2, const Swk* const* rlk_v,
                                      const Swk& crs_u, Ct& out) {
    if (!is_bfv()) throw Error("mkhe: not a BFV context");
    const int level = nq - 1, L = nq, n0 = op0.n, n1 = op1.n;
    std::vector<int> slot0, slot1;
    bfv_slots(op0, op1, out, slot0, slot1);
    for (int a = 0; a < n0; ++a) if (!rlk_d1[a] || !rlk_d2[a] || !rlk_v[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
    for (int a = 0; a < n1; ++a) if (!rlk_b1[a] || !rlk_b2[a]) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
    const size_t PR = 2 * (size_t)nq * N, PQ = (size_t)nq * N;
    const int np0 = 1 + n0, np1 = 1 + n1, npo = 1 + out.n;
    u64* rb = scratch(rbuf_, rbuf_words_, (size_t)(2 * (np0 + np1) + npo) * PR);
    u64 *r0 = rb, *r1 = rb + (size_t)np0 * PR, *f0 = r1 + (size_t)np1 * PR, *f1 = f0 + (size_t)np0 * PR, *tz = f1 + (size_t)np1 * PR;
    bfv_modup_q_to_r(op0.d, r0, np0);                  // evaluator.go:102-105
    bfv_rescale(op1.d, r1, np1);                       // evaluator.go:107-110
    u64* p1 = hoist_slot(3, 0).d; u64* p2 = hoist_slot(3, 1).d;        // ks.swkPool1, ks.swkPool2
    const int nslots = L + np;
    // x1, x2 (keyswitch.go:157-171), then y1, y2 (:173-187)
    for (int side = 0; side < 2; ++side) {
        const int n = side ? n1 : n0;
        u64* s1 = side ? y_ : x_; u64* s2 = side ? y2_ : x2_;
        if (n == 0) { MKHE_HIP(hipMemsetAsync(s1, 0, swk_words() * sizeof(u64), s_)); MKHE_HIP(hipMemsetAsync(s2, 0, swk_words() * sizeof(u64), s_)); }
        for (int a = 0; a < n; ++a) {
            bfv_decompose_batch({(side ? r1 : r0) + (size_t)(1 + a) * PR}, {p1}, {p2}, true);
            for (int half = 0; half < 2; ++half) {
                InnerProductArgs ip{};
                ip.a[0] = (side ? (half ? rlk_b2[a] : rlk_b1[a]) : (half ? rlk_d2[a] : rlk_d1[a]))->d;
                ip.b[0] = half ? p2 : p1;
                ip.out = half ? s2 : s1;
                ip.addend = a ? ip.out : nullptr;          // the pool vector was zeroed at :146-155
                ip.mods = d_mods; ip.map = map_qp(level);
                ip.term_outer = ip.out_outer = (long)mtot * N; ip.nterms = 1; ip.nslots = nslots; ip.nouter = beta_max; ip.N = N;
                ip.mform_out = a == n - 1 ? 1 : 0;         // MFormLvl after the last party (:168-171, :184-187)
                { ProfScope ps(this, PROF_INNER, 8.0 * N * nslots * beta_max * (a ? 4.0 : 3.0)); launch_inner_product(ip, s_); }
            }
        }
    }
    // tensor over R + Quantize (:189-233), in the batched form of bfv_mr_partial (same kernels, same integers)
    {
        ntt_r(r0, f0, np0 + np1, false);
        TensorArgs ta{};
        ta.a0 = f0; ta.b0 = f1; ta.out = tz; ta.mods = d_mods; ta.map = d_map_r; ta.scale = d_t_mont;
        ta.nout = out.n; ta.L = 2 * nq; ta.N = N; ta.with_c0 = 1;
        for (int a = 0; a < n0; ++a) { ta.a[1 + slot0[a]] = f0 + (size_t)(1 + a) * PR; ta.a_ls[1 + slot0[a]] = N; }
        for (int a = 0; a < n1; ++a) { ta.b[1 + slot1[a]] = f1 + (size_t)(1 + a) * PR; ta.b_ls[1 + slot1[a]] = N; }
        { ProfScope ps(this, PROF_TENSOR, 8.0 * N * 2 * nq * (2.0 + n0 + n1 + npo)); launch_tensor(ta, s_); }
        ntt_r(tz, tz, npo, true);
        BasisConvArgs qa{};
        quantize_tail_args(qa, *this, tz, out.d, npo);
        { ProfScope ps(this, PROF_BASISCONV, 8.0 * N * npo * 3.0 * nq); launch_basis_conv(qa, s_); }
    }
    // ctOut_j += ExternalProductBFV(op1_j, x1, x2)   (:235-239)
    for (int a = 0; a < n1; ++a) {
        bfv_decompose_batch({r1 + (size_t)(1 + a) * PR}, {p1}, {p2}, true);
        ExtItem it{p1, x_, out.d + (size_t)(1 + slot1[a]) * PQ, true}; it.ah2 = p2; it.bg2 = x2_;
        ext_batch(level, {it});
    }
    // t = ExternalProductBFV(op0_i, y1, y2); ctOut_0 += ExternalProduct(t, v_i); ctOut_i += ExternalProduct(t, u)   (:241-250)
    u64* t = scratch(tbuf_, tbuf_words_, PQ);
    for (int a = 0; a < n0; ++a) {
        bfv_decompose_batch({r0 + (size_t)(1 + a) * PR}, {p1}, {p2}, true);
        ExtItem it{p1, y_, t, false}; it.ah2 = p2; it.bg2 = y2_;
        ext_batch(level, {it});
        external_product(level, false, t, rlk_v[a]->d, out.d, true);
        external_product(level, false, t, crs_u.d, out.d + (size_t)(1 + slot0[a]) * PQ, true);
    }
    MKHE_HIP(hipGetLastError());
}

}  // namespace mkhe
