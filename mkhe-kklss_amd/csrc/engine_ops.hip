// engine_ops.hip -- Context: Rotate[Hoisted] / Conjugate, Rescale, the elementwise evaluator operations
#include "engine.h"
#include <algorithm>
#include <cstring>
#include <cstdlib>

namespace mkhe {

// ------------------------------------------------------------------ Rotate[Hoisted] / Conjugate
// keyswitch.go:234-298, keyswitch_hoisted.go:183-247
void Context::rotate(u64 galEl, const Ct& in, const Swk* const* hoist, const Swk* const* rk, const Swk& crs, Ct& out) {
    const int L = out.limbs, n = in.n;
    if (n > 0 && 2 * n <= EXT_MAX_ITEMS && in.d != out.d) {
        // the signed permutation and the + c_0 ride on the ModDown of the external products: no staging copy of the
        // ciphertext, no separate permutation kernel
        rotate_core(in, hoist, rk, crs, true, out, galEl);
        return;
    }
    Ct tmp; tmp.n = n; tmp.limbs = L; tmp.ids = in.ids;
    tmp.d = scratch(ctbuf_, ctbuf_words_, (size_t)(1 + n) * L * N);
    rotate_partial(in, hoist, rk, crs, true, tmp);
    automorphism(galEl, tmp, out);
}

// Rotate without the final permutation: out_0 = [c_0 +] sum_i <h(c_i), rk_i>_P, out_i = <h(c_i), crs>_P
// (keyswitch.go:251-265).  with_c0 = false leaves c_0 out (party-sharded evaluation: one rank adds it).
void Context::rotate_partial(const Ct& in, const Swk* const* hoist, const Swk* const* rk, const Swk& crs, bool with_c0, Ct& out) {
    rotate_core(in, hoist, rk, crs, with_c0, out, 0);
}
void Context::rotate_core(const Ct& in, const Swk* const* hoist, const Swk* const* rk, const Swk& crs, bool with_c0, Ct& out, u64 galEl) {
    const int level = out.limbs - 1, L = level + 1, n = in.n;
    check_level(level);
    if (in.limbs < L) throw Error("Cannot Rotate: ctIn and ctOut have different levels");
    if (out.n != n || out.ids != in.ids) throw Error("mkhe: ctOut must carry the ids of ctIn");
    const size_t PI = (size_t)in.limbs * N, PO = (size_t)L * N;
    u64* tmp = out.d;
    const bool fused = galEl != 0;          // c_0 enters as the addend of the first accumulating item, stores are permuted
    if (fused && !with_c0) throw Error("mkhe: a fused rotation includes c_0");
    if (!fused) {
        if (with_c0) MKHE_HIP(hipMemcpyAsync(tmp, in.d, PO * sizeof(u64), hipMemcpyDeviceToDevice, s_));
        else MKHE_HIP(hipMemsetAsync(tmp, 0, PO * sizeof(u64), s_));
    }
    std::vector<const u64*> h(n);
    bool f2 = false;
    {
        std::vector<const u64*> dsrc; std::vector<u64*> ddst;
        for (int a = 0; a < n; ++a) {
            if (!rk[a]) throw Error("cannot GetRotationKeys: there is no rotation key with given id");
            if (hoist) { if (!hoist[a]) throw Error("mkhe: missing hoisted form"); h[a] = hoist[a]->d; }
            else { Swk& s = hoist_slot(0, a); dsrc.push_back(in.d + (1 + a) * PI); ddst.push_back(s.d); h[a] = s.d; }
        }
        // (a rotation of a ciphertext without a hoisted form: its digits are read once, by the two products below -- small launches finish their
        // transform inside the product kernel; N = 2^15 launches that fill the chip never store them at all: ntt16_f2_kernel, as step F2 of MulAndRelin)
        f2 = !hoist && n >= 2 && f2_fused_ok(level, n, 0);
        const bool stage = !f2 && !hoist && ext_fused_ok(level, (int)dsrc.size());
        if (f2) { ext_f2_src_.assign(dsrc.begin(), dsrc.end()); for (int a = 0; a < n; ++a) h[a] = dsrc[a]; }
        else if (!dsrc.empty()) decompose_batch(level, dsrc, ddst, true, stage);
        if (stage) ext_staged_.assign(ddst.begin(), ddst.end());
    }
    std::vector<ExtItem> items;
    for (int a = 0; a < n; ++a) {
        items.push_back(ExtItem{h[a], rk[a]->d, tmp, true});
        if (fused && a == 0) items.back().addend = in.d;
        if (f2) { items.back().f2_party = a; items.back().f2_key = 0; }
        items.push_back(ExtItem{h[a], crs.d, tmp + (size_t)(1 + a) * PO, false});
        if (f2) { items.back().f2_party = a; items.back().f2_key = 1; }
    }
    try { ext_batch(level, items, -1, 0, galEl); } catch (...) { ext_staged_.clear(); ext_f2_src_.clear(); staged_open_.clear(); throw; }
    ext_staged_.clear(); ext_f2_src_.clear();
    MKHE_HIP(hipGetLastError());
}

// signed coefficient permutation X -> X^galEl of every component (keyswitch.go:267-296)
void Context::automorphism(u64 galEl, const Ct& in, Ct& out) {
    const int L = out.limbs;
    if (in.limbs != L || in.n != out.n) throw Error("mkhe: automorphism operands differ in shape");
    if (in.d == out.d) throw Error("mkhe: automorphism cannot run in place");
    launch_automorphism(out.d, in.d, d_mods, L, logN, galEl, 1 + in.n, s_);
    MKHE_HIP(hipGetLastError());
}

// keyswitch.go:302-332
void Context::conjugate(u64 galEl, const Ct& in, const Swk* const* ck, const Swk& crs, Ct& out) {
    const int level = out.limbs - 1, L = level + 1, n = in.n;
    check_level(level);
    if (in.limbs < L) throw Error("Cannot Conjugate: ctIn and ctOut have different levels");
    if (out.n != n || out.ids != in.ids) throw Error("mkhe: ctOut must carry the ids of ctIn");
    const size_t PI = (size_t)in.limbs * N, PO = (size_t)L * N;
    u64* tmp = scratch(ctbuf_, ctbuf_words_, (size_t)(1 + n) * PO);
    if (in.limbs == L) launch_automorphism(tmp, in.d, d_mods, L, logN, galEl, 1 + n, s_);
    else for (int a = 0; a <= n; ++a) launch_automorphism(tmp + a * PO, in.d + a * PI, d_mods, L, logN, galEl, 1, s_);
    if (n == 0) { MKHE_HIP(hipMemcpyAsync(out.d, tmp, PO * sizeof(u64), hipMemcpyDeviceToDevice, s_)); return; }
    // all parties in one Decompose launch and one batch of external products, like Rotate; sigma(c_0) enters as the addend
    // of the first accumulating item (the permuted polynomials keep q for a sign-flipped 0, exactly what the reference
    // decomposes)
    std::vector<const u64*> dsrc; std::vector<u64*> ddst;
    for (int a = 0; a < n; ++a) {
        if (!ck[a]) throw Error("cannot GetConjugationKey: there is no conjugation key with given id");
        dsrc.push_back(tmp + (size_t)(1 + a) * PO); ddst.push_back(hoist_slot(0, a).d);
    }
    const bool f2 = n >= 2 && f2_fused_ok(level, n, 0);           // (as Rotate: the digits of the permuted polynomials stay in registers)
    const bool stage = !f2 && ext_fused_ok(level, n);
    if (f2) ext_f2_src_.assign(dsrc.begin(), dsrc.end());
    else decompose_batch(level, dsrc, ddst, true, stage);
    if (stage) ext_staged_.assign(ddst.begin(), ddst.end());
    std::vector<ExtItem> items;
    for (int a = 0; a < n; ++a) {
        items.push_back(ExtItem{f2 ? dsrc[a] : ddst[a], ck[a]->d, out.d, true});
        if (a == 0) items.back().addend = tmp;
        if (f2) { items.back().f2_party = a; items.back().f2_key = 0; }
        items.push_back(ExtItem{f2 ? dsrc[a] : ddst[a], crs.d, out.d + (size_t)(1 + a) * PO, false});
        if (f2) { items.back().f2_party = a; items.back().f2_key = 1; }
    }
    try { ext_batch(level, items); } catch (...) { ext_staged_.clear(); ext_f2_src_.clear(); staged_open_.clear(); throw; }
    ext_staged_.clear(); ext_f2_src_.clear();
    MKHE_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ Rescale body
// mkckks/evaluator.go:385-391 -> lattigo DivRoundByLastModulusManyLvl.  The reference's in-place
// "+ (q_L-1)/2" on the dropped limb of ctIn is NOT reproduced (ctIn stays untouched).
void Context::rescale(const Ct& in, int nb, Ct& out) {
    const int level = in.limbs - 1;
    check_level(level);
    if (nb < 0 || nb > level) throw Error("cannot Rescale: input Ciphertext already at level 0");
    if (out.limbs != in.limbs - nb || out.n != in.n) throw Error("mkhe: ctOut shape does not match the rescaled ciphertext");
    const int np_ = 1 + in.n;
    const size_t PI = (size_t)in.limbs * N, PO = (size_t)out.limbs * N;
    if (nb == 0) { if (out.d != in.d) MKHE_HIP(hipMemcpyAsync(out.d, in.d, np_ * PI * sizeof(u64), hipMemcpyDeviceToDevice, s_)); return; }
    // source and destination polynomials have different strides (in.limbs vs out.limbs): in place the threads of one polynomial
    // would overwrite limbs of the next one that other threads still read
    if (out.d == in.d) throw Error("cannot Rescale in place: ctOut must not alias ctIn when levels are dropped");
    if (nb == 1) {
        launch_div_round_last(out.d, in.d, d_mods, d_rescale + (size_t)(level - 1) * nq, level, N, np_, (long)PI, (long)PO, s_);
    } else {
        u64* tmp = scratch(ctbuf_, ctbuf_words_, (size_t)np_ * PI);
        launch_div_round_last(tmp, in.d, d_mods, d_rescale + (size_t)(level - 1) * nq, level, N, np_, (long)PI, (long)PI, s_);
        for (int k = 1; k < nb; ++k) {
            const int lv = level - k;
            const bool last = (k == nb - 1);
            launch_div_round_last(last ? out.d : tmp, tmp, d_mods, d_rescale + (size_t)(lv - 1) * nq, lv, N, np_,
                                  (long)PI, last ? (long)PO : (long)PI, s_);
        }
    }
    MKHE_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ elementwise evaluator ops
// evaluateInPlace of mkckks/evaluator.go:41-70 and mkbfv/evaluator.go:27-62: c_0 and the components both
// operands have are combined, the others are copied (Sub: negated when only op1 has them, :59-66).
void Context::ct_binary(int op, const Ct& a, const Ct& b, Ct& out) {
    const int L = out.limbs;
    if (a.limbs < L || b.limbs < L) throw Error("mkhe: operand level below ctOut level");
    const size_t PA = (size_t)a.limbs * N, PB = (size_t)b.limbs * N, PO = (size_t)L * N;
    auto find = [](const Ct& c, int id) { for (int i = 0; i < c.n; ++i) if (c.ids[i] == id) return i; return -1; };
    if (1 + out.n > CTBIN_MAX) throw Error("mkhe: too many parties in one ciphertext");
    CtBinArgs ba{};
    ba.mods = d_mods; ba.L = L; ba.N = N; ba.ncomp = 1 + out.n;
    double bytes = 0;
    for (int o = -1; o < out.n; ++o) {
        const int ia = o < 0 ? 0 : 1 + find(a, out.ids[o]), ib = o < 0 ? 0 : 1 + find(b, out.ids[o]);
        const bool ha = o < 0 || ia > 0, hb = o < 0 || ib > 0;
        if (!ha && !hb) throw Error("mkhe: ctOut has an id that neither operand has");
        const int c = 1 + o;
        ba.dst[c] = out.d + (size_t)c * PO;
        ba.a[c] = ha ? a.d + ia * PA : nullptr;
        ba.b[c] = hb ? b.d + ib * PB : nullptr;
        ba.mode[c] = (ha && hb) ? (op == 0 ? 0 : 1) : ha ? 2 : (op == 0 ? 3 : 4);
        bytes += 8.0 * N * L * ((ha && hb) ? 3 : 2);
    }
    { ProfScope ps(this, PROF_OTHER, bytes); launch_ct_binary(ba, s_); }
    MKHE_HIP(hipGetLastError());
}

void Context::ct_mul_const(const Ct& in, const u64* c_first, const u64* c_second, Ct& out) {
    const int L = std::min(in.limbs, out.limbs);                   // level := min(ct0.Level(), ctOut.Level()), :119
    if (in.n != out.n || in.ids != out.ids) throw Error("mkhe: ctOut must carry the ids of ct0");
    if (L > 48) throw Error("mkhe: too many limbs");
    MulConstArgs a{};
    a.src = in.d; a.dst = out.d; a.mods = d_mods; a.src_poly = (long)in.limbs * N; a.dst_poly = (long)out.limbs * N;
    a.L = L; a.N = N; a.npolys = 1 + in.n;
    for (int l = 0; l < L; ++l) {
        if (c_first[l] >= moduli[l] || c_second[l] >= moduli[l]) throw Error("mkhe: MultByConst constants must be reduced");
        a.c[0][l] = c_first[l]; a.c[1][l] = c_second[l];
    }
    { ProfScope ps(this, PROF_OTHER, 16.0 * N * L * (1 + in.n)); launch_mul_const_halves(a, s_); }
    MKHE_HIP(hipGetLastError());
}

void Context::ct_mul_ptxt(const Ct& in, const u64* dev_pt, Ct& out) {
    const int L = in.limbs, np_ = 1 + in.n;
    if (out.limbs != L || out.n != in.n || out.ids != in.ids) throw Error("mkhe: ctOut shape does not match ct");
    const size_t PO = (size_t)L * N;
    u64* tmp = scratch(ctbuf_, ctbuf_words_, (size_t)(1 + np_) * PO);
    ntt(dev_pt, tmp, 1, L, 0, false, false);
    ntt(in.d, tmp + PO, np_, L, 0, false, false);
    { ProfScope ps(this, PROF_OTHER, 8.0 * N * L * (2.0 * np_ + 1)); launch_mul_by_poly(tmp + PO, tmp + PO, tmp, d_mods, L, N, np_, s_); }
    ntt(tmp + PO, out.d, np_, L, 0, true, false);
    MKHE_HIP(hipGetLastError());
}

}  // namespace mkhe
