// mirror_check.cpp -- TEST INFRASTRUCTURE: drives the C++ host mirror (include/mkhe.hpp) the way a user of the Go
// reference would (mkckks.Evaluator.MulRelinNew, RotateHoistedNew, mkbfv.Evaluator.MulRelinNew on two parties) and checks
// every result bit for bit against the CPU oracle (oracle/ora_*.h) on the same seeded uniform inputs.
//   g++ -std=c++17 -I include -I oracle tests/cpp/mirror_check.cpp -L mkhe-kklss_amd/lib -lmkhe_hip -L oracle/_build -lmkhe_oracle
// Built by tests/test_cpp_mirror.py (compile + link on CPU; run on the GPU box).
#include <cstdio>
#include <cstring>
#include "mkhe.hpp"
extern "C" {
#include "ora_keygen.h"
}

typedef std::vector<uint64_t> vec;
static uint64_t rng_state = 0x4D4B4845ull;
static uint64_t next64() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static void fill_poly(uint64_t* p, const vec& mods, size_t N) { for (size_t l = 0; l < mods.size(); ++l) for (size_t i = 0; i < N; ++i) p[l * N + i] = next64() % mods[l]; }
static vec rand_swk(const vec& Q, const vec& P, size_t N) {
    vec QP(Q); QP.insert(QP.end(), P.begin(), P.end());
    vec s(Q.size() * QP.size() * N);                      // alpha = 1: beta = nQ digits
    for (size_t i = 0; i < Q.size(); ++i) fill_poly(&s[i * QP.size() * N], QP, N);
    return s;
}
static vec rand_ct(const vec& Q, int n, size_t N) { vec c((1 + n) * Q.size() * N); for (int s = 0; s <= n; ++s) fill_poly(&c[s * Q.size() * N], Q, N); return c; }
static int fails = 0;
static void expect(bool ok, const char* what) { std::printf("%-64s %s\n", what, ok ? "ok" : "MISMATCH"); if (!ok) ++fails; }

int main() {
    const vec Q = {0xfffffffff6a0001ull, 0x3fffffffd60001ull, 0x3fffffffca0001ull, 0x3fffffff6d0001ull};
    const vec P = {0x7ffffffffe70001ull, 0x7ffffffffe10001ull};
    const std::vector<std::string> names = {"alice", "bob"};
    {   // ---------------- mkckks: MulRelinNew (hoist + MulAndRelinHoisted + Rescale) and RotateHoistedNew
        const int logN = 12; const size_t N = 1u << logN; const int level = (int)Q.size() - 1;
        const double scale = 18014398509481984.0;   // 2^54
        mkckks::Parameters params(logN, Q, P, scale);
        mkckks::Evaluator eval(params);
        vec h0 = rand_ct(Q, 2, N), h1 = rand_ct(Q, 2, N), u = rand_swk(Q, P, N);
        params.AddCRS(-1, u.data());
        mkrlwe::RelinearizationKeySet rlkSet;
        std::vector<vec> kb, kd, kv;
        for (auto& n : names) {
            kb.push_back(rand_swk(Q, P, N)); kd.push_back(rand_swk(Q, P, N)); kv.push_back(rand_swk(Q, P, N));
            rlkSet.AddRelinearizationKey(std::make_shared<mkrlwe::RelinearizationKey>(params, n, kb.back().data(), kd.back().data(), kv.back().data()));
        }
        mkrlwe::IDSet ids(names.begin(), names.end());
        mkckks::Ciphertext ct0(params, ids, level, scale), ct1(params, ids, level, scale);
        ct0.upload(h0.data()); ct1.upload(h1.data());
        auto res = eval.MulRelinNew(ct0, ct1, rlkSet);
        vec got(res->words());
        res->download(got.data());

        ora_ks* ks = ora_ks_new(logN, Q.data(), (int)Q.size(), P.data(), (int)P.size(), 2, nullptr, nullptr);
        const int idv[2] = {0, 1};
        const uint64_t *pb[2] = {kb[0].data(), kb[1].data()}, *pd[2] = {kd[0].data(), kd[1].data()}, *pv[2] = {kv[0].data(), kv[1].data()};
        vec full(3 * Q.size() * N);
        ora_mul_and_relin(ks, level, 2, idv, h0.data(), (int)Q.size(), 2, idv, h1.data(), (int)Q.size(), nullptr, nullptr, pb, pd, pv, u.data(), 2, idv, full.data());
        double sc = scale * scale;
        const int nb = ora_ckks_nb_rescales(ora_ks_ringq(ks), level, &sc, scale);
        vec ref(3 * (Q.size() - nb) * N);
        for (int s = 0; s < 3; ++s) ora_div_round_last_many(ora_ks_ringq(ks), level, nb, &full[s * Q.size() * N], &ref[s * (Q.size() - nb) * N]);
        expect(res->Level() == level - nb && res->Scale == sc && got == ref, "mkckks.Evaluator.MulRelinNew (2 parties, N=2^12)");

        const int rot = 3;
        vec crs = rand_swk(Q, P, N);
        params.AddCRS(rot, crs.data());
        mkrlwe::RotationKeySet rkSet;
        std::vector<vec> rk;
        for (auto& n : names) { rk.push_back(rand_swk(Q, P, N)); rkSet.AddRotationKey(std::make_shared<mkrlwe::RotationKey>(params, rot, n, rk.back().data())); }
        auto hoisted = eval.HoistedForm(ct0);
        auto rres = eval.RotateHoistedNew(ct0, rot, *hoisted, rkSet);
        vec rgot(rres->words()), rref(3 * Q.size() * N);
        rres->download(rgot.data());
        const uint64_t* prk[2] = {rk[0].data(), rk[1].data()};
        ora_rotate(ks, level, params.GaloisElementForColumnRotationBy(rot), 2, idv, h0.data(), (int)Q.size(), nullptr, prk, crs.data(), rref.data());
        expect(rgot == rref, "mkckks.Evaluator.RotateHoistedNew");
        auto sum = eval.AddNew(ct0, ct1);
        vec sgot(sum->words());
        sum->download(sgot.data());
        bool okadd = true;
        for (size_t s = 0; s < 3 && okadd; ++s) for (size_t l = 0; l < Q.size() && okadd; ++l) for (size_t i = 0; i < N; ++i) {
            const size_t e = (s * Q.size() + l) * N + i; uint64_t v = h0[e] + h1[e]; if (v >= Q[l]) v -= Q[l];
            if (sgot[e] != v) { okadd = false; break; }
        }
        expect(okadd, "mkckks.Evaluator.AddNew");
        {   // unequal scales: the operand with the smaller scale is multiplied by floor(ratio) first (evaluator.go:214-281)
            const double keep = ct0.Scale;
            ct0.Scale = keep / 5.25;                       // floor(ratio) = 5
            auto d = eval.SubNew(ct0, ct1);
            vec dgot(d->words());
            d->download(dgot.data());
            bool ok = d->Scale == ct1.Scale;
            for (size_t s = 0; s < 3 && ok; ++s) for (size_t l = 0; l < Q.size() && ok; ++l) for (size_t i = 0; i < N; ++i) {
                const size_t e = (s * Q.size() + l) * N + i;
                const uint64_t a5 = (uint64_t)(((unsigned __int128)h0[e] * 5) % Q[l]);
                uint64_t v = a5 + Q[l] - h1[e]; if (v >= Q[l]) v -= Q[l];
                if (dgot[e] != v) { ok = false; break; }
            }
            ct0.Scale = keep;
            expect(ok, "mkckks.Evaluator.SubNew with unequal scales (MultByConst first)");
        }
        bool threw = false;
        try { mkrlwe::RelinearizationKeySet empty; eval.MulRelinNew(ct0, ct1, empty); } catch (const mkhe::Error& e) { threw = std::strstr(e.what(), "cannot GetRelinearizationKey") != nullptr; }
        expect(threw, "missing relinearization key raises the reference's panic text");
        // ---- mkrlwe.KeyGenerator on the device with a CRS expanded from a public seed, against the oracle on the same samples
        auto a = params.AddCRS(0, (uint64_t)0x5EED), uu = params.AddCRS(-1, (uint64_t)0x5EED), a3 = params.AddCRS(3, (uint64_t)0x5EED);
        const size_t beta = Q.size();
        vec ah(u.size()), uh(u.size()), a3h(u.size()), oexp(u.size());
        a->download(ah.data()); uu->download(uh.data()); a3->download(a3h.data());
        ora_crs_expand(ks, 0x5EED, 0, oexp.data());
        expect(ah == oexp, "Parameters.AddCRS(idx, seed): CRS expanded on the device");
        std::vector<int32_t> s(N), r(N), e(3 * beta * N);
        for (auto& x : s) x = (int32_t)(next64() % 3) - 1;
        for (auto& x : r) x = (int32_t)(next64() % 3) - 1;
        for (auto& x : e) x = (int32_t)(next64() % 39) - 19;
        mkrlwe::KeyGenerator kgen(params);
        auto sk = kgen.GenSecretKey("alice", s.data()), rr = kgen.GenSecretKey("alice", r.data());
        auto gen = kgen.GenRelinearizationKey(*sk, *rr, e.data());
        vec skh((Q.size() + P.size()) * N), rh(skh.size()), ob(u.size()), od(u.size()), ov(u.size()), gb(u.size()), gd(u.size()), gv(u.size());
        ora_gen_secret_key(ks, s.data(), skh.data()); ora_gen_secret_key(ks, r.data(), rh.data());
        ora_gen_relin_key(ks, skh.data(), rh.data(), e.data(), ah.data(), uh.data(), ob.data(), od.data(), ov.data());
        gen->Value[0]->download(gb.data()); gen->Value[1]->download(gd.data()); gen->Value[2]->download(gv.data());
        expect(gb == ob && gd == od && gv == ov, "mkrlwe.KeyGenerator.GenRelinearizationKey");
        auto grk = kgen.GenRotationKey(rot, *sk, e.data());
        ora_gen_rotation_key(ks, params.GaloisElementForColumnRotationBy(rot), skh.data(), e.data(), a3h.data(), ob.data());
        grk->Value->download(gb.data());
        expect(gb == ob, "mkrlwe.KeyGenerator.GenRotationKey");
        ora_ks_free(ks);
    }
    {   // ---------------- mkbfv: MulRelinNew
        const int logN = 11; const size_t N = 1u << logN;
        const vec BQ = {0x3fffffffd60001ull, 0x3fffffff6d0001ull, 0x3fffffff550001ull}, BQM = {0x3fffffffca0001ull, 0x3fffffff5d0001ull, 0x3fffffff390001ull};
        const vec BP = {0xffffffffffc0001ull, 0xfffffffff840001ull};
        mkbfv::Parameters params(logN, BQ, BQM, BP, 65537);
        mkbfv::Evaluator eval(params);
        vec h0 = rand_ct(BQ, 2, N), h1 = rand_ct(BQ, 2, N), u = rand_swk(BQ, BP, N);
        params.AddCRS(-1, u.data());
        mkbfv::RelinearizationKeySet rlkSet;
        std::vector<vec> k[5];
        for (auto& n : names) {
            for (auto& kk : k) kk.push_back(rand_swk(BQ, BP, N));
            rlkSet.AddRelinearizationKey(std::make_shared<mkbfv::RelinearizationKey>(params, n, k[0].back().data(), k[1].back().data(), k[2].back().data(),
                                                                                     k[3].back().data(), k[4].back().data()));
        }
        mkrlwe::IDSet ids(names.begin(), names.end());
        mkbfv::Ciphertext ct0(params, ids), ct1(params, ids);
        ct0.upload(h0.data()); ct1.upload(h1.data());
        auto res = eval.MulRelinNew(ct0, ct1, rlkSet);
        vec got(res->words()), ref(res->words());
        res->download(got.data());
        ora_bfv* b = ora_bfv_new(logN, BQ.data(), BQM.data(), (int)BQ.size(), BP.data(), (int)BP.size(), 2, 65537);
        const int idv[2] = {0, 1};
        const uint64_t* p[5][2];
        for (int j = 0; j < 5; ++j) { p[j][0] = k[j][0].data(); p[j][1] = k[j][1].data(); }
        ora_bfv_mul_relin_new(b, 2, idv, h0.data(), 2, idv, h1.data(), p[0], p[1], p[2], p[3], p[4], u.data(), 1, 2, idv, ref.data());
        expect(got == ref, "mkbfv.Evaluator.MulRelinNew (2 parties, N=2^11)");
        ora_bfv_free(b);
    }
    std::printf("%s\n", fails ? "FAILED" : "all C++ mirror checks passed");
    return fails ? 1 : 0;
}
