// mkhe.hpp -- C++ host-side mirror of the reference packages mkrlwe / mkckks / mkbfv over the C ABI (mkhe.h).
//
// The reference is Go; this image has no Go toolchain, so the compiled-language host layer a user of the reference
// would program against is given here in C++ (header-only, C++17): the same type and method names, argument meaning
// and error behaviour as the Go API (a Go `panic` becomes a thrown mkhe::Error carrying the same text), all polynomial
// data resident in HBM behind RAII handles.  Reference file:line per item; the un-built cgo equivalent is shim/go/.
// (mkhe-kklss_amd/*.py is the same mirror for the Python tests and the bench.)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>
#include "mkhe.h"

namespace mkhe {
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(int rc) { if (rc) throw Error(mkhe_last_error()); }
}  // namespace mkhe

namespace mkrlwe {
using mkhe::check;
using mkhe::Error;
typedef std::set<std::string> IDSet;                                   // mkrlwe/idset.go

class SwitchingKey;

// mkrlwe.Parameters (params.go:8-12): ring parameters + CRS map + gamma, plus the engine context
// (= NewKeySwitcher state, keyswitch.go:33-47).
class Parameters {
  public:
    Parameters(int logN, std::vector<uint64_t> Q, std::vector<uint64_t> P, int gamma = 2, int device = 0)
        : logN_(logN), Q_(std::move(Q)), P_(std::move(P)), gamma_(gamma) {
        check(mkhe_ctx_create(&ctx, logN, Q_.data(), (int)Q_.size(), P_.data(), (int)P_.size(), gamma, nullptr, nullptr, device));
    }
    virtual ~Parameters() { CRS.clear(); if (ctx) mkhe_ctx_destroy(ctx); }
    Parameters(const Parameters&) = delete;
    Parameters& operator=(const Parameters&) = delete;

    int N() const { return 1 << logN_; }
    int LogN() const { return logN_; }
    int QCount() const { return (int)Q_.size(); }
    int PCount() const { return (int)P_.size(); }
    int MaxLevel() const { return QCount() - 1; }
    int Gamma() const { return gamma_; }
    int Alpha() const { return PCount() / gamma_; }                                     // params.go:63-65
    int Beta(int levelQ) const { return (levelQ + 1 + Alpha() - 1) / Alpha(); }          // params.go:67-71
    size_t SwkWords() const { return mkhe_ctx_swk_words(ctx); }
    const std::vector<uint64_t>& Q() const { return Q_; }
    const std::vector<uint64_t>& P() const { return P_; }
    uint64_t GaloisElementForColumnRotationBy(int k) const {
        const uint64_t m = 2ull * N();
        uint64_t r = 1, b = 5, e = (uint64_t)(((k % (int)m) + (int)m) % (int)m);
        for (; e; e >>= 1) { if (e & 1) r = r * b % m; b = b * b % m; }
        return r;
    }
    uint64_t GaloisElementForRowRotation() const { return 2ull * N() - 1; }
    // params.AddCRS / NewParameters CRS slots (params.go:37-61): uniform polys (NTT + Montgomery form), uploaded once
    std::shared_ptr<SwitchingKey> AddCRS(int idx, const uint64_t* host_swk);
    // the same slot expanded on the device from a public seed (mkhe_crs_expand): nothing is uploaded
    std::shared_ptr<SwitchingKey> AddCRS(int idx, uint64_t seed);
    int party_index(const std::string& id) {
        if (id == "0") throw Error("Cannot IDSet Add : 0 cannot be used");                // idset.go:12-16
        auto it = ids_.find(id);
        if (it != ids_.end()) return it->second;
        const int v = (int)ids_.size();
        ids_[id] = v;
        return v;
    }
    void sync() { check(mkhe_ctx_sync(ctx)); }
    // work issued through this context from now on starts after everything issued through `other` so far (another context
    // over the same ring on the same device: own stream and scratch pools, all handles shared) -- mkhe_ctx_wait_for
    void wait_for(const Parameters& other) { check(mkhe_ctx_wait_for(ctx, other.ctx)); }

    mkhe_ctx* ctx = nullptr;
    std::map<int, std::shared_ptr<SwitchingKey>> CRS;                                    // params.go:37-46

  protected:
    Parameters(int logN, std::vector<uint64_t> Q, std::vector<uint64_t> P, int gamma, bool /*no context yet*/)
        : logN_(logN), Q_(std::move(Q)), P_(std::move(P)), gamma_(gamma) {}
    int logN_;
    std::vector<uint64_t> Q_, P_;
    int gamma_;
    std::map<std::string, int> ids_;
};

// mkrlwe.SwitchingKey (keys.go:23-25): host layout uint64[beta][nQ+nP][N]
class SwitchingKey {
  public:
    explicit SwitchingKey(Parameters& p, const uint64_t* host = nullptr) : params(p) {
        check(mkhe_swk_create(p.ctx, &h));
        if (host) upload(host);
    }
    ~SwitchingKey() { if (h) mkhe_swk_destroy(params.ctx, h); }
    SwitchingKey(const SwitchingKey&) = delete;
    SwitchingKey& operator=(const SwitchingKey&) = delete;
    void upload(const uint64_t* host) { check(mkhe_swk_upload(params.ctx, h, host)); }
    void download(uint64_t* host) const { check(mkhe_swk_download(params.ctx, h, host)); }
    Parameters& params;
    mkhe_swk* h = nullptr;
};
inline std::shared_ptr<SwitchingKey> NewSwitchingKey(Parameters& p) { return std::make_shared<SwitchingKey>(p); }   // keys.go:245-255
inline std::shared_ptr<SwitchingKey> Parameters::AddCRS(int idx, const uint64_t* host_swk) {
    return CRS[idx] = std::make_shared<SwitchingKey>(*this, host_swk);
}

inline std::shared_ptr<SwitchingKey> Parameters::AddCRS(int idx, uint64_t seed) {
    auto k = std::make_shared<SwitchingKey>(*this);
    check(mkhe_crs_expand(ctx, seed, idx, k->h));
    return CRS[idx] = k;
}

// raw device words behind mkhe_buf_* (secret / public keys: PolyQP buffers)
class DeviceWords {
  public:
    DeviceWords(Parameters& p, size_t words_) : params(p), words(words_) { check(mkhe_buf_alloc(p.ctx, words, &d)); }
    ~DeviceWords() { if (d) mkhe_buf_free(params.ctx, d); }
    DeviceWords(const DeviceWords&) = delete;
    DeviceWords& operator=(const DeviceWords&) = delete;
    void download(uint64_t* host) const { check(mkhe_buf_download(params.ctx, d, host, words)); }
    Parameters& params;
    size_t words;
    void* d = nullptr;
};
// mkrlwe.SecretKey keys.go:9-12 / PublicKey :15-18: PolyQP (NTT, Montgomery form) resp. two of them, on the device
struct SecretKey {
    SecretKey(Parameters& p, std::string id) : ID(std::move(id)), Value(p, (size_t)(p.QCount() + p.PCount()) * p.N()) {}
    std::string ID;
    DeviceWords Value;
};
struct PublicKey {
    PublicKey(Parameters& p, std::string id) : ID(std::move(id)), Value(p, 2 * (size_t)(p.QCount() + p.PCount()) * p.N()) {}
    std::string ID;
    DeviceWords Value;
};

// keys.go:34-37: Value = (b, d, v)
struct RelinearizationKey {
    RelinearizationKey(Parameters& p, std::string id, const uint64_t* b, const uint64_t* d, const uint64_t* v) : ID(std::move(id)) {
        Value[0] = std::make_shared<SwitchingKey>(p, b); Value[1] = std::make_shared<SwitchingKey>(p, d); Value[2] = std::make_shared<SwitchingKey>(p, v);
    }
    std::string ID;
    std::shared_ptr<SwitchingKey> Value[3];
};
// keys.go:53-57,165-198
struct RelinearizationKeySet {
    void AddRelinearizationKey(std::shared_ptr<RelinearizationKey> k) { Value[k->ID] = std::move(k); }
    void DelRelinearizationKey(const std::string& id) { Value.erase(id); }
    RelinearizationKey& GetRelinearizationKey(const std::string& id) {
        auto it = Value.find(id);
        if (it == Value.end()) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
        return *it->second;
    }
    std::map<std::string, std::shared_ptr<RelinearizationKey>> Value;
};
// keys.go:40-44,60-62,128-162
struct RotationKey {
    RotationKey(Parameters& p, int rotidx, std::string id, const uint64_t* v) : ID(std::move(id)), RotIdx(rotidx), Value(std::make_shared<SwitchingKey>(p, v)) {}
    std::string ID; int RotIdx; std::shared_ptr<SwitchingKey> Value;
};
struct RotationKeySet {
    void AddRotationKey(std::shared_ptr<RotationKey> k) { Value[k->ID][k->RotIdx] = std::move(k); }
    RotationKey& GetRotationKey(const std::string& id, int rotidx) {
        auto it = Value.find(id);
        if (it == Value.end() || !it->second.count(rotidx)) throw Error("cannot GetRotationKeys: there is no rotation key with given id");
        return *it->second[rotidx];
    }
    std::map<std::string, std::map<int, std::shared_ptr<RotationKey>>> Value;
};
// keys.go:47-50,65-67,200-228
struct ConjugationKey {
    ConjugationKey(Parameters& p, std::string id, const uint64_t* v) : ID(std::move(id)), Value(std::make_shared<SwitchingKey>(p, v)) {}
    std::string ID; std::shared_ptr<SwitchingKey> Value;
};
struct ConjugationKeySet {
    void AddConjugationKey(std::shared_ptr<ConjugationKey> k) { Value[k->ID] = std::move(k); }
    ConjugationKey& GetConjugationKey(const std::string& id) {
        auto it = Value.find(id);
        if (it == Value.end()) throw Error("cannot GetConjugationKey: there is no conjugation key with given id");
        return *it->second;
    }
    std::map<std::string, std::shared_ptr<ConjugationKey>> Value;
};
// elements.go:5-15
struct HoistedCiphertext { std::map<std::string, std::shared_ptr<SwitchingKey>> Value; };

// mkrlwe.Ciphertext (elements.go:17-33): Value["0"] + one poly per id; device layout uint64[1+n][level+1][N], ids sorted
class Ciphertext {
  public:
    Ciphertext(Parameters& p, const IDSet& idset, int level, bool zero = true) : params(p), ids(idset.begin(), idset.end()), level_(level) {
        std::vector<int> dense;
        for (auto& s : ids) dense.push_back(p.party_index(s));
        check((zero ? mkhe_ct_create : mkhe_ct_create_uninit)(p.ctx, (int)ids.size(), dense.data(), level + 1, &h));
    }
    virtual ~Ciphertext() { if (h) mkhe_ct_destroy(params.ctx, h); }
    Ciphertext(const Ciphertext&) = delete;
    Ciphertext& operator=(const Ciphertext&) = delete;
    IDSet IDSet_() const { return IDSet(ids.begin(), ids.end()); }
    int Level() const { return level_; }
    int slot(const std::string& id) const {
        if (id == "0") return 0;
        auto it = std::find(ids.begin(), ids.end(), id);
        if (it == ids.end()) throw Error("mkhe: ciphertext has no component of that id");
        return 1 + (int)(it - ids.begin());
    }
    size_t words() const { return (size_t)(1 + ids.size()) * (level_ + 1) * params.N(); }
    void upload(const uint64_t* host) { check(mkhe_ct_upload(params.ctx, h, host)); }
    void download(uint64_t* host) const { check(mkhe_ct_download(params.ctx, h, host)); }
    Parameters& params;
    std::vector<std::string> ids;
    mkhe_ct* h = nullptr;
  protected:
    int level_;
};

// mkrlwe.KeySwitcher (keyswitch.go:8-47); the reference's pools are engine-internal device buffers
class KeySwitcher {
  public:
    explicit KeySwitcher(Parameters& p) : params(p) {}
    void Decompose(int levelQ, const Ciphertext& ct, const std::string& id, SwitchingKey& ad, bool isNTT = false) {     // keyswitch.go:49-73
        check(mkhe_decompose(params.ctx, levelQ, isNTT ? 1 : 0, ct.h, ct.slot(id), ad.h));
    }
    void ExternalProduct(int levelQ, const Ciphertext& ct, const std::string& id, const SwitchingKey& bg, Ciphertext& out, const std::string& out_id, bool isNTT = false) {
        check(mkhe_external_product(params.ctx, levelQ, isNTT ? 1 : 0, ct.h, ct.slot(id), bg.h, out.h, out.slot(out_id)));      // keyswitch.go:79-118
    }
    void ExternalProductHoisted(int levelQ, const SwitchingKey& aHoisted, const SwitchingKey& bg, Ciphertext& out, const std::string& out_id) {
        check(mkhe_external_product_hoisted(params.ctx, levelQ, aHoisted.h, bg.h, out.h, out.slot(out_id)));                    // keyswitch_hoisted.go:10-40
    }
    void MulAndRelin(const Ciphertext& op0, const Ciphertext& op1, RelinearizationKeySet& rlkSet, Ciphertext& ctOut) {          // keyswitch.go:122-230
        MulAndRelinHoisted(op0, op1, nullptr, nullptr, rlkSet, ctOut);
    }
    // rescaled = true: ctOut is one level below the product and receives Rescale(MulAndRelin(..)) in one engine call (mkhe_mul_relin_rescale)
    void MulAndRelinHoisted(const Ciphertext& op0, const Ciphertext& op1, const HoistedCiphertext* op0Hoisted, const HoistedCiphertext* op1Hoisted,
                            RelinearizationKeySet& rlkSet, Ciphertext& ctOut, bool rescaled = false) {                          // keyswitch_hoisted.go:44-179
        const int lvl = ctOut.Level() + (rescaled ? 1 : 0);
        if (op0.Level() < lvl || op1.Level() < lvl) throw Error("Cannot MulAndRelin: op0 and op1 have different levels");
        if (!params.CRS.count(-1)) throw Error("mkhe: CRS[-1] (u) has not been uploaded");
        std::vector<const mkhe_swk*> d0, v0, b1, h0, h1;
        for (auto& i : op0.ids) { auto& k = rlkSet.GetRelinearizationKey(i); d0.push_back(k.Value[1]->h); v0.push_back(k.Value[2]->h); }
        for (auto& i : op1.ids) b1.push_back(rlkSet.GetRelinearizationKey(i).Value[0]->h);
        if (op0Hoisted) for (auto& i : op0.ids) h0.push_back(op0Hoisted->Value.at(i)->h);
        if (op1Hoisted) for (auto& i : op1.ids) h1.push_back(op1Hoisted->Value.at(i)->h);
        check((rescaled ? mkhe_mul_relin_rescale : mkhe_mul_and_relin)(params.ctx, op0.h, op1.h, op0Hoisted ? h0.data() : nullptr, op1Hoisted ? h1.data() : nullptr,
                                 b1.data(), d0.data(), v0.data(), params.CRS[-1]->h, ctOut.h));
    }
    void Rotate(const Ciphertext& ctIn, int rotidx, RotationKeySet& rkSet, Ciphertext& ctOut) { RotateHoisted(ctIn, rotidx, nullptr, rkSet, ctOut); }   // keyswitch.go:234-298
    void RotateHoisted(const Ciphertext& ctIn, int rotidx, const HoistedCiphertext* ctInHoisted, RotationKeySet& rkSet, Ciphertext& ctOut) {        // keyswitch_hoisted.go:183-247
        if (ctIn.Level() < ctOut.Level()) throw Error("Cannot Rotate: ctIn and ctOut have different levels");
        const int n2 = params.N() / 2;
        while (rotidx < 0) rotidx += n2;                                                                                                             // keyswitch.go:246-249
        if (!params.CRS.count(rotidx)) throw Error("mkhe: no CRS for this rotation index");
        std::vector<const mkhe_swk*> rk, hs;
        for (auto& i : ctIn.ids) rk.push_back(rkSet.GetRotationKey(i, rotidx).Value->h);
        if (ctInHoisted) for (auto& i : ctIn.ids) hs.push_back(ctInHoisted->Value.at(i)->h);
        check(mkhe_rotate(params.ctx, params.GaloisElementForColumnRotationBy(rotidx), ctIn.h, ctInHoisted ? hs.data() : nullptr, rk.data(),
                          params.CRS[rotidx]->h, ctOut.h));
    }
    void Conjugate(const Ciphertext& ctIn, ConjugationKeySet& ckSet, Ciphertext& ctOut) {                                                           // keyswitch.go:302-332
        if (ctIn.Level() < ctOut.Level()) throw Error("Cannot Conjugate: ctIn and ctOut have different levels");
        std::vector<const mkhe_swk*> ck;
        for (auto& i : ctIn.ids) ck.push_back(ckSet.GetConjugationKey(i).Value->h);
        check(mkhe_conjugate(params.ctx, params.GaloisElementForRowRotation(), ctIn.h, ck.data(), params.CRS.at(-2)->h, ctOut.h));
    }
    Parameters& params;
};
inline IDSet Union(const IDSet& a, const IDSet& b) { IDSet r = a; r.insert(b.begin(), b.end()); return r; }

// mkrlwe.KeyGenerator (keygen.go:13-40) on the device.  The small-norm SAMPLES (what lattigo's ternary / Gaussian samplers
// draw) are arguments: int32, N per polynomial, from the caller's CSPRNG -- secret randomness never comes from the GPU.
class KeyGenerator {
  public:
    explicit KeyGenerator(Parameters& p) : params(p) {}
    std::unique_ptr<SecretKey> GenSecretKey(const std::string& id, const int32_t* s) {                                   // keygen.go:44-76
        auto sk = std::make_unique<SecretKey>(params, id);
        check(mkhe_keygen_secret(params.ctx, s, sk->Value.d));
        return sk;
    }
    std::unique_ptr<PublicKey> GenPublicKey(const SecretKey& sk, const int32_t* e) {                                     // keygen.go:88-109
        auto pk = std::make_unique<PublicKey>(params, sk.ID);
        check(mkhe_keygen_public_key(params.ctx, sk.Value.d, e, crs(0)->h, pk->Value.d));
        return pk;
    }
    void GenSwitchingKey(const SecretKey& skIn, SwitchingKey& swk, const int32_t* e) {                                   // keygen.go:270-327
        check(mkhe_keygen_switching_key(params.ctx, skIn.Value.d, e, swk.h));
    }
    std::shared_ptr<RelinearizationKey> GenRelinearizationKey(const SecretKey& sk, const SecretKey& r, const int32_t* e) {   // keygen.go:137-187
        if (params.PCount() == 0) throw Error("modulus P is empty");
        auto rlk = std::make_shared<RelinearizationKey>(params, sk.ID, nullptr, nullptr, nullptr);
        check(mkhe_keygen_relin_key(params.ctx, sk.Value.d, r.Value.d, e, crs(0)->h, crs(-1)->h, rlk->Value[0]->h, rlk->Value[1]->h, rlk->Value[2]->h));
        return rlk;
    }
    std::shared_ptr<RotationKey> GenRotationKey(int rotidx, const SecretKey& sk, const int32_t* e) {                     // keygen.go:190-229
        if (!params.CRS.count(rotidx)) throw Error("Cannot GenRotationKey: CRS for given rot idx is not generated");
        auto a = params.CRS.at(rotidx);
        while (rotidx < 0) rotidx += params.N() / 2;
        auto rk = std::make_shared<RotationKey>(params, rotidx, sk.ID, nullptr);
        check(mkhe_keygen_rotation_key(params.ctx, params.GaloisElementForColumnRotationBy(rotidx), sk.Value.d, e, a->h, rk->Value->h));
        return rk;
    }
    std::shared_ptr<ConjugationKey> GenConjugationKey(const SecretKey& sk, const int32_t* e) {                           // keygen.go:240-268
        auto ck = std::make_shared<ConjugationKey>(params, sk.ID, nullptr);
        check(mkhe_keygen_conjugation_key(params.ctx, sk.Value.d, e, crs(-2)->h, ck->Value->h));
        return ck;
    }
    Parameters& params;

  protected:
    std::shared_ptr<SwitchingKey> crs(int idx) {
        auto it = params.CRS.find(idx);
        if (it == params.CRS.end()) throw Error("mkhe: CRS[" + std::to_string(idx) + "] is not generated");
        return it->second;
    }
};
}  // namespace mkrlwe

namespace mkckks {
using mkhe::check;
using mkhe::Error;

// mkckks.Parameters (params.go:11-24): mkrlwe parameters with gamma = 2 + default scale
class Parameters : public mkrlwe::Parameters {
  public:
    Parameters(int logN, std::vector<uint64_t> Q, std::vector<uint64_t> P, double scale, int device = 0)
        : mkrlwe::Parameters(logN, std::move(Q), std::move(P), 2, device), scale_(scale) {}
    double Scale() const { return scale_; }
  private:
    double scale_;
};
// mkckks.Ciphertext (elements.go:5-17)
class Ciphertext : public mkrlwe::Ciphertext {
  public:
    Ciphertext(Parameters& p, const mkrlwe::IDSet& idset, int level, double scale, bool zero = true) : mkrlwe::Ciphertext(p, idset, level, zero), Scale(scale) {}
    double ScalingFactor() const { return Scale; }
    double Scale;
};
typedef std::unique_ptr<Ciphertext> CiphertextPtr;

// mkckks.Evaluator (evaluator.go:13-39)
class Evaluator {
  public:
    explicit Evaluator(Parameters& p) : params(p), ksw(p) {}
    CiphertextPtr AddNew(const Ciphertext& op0, const Ciphertext& op1) { return binary(op0, op1, mkhe_ct_add); }      // evaluator.go:316-327
    CiphertextPtr SubNew(const Ciphertext& op0, const Ciphertext& op1) { return binary(op0, op1, mkhe_ct_sub); }      // evaluator.go:329-357
    // evaluator.go:376-384
    int nbRescales(const Ciphertext& ctIn, double minScale, double* scaleOut) const {
        double scale = ctIn.Scale; int nb = 0;
        while (ctIn.Level() - nb >= 0 && scale / (double)params.Q()[ctIn.Level() - nb] >= minScale / 2) { scale /= (double)params.Q()[ctIn.Level() - nb]; ++nb; }
        *scaleOut = scale;
        return nb;
    }
    // evaluator.go:96-114: keep the first Level()+1-levels limbs of every component
    CiphertextPtr DropLevelNew(const Ciphertext& ct0, int levels) {
        auto out = std::make_unique<Ciphertext>(params, ct0.IDSet_(), ct0.Level() - levels, ct0.Scale, false);
        std::vector<uint64_t> one;
        for (int i = 0; i <= out->Level(); ++i) one.push_back((uint64_t)((((unsigned __int128)1) << 64) % params.Q()[i]));     // MForm(1)
        check(mkhe_ct_mul_const(params.ctx, ct0.h, one.data(), one.data(), out->h));
        return out;
    }
    // evaluator.go:465-481: dev_pt = the plaintext polynomial uint64[Level()+1][N] resident on the device (mkhe_buf_*), pt_scale its scale
    CiphertextPtr MulPtxtNew(const Ciphertext& ct, const void* dev_pt, double pt_scale) {
        auto out = std::make_unique<Ciphertext>(params, ct.IDSet_(), ct.Level(), ct.Scale * pt_scale, false);
        check(mkhe_ct_mul_ptxt(params.ctx, ct.h, dev_pt, out->h));
        if (out->Level() == 0) return out;
        double scale; const int nb = nbRescales(*out, params.Scale(), &scale);
        if (nb == 0) return out;
        auto res = std::make_unique<Ciphertext>(params, out->IDSet_(), out->Level() - nb, scale, false);
        check(mkhe_rescale(params.ctx, out->h, nb, res->h));
        return res;
    }
    CiphertextPtr RescaleNew(const Ciphertext& ct0, double threshold) {                                                 // evaluator.go:359-414
        if (threshold <= 0) throw Error("cannot Rescale: minScale is 0");
        if (ct0.Scale == 0) throw Error("cannot Rescale: ciphertext scale is 0");
        if (ct0.Level() == 0) throw Error("cannot Rescale: input Ciphertext already at level 0");
        double scale; const int nb = nbRescales(ct0, threshold, &scale);
        auto out = std::make_unique<Ciphertext>(params, ct0.IDSet_(), ct0.Level() - nb, scale, false);
        check(mkhe_rescale(params.ctx, ct0.h, nb, out->h));
        return out;
    }
    std::unique_ptr<mkrlwe::HoistedCiphertext> HoistedForm(const Ciphertext& ct) {                                      // evaluator.go:543-553
        auto h = std::make_unique<mkrlwe::HoistedCiphertext>();
        std::vector<mkhe_swk*> outs;
        for (auto& id : ct.ids) { h->Value[id] = mkrlwe::NewSwitchingKey(params); outs.push_back(h->Value[id]->h); }
        check(mkhe_hoisted_form(params.ctx, ct.Level(), ct.h, outs.data()));          // all party components in one batched launch
        return h;
    }
    CiphertextPtr MulRelinNew(const Ciphertext& op0, const Ciphertext& op1, mkrlwe::RelinearizationKeySet& rlkSet) {      // evaluator.go:416-443
        return MulRelinHoistedNew(op0, op1, nullptr, nullptr, rlkSet);
    }
    CiphertextPtr MulRelinHoistedNew(const Ciphertext& op0, const Ciphertext& op1, const mkrlwe::HoistedCiphertext* h0, const mkrlwe::HoistedCiphertext* h1,
                                     mkrlwe::RelinearizationKeySet& rlkSet) {                                            // evaluator.go:558-581
        {   // the usual single Rescale rides on the engine call (the number of rescales depends on scales and moduli only)
            const int level = std::min(op0.Level(), op1.Level());
            double sc = op0.ScalingFactor() * op1.ScalingFactor(); int nb1 = 0;
            while (level - nb1 >= 0 && sc / (double)params.Q()[level - nb1] >= params.Scale() / 2) { sc /= (double)params.Q()[level - nb1]; ++nb1; }
            if (nb1 == 1 && level >= 1) {
                auto res = std::make_unique<Ciphertext>(params, mkrlwe::Union(op0.IDSet_(), op1.IDSet_()), level - 1, sc, false);
                ksw.MulAndRelinHoisted(op0, op1, h0, h1, rlkSet, *res, true);
                return res;
            }
        }
        auto ctOut = std::make_unique<Ciphertext>(params, mkrlwe::Union(op0.IDSet_(), op1.IDSet_()), std::min(op0.Level(), op1.Level()),
                                                  op0.ScalingFactor() * op1.ScalingFactor(), false);
        ksw.MulAndRelinHoisted(op0, op1, h0, h1, rlkSet, *ctOut);
        double scale; const int nb = nbRescales(*ctOut, params.Scale(), &scale);
        if (nb == 0 || ctOut->Level() == 0) return ctOut;
        auto res = std::make_unique<Ciphertext>(params, ctOut->IDSet_(), ctOut->Level() - nb, scale, false);
        check(mkhe_rescale(params.ctx, ctOut->h, nb, res->h));
        return res;
    }
    CiphertextPtr RotateNew(const Ciphertext& ct0, int rotidx, mkrlwe::RotationKeySet& rkSet) {                          // evaluator.go:485-525
        const int n2 = params.N() / 2;
        rotidx = ((rotidx % n2) + n2) % n2;
        if (rotidx == 0) return copy(ct0);
        if (params.CRS.count(rotidx)) { auto out = like(ct0); ksw.Rotate(ct0, rotidx, rkSet, *out); return out; }
        CiphertextPtr cur; const Ciphertext* src = &ct0;
        for (int k = 1; rotidx > 0; k *= 2, rotidx /= 2) {                                                                // power-of-two decomposition, :516-523
            if (rotidx % 2 == 0) continue;
            auto nxt = like(ct0);
            ksw.Rotate(*src, k, rkSet, *nxt);
            cur = std::move(nxt); src = cur.get();
        }
        return cur;
    }
    CiphertextPtr RotateHoistedNew(const Ciphertext& ct0, int rotidx, const mkrlwe::HoistedCiphertext& hoisted, mkrlwe::RotationKeySet& rkSet) {   // evaluator.go:585-617
        const int n2 = params.N() / 2;
        rotidx = ((rotidx % n2) + n2) % n2;
        if (rotidx == 0) return copy(ct0);
        if (!params.CRS.count(rotidx)) throw Error("Hoisted rotation only works for precomputed rotation keys");
        auto out = like(ct0);
        ksw.RotateHoisted(ct0, rotidx, &hoisted, rkSet, *out);
        return out;
    }
    CiphertextPtr ConjugateNew(const Ciphertext& ct0, mkrlwe::ConjugationKeySet& ckSet) {                                // evaluator.go:527-541
        auto out = like(ct0);
        ksw.Conjugate(ct0, ckSet, *out);
        return out;
    }
    Parameters& params;
    mkrlwe::KeySwitcher ksw;

  private:
    CiphertextPtr like(const Ciphertext& c) { return std::make_unique<Ciphertext>(params, c.IDSet_(), c.Level(), c.Scale, false); }
    CiphertextPtr copy(const Ciphertext& c) {
        auto out = like(c);
        check(mkhe_ct_copy(params.ctx, c.h, out->h));
        return out;
    }
    // the operand times an integer factor (MultByConst with an integer-valued float64: constant scale 1, evaluator.go:117-199), scale unchanged
    CiphertextPtr timesInteger(const Ciphertext& op, double f) {
        if (!(f < 9223372036854775808.0)) throw Error("mkhe: scale ratio beyond 2^63");
        const uint64_t fi = (uint64_t)f;
        std::vector<uint64_t> c;
        for (int i = 0; i <= op.Level(); ++i) { const uint64_t q = params.Q()[i]; c.push_back((uint64_t)((((unsigned __int128)(fi % q)) << 64) % q)); }     // MForm(f mod q_i)
        auto tmp = like(op);
        check(mkhe_ct_mul_const(params.ctx, op.h, c.data(), c.data(), tmp->h));
        return tmp;
    }
    // AddNew / SubNew with the scale matching of evaluateInPlace (evaluator.go:214-281, the fresh-ctOut branch): the operand with the
    // smaller scale is first multiplied by floor(ratio) when that is > 1; the result carries max(s0, s1) like the reference's
    template <typename F> CiphertextPtr binary(const Ciphertext& op0, const Ciphertext& op1, F fn) {
        const double s0 = op0.Scale, s1 = op1.Scale;
        CiphertextPtr t0, t1;
        if (s1 > s0 && std::floor(s1 / s0) > 1) t0 = timesInteger(op0, std::floor(s1 / s0));
        else if (s0 > s1 && std::floor(s0 / s1) > 1) t1 = timesInteger(op1, std::floor(s0 / s1));
        const Ciphertext& a = t0 ? *t0 : op0;
        const Ciphertext& b = t1 ? *t1 : op1;
        auto out = std::make_unique<Ciphertext>(params, mkrlwe::Union(op0.IDSet_(), op1.IDSet_()), std::min(op0.Level(), op1.Level()), std::max(s0, s1), false);
        check(fn(params.ctx, a.h, b.h, out->h));
        return out;
    }
};
}  // namespace mkckks

namespace mkbfv {
using mkhe::check;
using mkhe::Error;

// mkbfv.Parameters (params.go:21-76): rings Q, QMul, R = Q || QMul, P, plaintext modulus T; gamma = 2
class Parameters : public mkrlwe::Parameters {
  public:
    Parameters(int logN, std::vector<uint64_t> Q, std::vector<uint64_t> QMul, std::vector<uint64_t> P, uint64_t T, int device = 0)
        : mkrlwe::Parameters(logN, std::move(Q), std::move(P), 2, true), QMul_(std::move(QMul)), T_(T) {
        if (Q_.size() != QMul_.size()) throw Error("cannot NewParametersFromLiteral: length of Q & QMul is not equal");            // params.go:30-32
        check(mkhe_ctx_create_bfv(&ctx, logN, Q_.data(), QMul_.data(), (int)Q_.size(), P_.data(), (int)P_.size(), 2, T, device));
    }
    uint64_t T() const { return T_; }
  private:
    std::vector<uint64_t> QMul_;
    uint64_t T_;
};
// mkbfv.Ciphertext (elements.go:5-11): always at MaxLevel, coefficient domain
class Ciphertext : public mkrlwe::Ciphertext {
  public:
    Ciphertext(Parameters& p, const mkrlwe::IDSet& idset, bool zero = true) : mkrlwe::Ciphertext(p, idset, p.MaxLevel(), zero) {}
};
typedef std::unique_ptr<Ciphertext> CiphertextPtr;
// mkbfv.RelinearizationKey (keys.go:6-9): Value[0] = (b1, d1, v), Value[1] = (b2, d2, -)
struct RelinearizationKey {
    RelinearizationKey(Parameters& p, std::string id, const uint64_t* b1, const uint64_t* b2, const uint64_t* d1, const uint64_t* d2, const uint64_t* v) : ID(id) {
        Value[0] = std::make_shared<mkrlwe::RelinearizationKey>(p, id, b1, d1, v);
        Value[1] = std::make_shared<mkrlwe::RelinearizationKey>(p, id, b2, d2, nullptr);
    }
    std::string ID;
    std::shared_ptr<mkrlwe::RelinearizationKey> Value[2];
};
struct RelinearizationKeySet {                                                                                                     // keys.go:11-21,33-82
    void AddRelinearizationKey(std::shared_ptr<RelinearizationKey> k) { Value[k->ID] = std::move(k); }
    RelinearizationKey& GetRelinearizationKey(const std::string& id) {
        auto it = Value.find(id);
        if (it == Value.end()) throw Error("cannot GetRelinearizationKey: there is no relinearization key with given id");
        return *it->second;
    }
    std::map<std::string, std::shared_ptr<RelinearizationKey>> Value;
};
// mkbfv.Evaluator (evaluator.go:7-20)
class Evaluator {
  public:
    explicit Evaluator(Parameters& p) : params(p), ksw(p) {}
    CiphertextPtr AddNew(const Ciphertext& op0, const Ciphertext& op1) { auto o = bin(op0, op1); check(mkhe_ct_add(params.ctx, op0.h, op1.h, o->h)); return o; }   // :44-52
    CiphertextPtr SubNew(const Ciphertext& op0, const Ciphertext& op1) { auto o = bin(op0, op1); check(mkhe_ct_sub(params.ctx, op0.h, op1.h, o->h)); return o; }   // :54-76
    // evaluator.go:95-113 -> KeySwitcher.MulAndRelinBFV (keyswitch.go:115-251): the non-hoisted twin on its own device path
    CiphertextPtr mulRelin(const Ciphertext& op0, const Ciphertext& op1, RelinearizationKeySet& rlkSet) { return MulRelinNew(op0, op1, rlkSet, false); }
    CiphertextPtr MulRelinNew(const Ciphertext& op0, const Ciphertext& op1, RelinearizationKeySet& rlkSet, bool hoisted = true) {  // evaluator.go:78-82,118-140
        if (!params.CRS.count(-1)) throw Error("mkhe: CRS[-1] (u) has not been uploaded");
        auto out = bin(op0, op1);
        std::vector<const mkhe_swk*> b1, b2, d1, d2, v;
        for (auto& i : op1.ids) { auto& k = rlkSet.GetRelinearizationKey(i); b1.push_back(k.Value[0]->Value[0]->h); b2.push_back(k.Value[1]->Value[0]->h); }
        for (auto& i : op0.ids) {
            auto& k = rlkSet.GetRelinearizationKey(i);
            d1.push_back(k.Value[0]->Value[1]->h); d2.push_back(k.Value[1]->Value[1]->h); v.push_back(k.Value[0]->Value[2]->h);
        }
        check((hoisted ? mkhe_bfv_mul_relin : mkhe_bfv_mul_relin_unhoisted)(params.ctx, op0.h, op1.h, b1.data(), b2.data(), d1.data(), d2.data(), v.data(), params.CRS[-1]->h, out->h));
        return out;
    }
    CiphertextPtr RotateNew(const Ciphertext& ct0, int rotidx, mkrlwe::RotationKeySet& rkSet) {                                    // evaluator.go:142-180 (precomputed indices)
        auto out = std::make_unique<Ciphertext>(params, ct0.IDSet_(), false);
        ksw.Rotate(ct0, rotidx, rkSet, *out);
        return out;
    }
    CiphertextPtr ConjugateNew(const Ciphertext& ct0, mkrlwe::ConjugationKeySet& ckSet) {                                          // evaluator.go:182-192
        auto out = std::make_unique<Ciphertext>(params, ct0.IDSet_(), false);
        ksw.Conjugate(ct0, ckSet, *out);
        return out;
    }
    Parameters& params;
    mkrlwe::KeySwitcher ksw;
  private:
    CiphertextPtr bin(const Ciphertext& a, const Ciphertext& b) { return std::make_unique<Ciphertext>(params, mkrlwe::Union(a.IDSet_(), b.IDSet_()), false); }
};
}  // namespace mkbfv
