// keygen.hip -- Context methods for key generation and CRS expansion.
//
// Every key of the reference is "NTT of small error polynomials, then one elementwise combination with secrets / CRS":
//   GenSwitchingKey     swk_i = MForm(NTT(e_i)) + P*sk on the Q limbs of digit i            mkrlwe/keygen.go:270-327
//   rlk.b_i             MForm(NTT(e_i) - InvMForm(a_i*sk))  =  MForm(NTT(e_i)) - a_i*sk      :163-169
//   rlk.d_i             swk(sk)_i - a_i*r                                                    :172-176
//   rlk.v_i             -(swk(r)_i + u_i*sk)                                                 :179-185
//   rk_i / ck_i         swk(.)_i - a_i*sigma(sk)  (sigma = NTT-domain index permutation)     :190-268
//   BFV swk_i           MForm(InvMForm(sk)*G_i + NTT(e_i))  =  MForm(NTT(e_i)) + sk*MForm(G_i)   mkbfv/keygen.go:91-162
// so a key is: upload beta*N int32 samples, small_expand into the output key, ONE batched forward NTT in place
// (beta*(nq+np) limbs), ONE keygen_combine pass.  Every stored value is the canonical representative the reference's
// CRed / MRed sequence produces (Neg writes q - x, 0 -> q, like lattigo).
#include "engine.h"

namespace mkhe {

typedef unsigned __int128 u128;
static u64 kg_mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
static u64 kg_to_mont(u64 a, u64 q) { return (u64)(((u128)(a % q) << 64) % q); }

void Context::kg_init() {
    if (kg_ready_) return;
    MKHE_HIP(hipMalloc(&kg_small_, (size_t)beta_max * N * sizeof(int32_t)));
    MKHE_HIP(hipMalloc(&kg_g_, 2 * (size_t)beta_max * mtot * sizeof(u64)));
    MKHE_HIP(hipMalloc(&kg_sk_, (size_t)mtot * N * sizeof(u64)));
    // mkrlwe gadget: MForm(P mod q_j) on the Q limbs [i*alpha, i*alpha + alpha) of digit i, nothing elsewhere
    std::vector<u64> g((size_t)beta_max * mtot, 0);
    for (int i = 0; i < beta_max; ++i)
        for (int j = i * alpha; j < std::min((i + 1) * alpha, nq); ++j) {
            u64 pm = 1;
            for (int k = 0; k < np; ++k) pm = kg_mulmod(pm, moduli[nq + k] % moduli[j], moduli[j]);
            g[(size_t)i * mtot + j] = kg_to_mont(pm, moduli[j]);
        }
    MKHE_HIP(hipMemcpy(kg_g_, g.data(), g.size() * sizeof(u64), hipMemcpyHostToDevice));
    kg_ready_ = true;
}

// the uploaded small-norm samples (secret / error coefficients) are consumed by small_expand_kernel on the same stream: zero the
// scratch right behind it, so that secret material does not linger in a reusable device buffer
void Context::wipe_samples(size_t count) {
    MKHE_HIP(hipMemsetAsync(kg_small_, 0, count * sizeof(int32_t), s_));
}

void Context::kg_upload_g(const u64* g_plain) {
    kg_init();
    std::vector<u64> g((size_t)beta_max * mtot);
    for (int i = 0; i < beta_max; ++i)
        for (int j = 0; j < mtot; ++j) g[(size_t)i * mtot + j] = kg_to_mont(g_plain[(size_t)i * mtot + j], moduli[j]);
    MKHE_HIP(hipMemcpyAsync(kg_g_ + (size_t)beta_max * mtot, g.data(), g.size() * sizeof(u64), hipMemcpyHostToDevice, stream));
    sync();
}

void Context::kg_key(const int32_t* e, int gadget, const u64* skA, const u64* crs, const u64* skB, int sign, bool neg, u64* out) {
    kg_init();
    MKHE_HIP(hipMemcpyAsync(kg_small_, e, (size_t)beta_max * N * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    sync();                                             // the host array may be pageable: do not return before it is consumed
    {
        ProfScope ps(this, PROF_OTHER, (double)beta_max * N * (4.0 + 8.0 * mtot));
        launch_small_expand(out, kg_small_, d_mods, beta_max, mtot, N, s_);
    }
    wipe_samples((size_t)beta_max * N);
    ntt(out, out, beta_max, mtot, 0, false, false);
    KeygenArgs a{};
    a.out = out; a.skA = skA; a.g = gadget ? kg_g_ + (gadget == 2 ? (size_t)beta_max * mtot : 0) : nullptr;
    a.crs = crs; a.skB = skB; a.mods = d_mods; a.beta = beta_max; a.mtot = mtot; a.N = N;
    a.sign = sign; a.neg = neg ? 1 : 0; a.mform_e = 1;
    {
        ProfScope ps(this, PROF_OTHER, 8.0 * N * beta_max * mtot * (2.0 + (crs ? 1.0 : 0.0)));
        launch_keygen_combine(a, s_);
    }
    MKHE_HIP(hipGetLastError());
}

void Context::keygen_secret(const int32_t* s, u64* dev_sk) {
    kg_init();
    MKHE_HIP(hipMemcpyAsync(kg_small_, s, (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    sync();
    launch_small_expand(dev_sk, kg_small_, d_mods, 1, mtot, N, s_);
    wipe_samples((size_t)N);
    ntt(dev_sk, dev_sk, 1, mtot, 0, false, false);
    launch_mform(dev_sk, dev_sk, d_mods, d_map_id, mtot, N, s_);
    MKHE_HIP(hipGetLastError());
}

void Context::keygen_switching_key(const u64* sk, const int32_t* e, u64* out) { kg_key(e, 1, sk, nullptr, nullptr, 0, false, out); }

void Context::keygen_public_key(const u64* sk, const int32_t* e, const u64* crs_a, u64* pk) {
    kg_init();
    const size_t pw = (size_t)mtot * N;
    MKHE_HIP(hipMemcpyAsync(kg_small_, e, (size_t)N * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    sync();
    launch_small_expand(pk, kg_small_, d_mods, 1, mtot, N, s_);
    wipe_samples((size_t)N);
    ntt(pk, pk, 1, mtot, 0, false, false);
    MKHE_HIP(hipMemcpyAsync(pk + pw, crs_a, pw * sizeof(u64), hipMemcpyDeviceToDevice, s_));      // pk[1] = CRS[0].Value[0]
    KeygenArgs a{};
    a.out = pk; a.crs = crs_a; a.skB = sk; a.mods = d_mods; a.beta = 1; a.mtot = mtot; a.N = N; a.sign = -1; a.mform_e = 0;
    launch_keygen_combine(a, s_);                                                                  // pk[0] = NTT(e) - sk*a
    MKHE_HIP(hipGetLastError());
}

void Context::keygen_relin_key(const u64* sk, const u64* r, const int32_t* e, const u64* crs_a, const u64* crs_u,
                               u64* b, u64* d, u64* v) {
    const size_t en = (size_t)beta_max * N;
    kg_key(e, 0, nullptr, crs_a, sk, -1, false, b);
    kg_key(e + en, 1, sk, crs_a, r, -1, false, d);
    kg_key(e + 2 * en, 1, r, crs_u, sk, +1, true, v);
}

void Context::keygen_rotation_key(u64 galEl, const u64* sk, const int32_t* e, const u64* crs, u64* out) {
    kg_init();
    const u64 n2 = 2 * (u64)N;
    if (!(galEl & 1) || galEl >= n2) throw Error("mkhe: galois element must be odd and < 2N");
    u64 inv = 1, bs = galEl;                                     // InverseGaloisElement: galEl^(2N-1) mod 2N
    for (u64 ex = n2 - 1; ex; ex >>= 1) { if (ex & 1) inv = inv * bs % n2; bs = bs * bs % n2; }
    launch_permute_ntt(kg_sk_, sk, mtot, logN, inv, s_);
    kg_key(e, 1, sk, crs, kg_sk_, -1, false, out);
    MKHE_HIP(hipMemsetAsync(kg_sk_, 0, (size_t)mtot * N * sizeof(u64), s_));       // the permuted secret does not outlive the call
}

void Context::keygen_conjugation_key(const u64* sk, const int32_t* e, const u64* crs, u64* out) {
    kg_init();
    launch_permute_ntt(kg_sk_, sk, mtot, logN, 2 * (u64)N - 1, s_);
    kg_key(e, 1, kg_sk_, crs, sk, -1, false, out);
    MKHE_HIP(hipMemsetAsync(kg_sk_, 0, (size_t)mtot * N * sizeof(u64), s_));
}

void Context::bfv_keygen_switching_key(const u64* sk, const u64* g, const int32_t* e, u64* out) {
    kg_upload_g(g);
    kg_key(e, 2, sk, nullptr, nullptr, 0, false, out);
}

void Context::bfv_keygen_relin_key(const u64* sk, const u64* r, const u64* g1, const u64* g2, const int32_t* e,
                                   const u64* a1, const u64* a2, const u64* u, u64* b1, u64* b2, u64* d1, u64* d2, u64* v) {
    const size_t en = (size_t)beta_max * N;
    kg_key(e, 0, nullptr, a1, sk, -1, false, b1);
    kg_key(e + en, 0, nullptr, a2, sk, -1, false, b2);
    kg_upload_g(g1);
    kg_key(e + 2 * en, 2, sk, a1, r, -1, false, d1);
    kg_upload_g(g2);                                             // syncs: d1's combine pass has consumed g1
    kg_key(e + 3 * en, 2, sk, a2, r, -1, false, d2);
    kg_key(e + 4 * en, 1, r, u, sk, +1, true, v);
}

void Context::crs_expand(u64 seed, int32_t idx, u64* out) {
    ProfScope ps(this, PROF_OTHER, 8.0 * N * beta_max * mtot);
    launch_crs_expand(out, d_mods, seed, idx, beta_max, mtot, N, s_);
    MKHE_HIP(hipGetLastError());
}

}  // namespace mkhe
