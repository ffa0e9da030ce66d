// engine.hip -- Context: host-side table generation, buffer pools, stream plumbing, the NTT launch choice, Decompose and the external-product batch
// (engine_mulrelin.hip, engine_ops.hip, engine_bfv.hip, batch.hip hold the operations built on them).
#include "engine.h"
#include <atomic>
#include <mutex>
#include <algorithm>
#include <cstring>
#include <cstdlib>

namespace mkhe {

int ab_fuse_x() { static const int v = MKHE_AB_INT("MKHE_FUSE_X", 1); return v; }
int ab_fuse_y() { static const int v = MKHE_AB_INT("MKHE_FUSE_Y", 1); return v; }
int ab_fuse_e() { static const int v = MKHE_AB_INT("MKHE_FUSE_E", 1); return v; }

typedef unsigned __int128 u128;

// ------------------------------------------------------------------ host number theory
static u64 mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
static u64 powmod(u64 x, u64 e, u64 q) {
    u64 r = 1 % q; x %= q;
    for (; e; e >>= 1) { if (e & 1) r = mulmod(r, x, q); x = mulmod(x, x, q); }
    return r;
}
static u64 inv64(u64 q) { u64 x = q; for (int i = 0; i < 6; ++i) x *= 2 - q * x; return x; }   // q^-1 mod 2^64 (Newton)
static u64 to_mont(u64 a, u64 q) { return (u64)(((u128)a << 64) % q); }
static u64 bitrev(u64 x, int bits) { u64 r = 0; for (int i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

static bool is_prime(u64 n) {
    if (n < 2) return false;
    for (u64 p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) if (n % p == 0) return n == p;
    u64 d = n - 1; int s = 0;
    while ((d & 1) == 0) { d >>= 1; ++s; }
    for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
        u64 x = powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int i = 1; i < s; ++i) { x = mulmod(x, x, n); if (x == n - 1) { comp = false; break; } }
        if (comp) return false;
    }
    return true;
}

// Same choice of 2N-th root as lattigo v2.3.0 NewRing (ring/primes.go primitiveRoot: the scan
// starts at g = 3), so that key material produced by the Go side (NTT domain) lines up when the
// caller does not pass its own roots.
static u64 default_psi(u64 q, u64 N) {
    std::vector<u64> f;
    u64 n = q - 1;
    for (u64 p = 2; p * p <= n; p += (p == 2 ? 1 : 2))
        if (n % p == 0) { f.push_back(p); while (n % p == 0) n /= p; }
    if (n > 1) f.push_back(n);
    u64 g = 2;
    for (;;) {
        ++g;
        bool ok = true;
        for (u64 p : f) if (powmod(g, (q - 1) / p, q) == 1) { ok = false; break; }
        if (ok) break;
    }
    return powmod(g, (q - 1) / (2 * N), q);
}

template <typename T> static T* dev_upload(const std::vector<T>& v) {
    T* d = nullptr;
    MKHE_HIP(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
    if (!v.empty()) MKHE_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
// Device allocation of the engine's buffers.  The stream-ordered pools (Context::pool_free) keep freed handles' memory for reuse; when the
// driver runs out, what the pools of this device hold is handed back (one device-wide synchronisation) and the allocation is tried once more,
// so that a process never fails with most of HBM sitting in free lists.
static size_t trim_device_pools();
static u64* dev_alloc_words(size_t w) {
    u64* d = nullptr;
    const size_t bytes = std::max<size_t>(w, 1) * sizeof(u64);
    hipError_t e = hipMalloc(&d, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        if (trim_device_pools() > 0) e = hipMalloc(&d, bytes);
    }
    if (e != hipSuccess) throw Error(std::string("hipMalloc(") + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
    return d;
}

// ------------------------------------------------------------------ construction
Context::Context(int logN_, const u64* Q, int nq_, const u64* P, int np_, int gamma_,
                 const u64* psiQ, const u64* psiP, int device_, const u64* QMul, int nqm_, u64 T)
    : logN(logN_), N(1 << logN_), nq(nq_), np(np_), mtot(nq_ + np_), gamma(gamma_), device(device_) {
    // The context enters the per-device registry (the pools of the other contexts walk it) only once it is complete: a constructor that
    // throws half way -- a bad root, an unsupported alpha, a failed allocation -- never runs the destructor, so everything built so
    // far is released here and no dangling pointer is ever visible to pool_free / pool_alloc of another context.
    try { init(Q, P, psiQ, psiP, QMul, nqm_, T); }
    catch (...) { release_all(); throw; }
    registry_add();
}
void Context::init(const u64* Q, const u64* P, const u64* psiQ, const u64* psiP, const u64* QMul, int nqm_, u64 T) {
    nqm = QMul ? nqm_ : 0; mall = mtot + nqm; bfv_t = T;
    if (logN < 10 || logN > 16) throw Error("mkhe: logN must be in [10,16]");
    if (nq < 1 || np < 1 || gamma < 1 || np / gamma < 1) throw Error("mkhe: need at least gamma special primes (PCount/gamma >= 1)");
    if (np > MAXP) throw Error("mkhe: too many special primes");
    alpha = np / gamma;                                   // mkrlwe/params.go:63-65
    beta_max = (nq + alpha - 1) / alpha;                  // params.go:67-71
    if (2 * beta_max > MAX_TERMS) throw Error("mkhe: too many gadget digits");
    for (int i = 0; i < nq; ++i) moduli.push_back(Q[i]);
    for (int i = 0; i < np; ++i) moduli.push_back(P[i]);
    for (int i = 0; i < nqm; ++i) moduli.push_back(QMul[i]);
    if (nqm) {
        // mkbfv/params.go:30-34 (len(Q) == len(QMul)); alpha = 1 only: DecomposeBFV asks the R-ring decomposer
        // for one-prime digits (keyswitch.go:73-76), and both reference parameter sets have alpha = 1
        if (nqm != nq) throw Error("cannot NewParametersFromLiteral: length of Q & QMul is not equal");
        if (alpha != 1) throw Error("mkhe: mkbfv needs PCount/gamma = 1");
        if (nq > BC_MAXS) throw Error("mkhe: too many primes in Q for the BFV basis conversion");
        if (T < 2) throw Error("mkhe: bad plaintext modulus");
    }
    for (int i = 0; i < mall; ++i) {
        const u64 q = moduli[i];
        if (q >= (1ull << 60) || !is_prime(q) || (q - 1) % (2ull * N) != 0) throw Error("mkhe: moduli must be primes < 2^60 with q = 1 mod 2N");   // 4q < 2^62: range of the lazy butterflies
        for (int j = 0; j < i; ++j) if (moduli[j] == q) throw Error("mkhe: repeated modulus");
    }
    if (mall > NTT_MAX_SLOTS) throw Error("mkhe: too many moduli");
    // moduli whose forward NTT runs without reductions (MODE 1): the signed lazy values stay below 14.65q (4q input + 16
    // stages of 0.65q, N = 2^16 included), internal digits leave the kernel as x + 16q < 31q and every consumer
    // (mont_mul_lazy / mont_mul_sd) takes operands below 2^62.  2^62 / 31 = 2^57.05: the 57-bit head prime of the
    // reference's PN14QP433 chain (2^57 + 0x2b0001) is still in; the 59/60-bit primes of PN15QP880 are not.
    for (int i = 0; i < mall; ++i) small_q_.push_back(moduli[i] < (1ull << 62) / 31 ? 1 : 0);
    // the H16 kernel (ntt16_kernels.hip) grows its never-reduced values by up to 1.03q per stage (one-round product): < 4q + 15 * 1.03q
    // < 20q, internal digits leave as x + 24q < 44q -- small-class moduli have to satisfy 48q < 2^62 there
    // Round 4: a modulus in that gap (31q < 2^62 <= 48q: the head prime 2^57 + 0x2b0001 of cnn's PN14QP433) takes the balanced path of the H16 / H32
    // kernels as a member of their LONG class instead (partial reductions where its schedule asks for them -- none at this size -- and canonical
    // outputs): small16_ is the class map those launchers get.  Until then one such prime kept a whole context on the round-1 kernels: every
    // forward and inverse NTT of the cnn ring (71 us per launch where the H16 family takes 20).  N = 2^16 keeps the old rule: its split launches are
    // cut into class parts by small_q_ before the H16 sub-transforms see them.
    for (int i = 0; i < mall; ++i) small16_.push_back(moduli[i] < (1ull << 62) / 48 ? 1 : 0);
    for (int i = 0; i < mall; ++i) if (logN == 16 && small_q_[i] && !small16_[i]) h16_gap_ = true;
    MKHE_HIP(hipSetDevice(device));
#ifdef MKHE_CU_PARTITION_EXPERIMENT
    // experiment (tools/build_variant.sh cupart engine -DMKHE_CU_PARTITION_EXPERIMENT): the streams of the contexts of a process take
    // turns on MKHE_CU_PARTS disjoint sets of CUs -- two MulRelin in flight, each on its own half of the chip, the ALU-bound phases of one beside the
    // bandwidth- and latency-bound phases of the other
    {
        static std::atomic<int> nctx{0};
        const char* e = std::getenv("MKHE_CU_PARTS");
        const int parts = e ? std::atoi(e) : 0;
        if (parts >= 2 && parts <= 8) {
            int ncu = 256;
            (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
            const char* m = std::getenv("MKHE_CU_PART_MODE");          // 0: contiguous bit ranges, 1: bits interleaved by `parts`
            const int mode = m ? std::atoi(m) : 0, idx = nctx++ % parts;
            uint32_t mask[16] = {};
            for (int cu = 0; cu < ncu && cu < 512; ++cu) {
                const int owner = mode ? cu % parts : (int)((long)cu * parts / ncu);
                if (owner == idx) mask[cu / 32] |= 1u << (cu % 32);
            }
            MKHE_HIP(hipExtStreamCreateWithCUMask(&stream, (uint32_t)((ncu + 31) / 32), mask));
            MKHE_HIP(hipExtStreamCreateWithCUMask(&stream2, (uint32_t)((ncu + 31) / 32), mask));
        } else {
            MKHE_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
            MKHE_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
        }
    }
#else
    MKHE_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    MKHE_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
#endif
    for (auto& e : ev_) MKHE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s_ = stream;

    std::vector<Mod> mods(mall);
    // inverse NTT constants, 6 words per modulus: whole transform, then the two sub-transforms of a split limb
    std::vector<u64> psi((size_t)mall * N), psiinv((size_t)mall * N), aux(6 * (size_t)mall);
    for (int i = 0; i < mall; ++i) {
        const u64 q = moduli[i];
        Mod& m = mods[i];
        m.q = q; m.q2 = 2 * q; m.qinv = inv64(q); m.ninv32 = (u32)(0 - m.qinv);
        { const float f = (float)(4294967296.0 / (double)q); std::memcpy(&m.finv, &f, 4); }
        m.r1 = to_mont(1, q); m.r2 = mulmod(m.r1, m.r1, q);
        m.qs = sd_split(q); m.r1s = sd_split(m.r1);
        u64 ps = (i < nq) ? (psiQ ? psiQ[i] : 0) : (i < mtot ? (psiP ? psiP[i - nq] : 0) : 0);
        if (!ps) ps = default_psi(q, N);
        if (powmod(ps, N, q) != q - 1) throw Error("mkhe: supplied psi is not a primitive 2N-th root");
        psi_plain.push_back(ps);
        const u64 psinv = powmod(ps, q - 2, q);
        u64 a = m.r1, b = m.r1;
        for (u64 j = 0; j < (u64)N; ++j) {
            const u64 r = bitrev(j, logN);
            psi[(size_t)i * N + r] = a; psiinv[(size_t)i * N + r] = b;
            a = mulmod(a, ps, q); b = mulmod(b, psinv, q);
        }
        const u64 ninv = powmod((u64)N, q - 2, q);
        for (int h = 0; h < 3; ++h) {
            aux[6 * i + 2 * h] = to_mont(ninv, q);
            aux[6 * i + 2 * h + 1] = mulmod(psiinv[(size_t)i * N + (h == 0 ? 1 : 1 + h)], ninv, q);
        }
    }
    // the NTT kernels take twiddles and the inverse-NTT constants in signed-split form (mont_mul_sd)
    for (auto& v : psi) v = sd_split(v);
    for (auto& v : psiinv) v = sd_split(v);
    for (auto& v : aux) v = sd_split(v);
    d_mods = dev_upload(mods); d_psi = dev_upload(psi); d_psiinv = dev_upload(psiinv); d_inv_aux = dev_upload(aux);
    if (logN >= 14 && !h16_gap_) {
        // one-round product of the H16 kernel: a * w = a0 * u + a1 * u' with u = w 2^31, u' = w 2^63 (mod q, balanced), then ONE Montgomery
        // round of radix 2^31.  psi holds w * 2^64 in signed-split form at this point: w * 2^31 = psi * 2^-33.
        // Round 3, "U class" (160 q < 2^62: 4q of input + 75q of growth either way + the 75q bias of internal digits): the data word enters
        // the product as hi 2^32 + lo with lo UNSIGNED, the pair is u = w 2^30 mod q in [0, q) with non-negative radix-2^30 digits and
        // v = w 2^62 mod q balanced, one round of radix 2^30 (ntt16_kernels.hip mm30u).  psi31n: the pairs of -psi[1..3] (the second pass of
        // the cross-half stage multiplies by -w; the unsigned digits cannot be negated in place), in the format of the modulus's class.
        static const int uclass_on = MKHE_AB_INT("MKHE_H16_UCLASS", 1);
        std::vector<u64> p31(2 * (size_t)mall * N), p31n(8 * (size_t)mall, 0);
        for (int i = 0; i < mall; ++i) {
            const u64 q = moduli[i];
            const bool uc = uclass_on && small16_[i] && q < (1ull << 62) / 160;
            if (uc) u_mods_ |= 1ull << i;
            const int sh = uc ? 30 : 31;
            const u64 cinv = powmod(powmod(2, 64 - sh, q), q - 2, q), c32 = powmod(2, 32, q);
            auto pack = [q, sh](u64 x, bool balanced) {
                const i64 b = balanced && x > q / 2 ? (i64)x - (i64)q : (i64)x;
                i64 d0 = (i64)((u64)b & ((1ull << sh) - 1));
                if (balanced && d0 >= (1ll << (sh - 1))) d0 -= 1ll << sh;             // low digit, sign-extended
                const i64 d1 = (b - d0) >> sh;
                return (u64)(u32)(i32)d0 | ((u64)(u32)(i32)d1 << 32);
            };
            auto pair_of = [&](u64 wR, u64* out) {                                   // wR = w 2^64 mod q
                const u64 u = mulmod(wR, cinv, q), v = mulmod(u, c32, q);           // w 2^sh, w 2^(sh + 32)
                out[0] = pack(u, !uc); out[1] = pack(v, true);
            };
            for (size_t j = 0; j < (size_t)N; ++j) {
                const u64 ps = psi[(size_t)i * N + j];
                const u64 wR = ps - ((u64)((u32)ps >> 31) << 32);                  // undo sd_split
                pair_of(wR, &p31[2 * ((size_t)i * N + j)]);
                if (j >= 1 && j <= 3) pair_of(wR ? q - wR : 0, &p31n[8 * (size_t)i + 2 * j]);
            }
        }
        d_psi31 = dev_upload(p31); d_psi31n = dev_upload(p31n);
        if (logN == 15) {
            // H32 (ntt32_kernels.hip): in its last register phase thread t = 64 wave + lane holds the coefficients 32 t .. 32 t + 31, and pair
            // number k = 2^j - 1 + i of the phase (stage 10 + j, i < 2^j) is twiddle ((1024 + t) << j) + i: sixteen consecutive pairs per lane in
            // the last stage, i.e. a stride of 256 B between the lanes of one load instruction (64 cache lines touched, 16 B of each used, and the
            // L1 does not hold them until the other fifteen loads come).  Stored once more in load order, every instruction reads 1 KiB in a row.
            std::vector<u64> p31c((size_t)mall * 16 * 31 * 64 * 2);
            for (int i = 0; i < mall; ++i)
                for (int w = 0; w < 16; ++w)
                    for (int k = 0; k < 31; ++k) {
                        int j = 0; while ((2 << j) - 1 <= k) ++j;
                        const int ii = k + 1 - (1 << j);
                        for (int l = 0; l < 64; ++l) {
                            const size_t tw = ((size_t)(1024 + 64 * w + l) << j) + ii;
                            const size_t o = 2 * ((((size_t)i * 16 + w) * 31 + k) * 64 + l);
                            p31c[o] = p31[2 * ((size_t)i * N + tw)]; p31c[o + 1] = p31[2 * ((size_t)i * N + tw) + 1];
                        }
                    }
            d_psi31c = dev_upload(p31c);
            // ... and its middle phase (stages 5..9): thread = (bits 14..10, bits 4..0) with wave = bits 14..11, so that a wave uses TWO sets of 31 pairs
            // (bit 10 = lane >> 5).  Per modulus and wave one 1 KiB row [half][31 pairs] (+ 2 unused entries): loaded by ONE coalesced instruction
            // per limb, parked in LDS and read from there by broadcast -- instead of 31 per-lane 16-byte loads with two distinct addresses each
            std::vector<u64> p31b((size_t)mall * 16 * 64 * 2, 0);
            for (int i = 0; i < mall; ++i)
                for (int w = 0; w < 16; ++w)
                    for (int h = 0; h < 2; ++h)
                        for (int k = 0; k < 31; ++k) {
                            int j = 0; while ((2 << j) - 1 <= k) ++j;
                            const size_t tw = ((size_t)(32 + 2 * w + h) << j) + (k + 1 - (1 << j));
                            const size_t o = 2 * (((size_t)i * 16 + w) * 64 + 31 * h + k);
                            p31b[o] = p31[2 * ((size_t)i * N + tw)]; p31b[o + 1] = p31[2 * ((size_t)i * N + tw) + 1];
                        }
            d_psi31b = dev_upload(p31b);
        }
        // the inverse kernel of the same family (ntt14_inv_kernel): the inverse twiddles as pairs in the format of the modulus's class, and per
        // modulus and sub-transform root 1..7 the two constants of its last stage, N^-1 and psiinv[root] N^-1 (N = this ring's degree)
        std::vector<u64> pi31(2 * (size_t)mall * N), fin((size_t)mall * 8 * 6, 0);
        for (int i = 0; i < mall; ++i) {
            const u64 q = moduli[i];
            const bool uc = ((u_mods_ >> i) & 1) != 0;
            const int sh = uc ? 30 : 31;
            const u64 cinv = powmod(powmod(2, 64 - sh, q), q - 2, q), c32 = powmod(2, 32, q);
            auto pack = [q, sh](u64 x, bool balanced) {
                const i64 b = balanced && x > q / 2 ? (i64)x - (i64)q : (i64)x;
                i64 d0 = (i64)((u64)b & ((1ull << sh) - 1));
                if (balanced && d0 >= (1ll << (sh - 1))) d0 -= 1ll << sh;
                const i64 d1 = (b - d0) >> sh;
                return (u64)(u32)(i32)d0 | ((u64)(u32)(i32)d1 << 32);
            };
            auto pair_of = [&](u64 wR, u64* out) {                                   // wR = w 2^64 mod q
                const u64 u = mulmod(wR, cinv, q), v = mulmod(u, c32, q);
                out[0] = pack(u, !uc); out[1] = pack(v, true);
            };
            const u64 ninvR = to_mont(powmod((u64)N, q - 2, q), q);
            for (size_t j = 0; j < (size_t)N; ++j) {
                const u64 ps = psiinv[(size_t)i * N + j];
                const u64 wR = ps - ((u64)((u32)ps >> 31) << 32);                  // undo sd_split
                pair_of(wR, &pi31[2 * ((size_t)i * N + j)]);
                if (j >= 1 && j < 8) {
                    const u64 wn = mulmod(wR, powmod((u64)N, q - 2, q), q);            // psiinv[j] N^-1 R
                    u64* f = &fin[((size_t)i * 8 + j) * 6];
                    pair_of(ninvR, f); pair_of(wn, f + 2);
                    f[4] = sd_split(ninvR); f[5] = sd_split(wn);                        // the same two for the two-round product
                }
            }
            if (small_q_[i]) small_mods_ |= 1ull << i;
        }
        d_psiinv31 = dev_upload(pi31); d_inv31c = dev_upload(fin);
        // F class (N = 2^16 only: the quarter sub-transforms behind the radix-4 producers): moduli with 80 q < 2^52 run double-precision butterflies;
        // their forward twiddles as plain residues in double format (MKHE_H16_FCLASS=0 switches the class off)
        static const int fclass_on = MKHE_AB_INT("MKHE_H16_FCLASS", 1);
        if (fclass_on && logN == 16) {
            std::vector<u64> pf((size_t)mall * N, 0);
            for (int i = 0; i < mall; ++i) {
                const u64 q = moduli[i];
                if (!(q < (1ull << 52) / 80)) continue;
                f_mods_ |= 1ull << i;
                const u64 rinv = powmod(powmod(2, 64, q), q - 2, q);
                for (size_t j = 0; j < (size_t)N; ++j) {
                    const u64 ps = psi[(size_t)i * N + j];
                    const double w = (double)mulmod(ps - ((u64)((u32)ps >> 31) << 32), rinv, q);      // undo sd_split, leave Montgomery form
                    std::memcpy(&pf[(size_t)i * N + j], &w, 8);
                }
            }
            if (f_mods_) d_psif = dev_upload(pf);
        }
        // Reduction schedule of the balanced path for inputs below 2^60 (canonical digits of any modulus): the never-reduced values must stay
        // below 2^62.9 (column sums of mm31) and grow by at most 1.03q per one-round stage (q + |x| q / 2^64) and q/2 + |x|/16 per two-round
        // stage of phase D; a partial reduction leaves |x| <= 0.51q.  Greedy from the load: reduce only where the next phase would overflow.
        static const int lightsched = MKHE_AB_INT("MKHE_H16_SCHED", 1);
        h16_sched_.assign(mall, 15);
        for (int i = 0; i < mall && lightsched; ++i) {
            const double q = (double)moduli[i], H = 0.98 * 8.6e18 / q;          // 2^62.9 = 8.606e18, 2 % of slack on the bound
            double b = 1152921504606846976.0 / q;                                // 2^60 / q
            int sc = 0;
            if (b + 5 * 1.04 > H) { sc |= 1; b = 0.51; }
            b += 5 * 1.04;                                                       // stage 0 + phase A
            if (b + 4 * 1.04 > H) { sc |= 2; b = 0.51; }
            b += 4 * 1.04;                                                       // phase B
            if (b + 4 * 1.04 > H) { sc |= 4; b = 0.51; }
            b += 4 * 1.04;                                                       // phase C
            // phase D runs on the two-round product, whose second round needs |a| < 2^62 (mont_mul_sd: m2 q0 + a1 w0 + ... < 2^63)
            if (b > 0.98 * 4.611686018427388e18 / q) sc |= 8;
            h16_sched_[i] = (unsigned char)sc;
        }
    }

    std::vector<int> map((size_t)nq * mtot, 0), ident(mtot);
    for (int l = 0; l < nq; ++l) {
        for (int j = 0; j <= l; ++j) map[(size_t)l * mtot + j] = j;
        for (int j = 0; j < np; ++j) map[(size_t)l * mtot + l + 1 + j] = nq + j;
    }
    for (int j = 0; j < mtot; ++j) ident[j] = j;
    d_map_qp = dev_upload(map); d_map_id = dev_upload(ident);

    // ModUpPtoQ / ModDown constants, basisextenderparameters(P, Q) at full P
    // (mkrlwe/basis_extension.go:34-54, 83-153); all are canonical values -> closed forms.
    std::vector<u64> t1(np), t2((size_t)nq * np), t3((size_t)nq * (np + 1)), t4(nq), t5(nq);
    for (int i = 0; i < np; ++i) {
        const u64 pi = P[i]; u64 star = 1;
        for (int j = 0; j < np; ++j) if (j != i) star = mulmod(star, P[j] % pi, pi);
        t1[i] = to_mont(powmod(star, pi - 2, pi), pi);
    }
    for (int j = 0; j < nq; ++j) {
        const u64 qj = Q[j]; u64 pm = 1;
        for (int i = 0; i < np; ++i) {
            u64 s = 1;
            for (int u = 0; u < np; ++u) if (u != i) s = mulmod(s, P[u] % qj, qj);
            t2[(size_t)j * np + i] = to_mont(s, qj);
            pm = mulmod(pm, P[i] % qj, qj);
        }
        const u64 v = qj - pm;
        t3[(size_t)j * (np + 1)] = 0;
        for (int i = 1; i <= np; ++i) { u64 s = t3[(size_t)j * (np + 1) + i - 1] + v; t3[(size_t)j * (np + 1) + i] = s >= qj ? s - qj : s; }
        t4[j] = qj - to_mont(powmod(pm, qj - 2, qj), qj);
        t5[j] = to_mont(pm, qj);
    }
    d_pmodq = dev_upload(t5);
    d_md_qoverqiinvqi = dev_upload(t1); d_md_qoverqimodp = dev_upload(t2); d_md_vtimes = dev_upload(t3); d_md_down = dev_upload(t4);

    // RescaleParams[L-1][i] = MForm(q_L^-1 mod q_i)  (lattigo ring.go genNTTParams)
    // second half of the table (round 5): BRedAdd((q_L - 1) / 2, q_i) = h mod q_i, the other per-(level, limb) constant of DivRoundByLastModulus -- the
    // merged ModDown computed it per coefficient and limb with a float64 division, 9 % of its instructions
    std::vector<u64> rs(2 * (size_t)nq * nq, 0);
    for (int L = 1; L < nq; ++L)
        for (int i = 0; i < L; ++i) {
            rs[(size_t)(L - 1) * nq + i] = to_mont(powmod(Q[L] % Q[i], Q[i] - 2, Q[i]), Q[i]);
            rs[(size_t)nq * nq + (size_t)(L - 1) * nq + i] = ((Q[L] - 1) >> 1) % Q[i];
        }
    d_rescale = dev_upload(rs);

    if (alpha > 1) {
        // NewDecomposer (mkrlwe/basis_extension.go:368-424) at full P: for digit d and nd = 2..alpha limbs,
        // basisextenderparameters(Q[alpha*d : alpha*d+nd], Q || P) in closed form
        if (alpha > DEC_MAXA) throw Error("mkhe: PCount/gamma > 4 is not supported");
        if (beta_max > 64) throw Error("mkhe: too many gadget digits");
        const int na = alpha - 1;
        std::vector<u64> ta((size_t)beta_max * na * DEC_MAXA, 0), tb((size_t)beta_max * na * mtot * DEC_MAXA, 0),
                         tc((size_t)beta_max * na * mtot * (DEC_MAXA + 1), 0);
        for (int d = 0; d < beta_max; ++d) {
            for (int nd = 2; nd <= alpha; ++nd) {
                if (d * alpha + nd > nq) break;
                const u64* S = Q + d * alpha;
                const size_t sel = (size_t)d * na + (nd - 2);
                for (int i = 0; i < nd; ++i) {
                    const u64 si = S[i]; u64 star = 1;
                    for (int j = 0; j < nd; ++j) if (j != i) star = mulmod(star, S[j] % si, si);
                    ta[sel * DEC_MAXA + i] = to_mont(powmod(star, si - 2, si), si);
                }
                for (int m = 0; m < mtot; ++m) {
                    const u64 tj = moduli[m]; u64 pm = 1;
                    for (int i = 0; i < nd; ++i) {
                        u64 sprod = 1;
                        for (int u = 0; u < nd; ++u) if (u != i) sprod = mulmod(sprod, S[u] % tj, tj);
                        tb[(sel * mtot + m) * DEC_MAXA + i] = to_mont(sprod, tj);
                        pm = mulmod(pm, S[i] % tj, tj);
                    }
                    const u64 v = tj - pm;            // tj - (Q_d mod tj); equals tj (not 0) when tj divides Q_d, as in the reference
                    u64* c = &tc[(sel * mtot + m) * (DEC_MAXA + 1)];
                    c[0] = 0;
                    for (int i = 1; i <= nd; ++i) { u64 sum = c[i - 1] + v; c[i] = sum >= tj ? sum - tj : sum; }
                }
            }
        }
        d_dec_a = dev_upload(ta); d_dec_b = dev_upload(tb); d_dec_c = dev_upload(tc);
        // Round 3, radix-4 digit spread in front of the N = 2^16 forward NTT (poly_kernels.hip decomp_spread4_kernel): two-limb digits and moduli
        // below 2^57 only; every constant t as the pair (t 2^30 mod p, t 2^62 mod p), radix-2^30 digits in the two halves of a word
        bool all57 = true;
        for (int m = 0; m < mtot; ++m) all57 = all57 && moduli[m] < (1ull << 57);
        if (alpha == 2 && logN == 16 && all57) {
            auto pack30 = [](u64 x) { return (x & ((1ull << 30) - 1)) | ((x >> 30) << 32); };
            std::vector<u64> tb30((size_t)beta_max * mtot * 4, 0), tw30((size_t)mtot * 8, 0);
            for (int m = 0; m < mtot; ++m) {
                const u64 tj = moduli[m], c30 = powmod(2, 30, tj), c62 = powmod(2, 62, tj);
                for (int d = 0; d < beta_max; ++d) {
                    if (d * 2 + 2 > nq) break;
                    const u64* S = Q + d * 2;
                    for (int i = 0; i < 2; ++i) {
                        const u64 t = S[1 - i] % tj;                       // Q_d / q_i mod p
                        tb30[((size_t)d * mtot + m) * 4 + 2 * i] = pack30(mulmod(t, c30, tj));
                        tb30[((size_t)d * mtot + m) * 4 + 2 * i + 1] = pack30(mulmod(t, c62, tj));
                    }
                }
                const u64 rinv = powmod(powmod(2, 64, tj), tj - 2, tj);
                for (int j = 0; j < 4; ++j) {
                    u64 t = 1;
                    if (j) { const u64 ps = psi[(size_t)m * N + j]; t = mulmod(ps - ((u64)((u32)ps >> 31) << 32), rinv, tj); }   // undo sd_split, leave Montgomery form
                    tw30[(size_t)m * 8 + 2 * j] = pack30(mulmod(t, c30, tj));
                    tw30[(size_t)m * 8 + 2 * j + 1] = pack30(mulmod(t, c62, tj));
                }
            }
            d_tb30 = dev_upload(tb30); d_tw30 = dev_upload(tw30);
        }
    }

    if (nqm) {
        // convQQMul = mkrlwe.NewFastBasisExtender(ringQ, ringQMul) at full levels (mkbfv/basis_extension.go:36),
        // basisextenderparameters (mkrlwe/basis_extension.go:83-153) in closed form, both directions
        auto conv = [&](const u64* S, const u64* Tm, int n, u64*& d1, u64*& d2, u64*& d3) {
            std::vector<u64> a(n), b((size_t)n * n), c((size_t)n * (n + 1));
            for (int i = 0; i < n; ++i) {
                const u64 si = S[i]; u64 star = 1;
                for (int j = 0; j < n; ++j) if (j != i) star = mulmod(star, S[j] % si, si);
                a[i] = to_mont(powmod(star, si - 2, si), si);
            }
            for (int j = 0; j < n; ++j) {
                const u64 tj = Tm[j]; u64 pm = 1;
                for (int i = 0; i < n; ++i) {
                    u64 sprod = 1;
                    for (int u = 0; u < n; ++u) if (u != i) sprod = mulmod(sprod, S[u] % tj, tj);
                    b[(size_t)j * n + i] = to_mont(sprod, tj);
                    pm = mulmod(pm, S[i] % tj, tj);
                }
                const u64 v = tj - pm;
                c[(size_t)j * (n + 1)] = 0;
                for (int i = 1; i <= n; ++i) { u64 sum = c[(size_t)j * (n + 1) + i - 1] + v; c[(size_t)j * (n + 1) + i] = sum >= tj ? sum - tj : sum; }
            }
            d1 = dev_upload(a); d2 = dev_upload(b); d3 = dev_upload(c);
        };
        conv(Q, QMul, nq, d_bq_qoverqiinvqi, d_bq_qoverqimodp, d_bq_vtimes);
        conv(QMul, Q, nq, d_bm_qoverqiinvqi, d_bm_qoverqimodp, d_bm_vtimes);
        std::vector<u64> dqm(nq), dmq(nq), mf(nq), tm(2 * (size_t)nq);
        std::vector<int> mr(2 * (size_t)nq);
        for (int i = 0; i < nq; ++i) {
            u64 qprod = 1, mprod = 1;                       // Q mod qm_i ; QMul mod q_i
            for (int j = 0; j < nq; ++j) { qprod = mulmod(qprod, Q[j] % QMul[i], QMul[i]); mprod = mulmod(mprod, QMul[j] % Q[i], Q[i]); }
            dqm[i] = QMul[i] - to_mont(powmod(qprod, QMul[i] - 2, QMul[i]), QMul[i]);    // genModDownParams, tail of ModDownQPtoP
            dmq[i] = Q[i] - to_mont(powmod(mprod, Q[i] - 2, Q[i]), Q[i]);                // tail of ModDownQPtoQ
            mf[i] = to_mont(mprod, Q[i]);                                                // mFormQMul (basis_extension.go:42-44)
            tm[i] = to_mont(T % Q[i], Q[i]); tm[nq + i] = to_mont(T % QMul[i], QMul[i]); // MulScalar(t) constants
            mr[i] = i; mr[nq + i] = mtot + i;
        }
        d_down_q_in_m = dev_upload(dqm); d_down_m_in_q = dev_upload(dmq); d_mform_qmul = dev_upload(mf); d_t_mont = dev_upload(tm);
        d_map_r = dev_upload(mr);
        x2_ = dev_alloc_words(swk_words()); y2_ = dev_alloc_words(swk_words());
    }
    x_ = dev_alloc_words(swk_words()); y_ = dev_alloc_words(swk_words()); swk3_ = dev_alloc_words(swk_words());
    c1_ = dev_alloc_words((size_t)mtot * N);
    for (auto& p : polyq_) p = dev_alloc_words((size_t)nq * N);
    invntt_ = dev_alloc_words((size_t)nq * N);
}

Context::~Context() {
    (void)hipSetDevice(device);
    // both streams: a pool of another context skips its fence for a context that is gone ("its destructor drained its streams")
    if (stream2) (void)hipStreamSynchronize(stream2);
    if (stream) (void)hipStreamSynchronize(stream);
    registry_remove();
    release_all();
}
void Context::release_all() noexcept {
    (void)hipSetDevice(device);
    for (void* p : {(void*)d_mods, (void*)d_psi, (void*)d_psiinv, (void*)d_inv_aux, (void*)d_map_qp, (void*)d_map_id,
                    (void*)d_md_qoverqiinvqi, (void*)d_md_qoverqimodp, (void*)d_md_vtimes, (void*)d_md_down, (void*)d_rescale, (void*)d_pmodq, (void*)tens_, (void*)d_psi31, (void*)d_psi31n, (void*)d_psi31c, (void*)d_psi31b, (void*)d_psiinv31, (void*)d_inv31c, (void*)d_psif, (void*)spreadbuf_,
                    (void*)d_dec_a, (void*)d_dec_b, (void*)d_dec_c, (void*)d_tb30, (void*)d_tw30, (void*)d_map_own, (void*)d_ownq,
                    (void*)x_, (void*)y_, (void*)swk3_, (void*)c1_, (void*)polyq_[0], (void*)polyq_[1], (void*)polyq_[2],
                    (void*)invntt_, (void*)nttbuf_, (void*)ctbuf_, (void*)c1b_, (void*)tbuf_, (void*)rbuf_, (void*)x2_, (void*)y2_,
                    (void*)d_map_r, (void*)d_bq_qoverqiinvqi, (void*)d_bq_qoverqimodp, (void*)d_bq_vtimes,
                    (void*)d_bm_qoverqiinvqi, (void*)d_bm_qoverqimodp, (void*)d_bm_vtimes,
                    (void*)d_down_q_in_m, (void*)d_down_m_in_q, (void*)d_mform_qmul, (void*)d_t_mont,
                    (void*)kg_small_, (void*)kg_g_, (void*)kg_sk_})
        if (p) (void)hipFree(p);
    for (auto& v : hoist_pool_) for (auto& s : v) if (s.d) (void)hipFree(s.d);
    for (auto& kv : f2_sched_) if (kv.second.d_segs) (void)hipFree(kv.second.d_segs);
    f2_sched_.clear();
    { std::vector<u64*> mine; (void)pool_take_all(mine); for (u64* p : mine) (void)hipFree(p); }
    for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
    for (auto& kv : ntt_tune_) for (int i = 0; i < NttTune::RING; ++i) if (kv.second.e0[i]) (void)hipEventDestroy(kv.second.e0[i]);
    ntt_tune_.clear();
    if (fence_ev_) (void)hipEventDestroy(fence_ev_);
    if (xev_) (void)hipEventDestroy(xev_);
    if (stream2) (void)hipStreamDestroy(stream2);
    if (stream) (void)hipStreamDestroy(stream);
    stream = stream2 = nullptr;
}

// live contexts per device: who may still be using a buffer that some context returns to its pool
namespace {
// pooled_words: what the free lists of ALL contexts of the device hold together -- the cache bound is one budget per device (a forked
// BatchEvaluator has four contexts; a bound per context let a process run out of memory with most of HBM in free lists)
struct DeviceRegistry { std::mutex mu; std::vector<Context*> live; unsigned long long next_uid = 1; std::atomic<size_t> pooled_words{0}; };
DeviceRegistry& registry(int device) { static DeviceRegistry r[64]; return r[device & 63]; }
// MKHE_POOL_GB (configuration, include/mkhe.h): device-wide bound of the pools in GiB, fractions allowed (0 = keep nothing)
size_t pool_cap_words() {
    static const size_t cap = [] { const char* e = getenv("MKHE_POOL_GB"); const double gb = (e && *e) ? atof(e) : 32.0; return (size_t)((gb < 0 ? 0 : gb) * (double)(1ull << 27)); }();
    return cap;
}
}
// every free-list entry of every live context of the current device goes back to the driver; returns the words released
static size_t trim_device_pools() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::vector<u64*> victims;
    size_t words = 0;
    { auto& r = registry(dev); std::lock_guard<std::mutex> g(r.mu); for (Context* c : r.live) words += c->pool_take_all(victims); }
    if (victims.empty()) return 0;
    (void)hipDeviceSynchronize();                      // (entries may still be read by queued work of any context)
    for (u64* p : victims) (void)hipFree(p);
    return words;
}
size_t Context::pool_trim_device() { MKHE_HIP(hipSetDevice(device)); return trim_device_pools(); }
size_t Context::pool_take_all(std::vector<u64*>& out) {
    std::lock_guard<std::mutex> g(pool_mu_);
    size_t words = 0;
    for (auto& f : free_list_) { out.push_back(f.p); words += f.words; }
    free_list_.clear();
    registry(device).pooled_words.fetch_sub(words);
    return words;
}
size_t Context::pool_held_words() { std::lock_guard<std::mutex> g(pool_mu_); size_t w = 0; for (auto& f : free_list_) w += f.words; return w; }
void Context::registry_add() { auto& r = registry(device); std::lock_guard<std::mutex> g(r.mu); uid_ = r.next_uid++; r.live.push_back(this); }
void Context::registry_remove() {
    auto& r = registry(device); std::lock_guard<std::mutex> g(r.mu);
    for (size_t i = 0; i < r.live.size(); ++i) if (r.live[i] == this) { r.live.erase(r.live.begin() + i); break; }
}
u64* Context::pool_alloc(size_t words) {
    FreeEntry e{0, nullptr, {}};
    {
        std::lock_guard<std::mutex> g(pool_mu_);
        for (size_t i = 0; i < free_list_.size(); ++i)
            if (free_list_[i].words == words) {
                e = std::move(free_list_[i]);
                free_list_.erase(free_list_.begin() + i);          // the oldest matching entry: most likely already ordered
                registry(device).pooled_words.fetch_sub(words);
                break;
            }
    }
    if (e.p) {
        {
            if (!e.behind.empty()) {
                hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
                (void)hipStreamIsCapturing(stream, &cs);
                if (cs == hipStreamCaptureStatusNone) {           // (a capture keeps its temporaries alive itself)
                    auto& r = registry(device);
                    std::lock_guard<std::mutex> g(r.mu);
                    for (const auto& b : e.behind) {
                        Context* m = nullptr;
                        for (Context* x : r.live) if (x->uid_ == b.first) m = x;
                        if (!m) continue;                         // destroyed since: its destructor drained its streams
                        if (std::max(synced_with(m->uid_), m->completed_.load()) >= b.second) continue;
                        const seq_t upto = m->enqueued_.load();     // read before the record: every call that has finished enqueuing is covered by it (a call
                                                                     // still in flight on another thread is counted in seq_ but its kernels come after the event)
                        if (!m->fence_ev_) MKHE_HIP(hipEventCreateWithFlags(&m->fence_ev_, hipEventDisableTiming));
                        MKHE_HIP(hipEventRecord(m->fence_ev_, m->stream));
                        MKHE_HIP(hipStreamWaitEvent(stream, m->fence_ev_, 0));
                        set_synced(m->uid_, upto);
                    }
                }
            }
            return e.p;
        }
    }
    MKHE_HIP(hipSetDevice(device));
    return dev_alloc_words(words);
}
void Context::note_use(HandleUsers& u) {
    touch();
    auto& r = registry(device);
    std::lock_guard<std::mutex> g(r.mu);
    const seq_t s = seq_.load();
    for (auto& e : u.v) if (e.first == uid_) { e.second = s; return; }
    if (u.v.size() >= 8) {            // a long-lived handle (a key) seen by many short-lived contexts: forget the ones that are gone
        size_t w = 0;
        for (size_t i = 0; i < u.v.size(); ++i) {
            bool live = false;
            for (const Context* m : r.live) if (m->uid_ == u.v[i].first) { live = true; break; }
            if (live) u.v[w++] = u.v[i];
        }
        u.v.resize(w);
    }
    u.v.push_back({uid_, s});
}
void Context::pool_free(u64* p, size_t words, const HandleUsers* users) {
    if (!p) return;
    // bound of the cache: entries and bytes.  Batched evaluation (batch.hip) frees and re-creates B output ciphertexts per operation, a few hundred
    // handles in flight at B = 16: the list is sized for that -- at the old bound of 64 entries every further free was a device-wide
    // synchronisation plus hipFree, and the next create a hipMalloc (119 ms per batched cnn step instead of 6).  Over the bound the OLDEST entry
    // goes back to the driver (its work is long done: one synchronisation, rare).
    // The byte bound is ONE budget per device (DeviceRegistry::pooled_words, a running counter), MKHE_POOL_GB GiB, default 32.
    auto& reg = registry(device);
    {
        const size_t cap_entries = 4096, cap_words = pool_cap_words();
        std::vector<u64*> evict;
        {
            std::lock_guard<std::mutex> g(pool_mu_);
            while (!free_list_.empty() && (free_list_.size() >= cap_entries || reg.pooled_words.load() + words > cap_words)) {
                reg.pooled_words.fetch_sub(free_list_.front().words);
                evict.push_back(free_list_.front().p);
                free_list_.erase(free_list_.begin());
            }
        }
        const bool keep = reg.pooled_words.load() + words <= cap_words;     // (other contexts hold the budget: this buffer goes back to the driver itself)
        if (!evict.empty() || !keep) {
            (void)hipSetDevice(device);
            (void)hipDeviceSynchronize();
            for (u64* q : evict) (void)hipFree(q);
            if (!keep) { (void)hipFree(p); return; }
        }
    }
    FreeEntry e{words, p, {}};
    {
        auto& r = registry(device);
        std::lock_guard<std::mutex> g(r.mu);
        if (r.live.size() > 1)
            for (Context* m : r.live) {
                if (m == this) continue;
                seq_t s = 0;
                if (!users || users->exposed || m->external_.load()) s = m->now_seq();          // unknown uses: anything it enqueued so far
                else { for (const auto& u : users->v) if (u.first == m->uid_) s = u.second; if (!s) continue; }
                if (std::max(synced_with(m->uid_), m->completed_.load()) < s) e.behind.push_back({m->uid_, s});
            }
    }
    std::lock_guard<std::mutex> g(pool_mu_);
    reg.pooled_words.fetch_add(words);
    free_list_.push_back(std::move(e));
}
u64* Context::scratch(u64*& p, size_t& have, size_t want) {
    if (have < want) {
        if (p) { MKHE_HIP(hipStreamSynchronize(stream)); MKHE_HIP(hipFree(p)); p = nullptr; }
        p = dev_alloc_words(want); have = want;
    }
    return p;
}
Swk& Context::hoist_slot(int which, int idx) {
    auto& v = hoist_pool_[which];
    while ((int)v.size() <= idx) { Swk s; s.d = dev_alloc_words(swk_words()); v.push_back(s); }
    return v[idx];
}
void Context::check_level(int level) const {
    if (level < 0 || level >= nq) throw Error("mkhe: level out of range");
}

// ------------------------------------------------------------------ profiling
hipEvent_t Context::prof_event() {
    if (!prof_pool_.empty()) { hipEvent_t e = prof_pool_.back(); prof_pool_.pop_back(); return e; }
    hipEvent_t e; MKHE_HIP(hipEventCreate(&e)); return e;
}
Context::ProfScope::ProfScope(Context* c_, int cls, double bytes) : c(c_), idx(0), on(c_->prof_on_), st(c_->s_) {
    if (!on) return;
    ProfRec r{c->prof_event(), c->prof_event(), cls, bytes};
    (void)hipEventRecord(r.e0, st);
    idx = c->prof_recs_.size();
    c->prof_recs_.push_back(r);
}
Context::ProfScope::~ProfScope() { if (on) (void)hipEventRecord(c->prof_recs_[idx].e1, st); }
void Context::fork_side(int k) { if (!overlap) return; MKHE_HIP(hipEventRecord(ev_[2 * k], s_)); MKHE_HIP(hipStreamWaitEvent(stream2, ev_[2 * k], 0)); }
void Context::side_done(int k) { if (!overlap) return; MKHE_HIP(hipEventRecord(ev_[2 * k + 1], stream2)); }
void Context::join_side(int k) { if (!overlap) return; MKHE_HIP(hipStreamWaitEvent(s_, ev_[2 * k + 1], 0)); }
void Context::recover() {
    s_ = stream;
    plan_.valid = false; plan_.x_pending = false; plan_.head_done = false; plan_.xkeys.clear(); ext_xout_ = ext_xout2_ = nullptr;
    rs_maps_.clear();
    bfv_plan_valid_ = false; bfv_xk1_.clear(); bfv_xk2_.clear();
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cs);
    if (cs == hipStreamCaptureStatusNone) { (void)hipStreamSynchronize(stream2); (void)hipStreamSynchronize(stream); }
    (void)hipGetLastError();
}
void Context::wait_for(Context& other) {
    if (&other == this) return;
    if (other.device != device) throw Error("mkhe: wait_for needs two contexts on the same device");
    if (!xev_) MKHE_HIP(hipEventCreateWithFlags(&xev_, hipEventDisableTiming));
    const seq_t upto = other.enqueued_.load();
    MKHE_HIP(hipEventRecord(xev_, other.stream));
    MKHE_HIP(hipStreamWaitEvent(stream, xev_, 0));
    { auto& r = registry(device); std::lock_guard<std::mutex> g(r.mu); set_synced(other.uid_, upto); }
}
void Context::prof_enable(bool on) { sync(); prof_on_ = on; }
void Context::prof_collect(double* ms, long* launches, double* alg_bytes) {
    sync();
    for (int i = 0; i < PROF_NCLASS; ++i) { ms[i] = 0; launches[i] = 0; alg_bytes[i] = 0; }
    for (auto& r : prof_recs_) {
        float t = 0; MKHE_HIP(hipEventElapsedTime(&t, r.e0, r.e1));
        ms[r.cls] += t; launches[r.cls] += 1; alg_bytes[r.cls] += r.bytes;
        prof_pool_.push_back(r.e0); prof_pool_.push_back(r.e1);
    }
    prof_recs_.clear();
}

// slot lists of an NttBatch: limbs 0..level of Q then every P limb (PolyQP shaped buffers) ...
void Context::slots_qp(NttBatch& b, int level) const {
    if (masked_) {                       // limb sharding: only the owned slots (buffers are always addressed "mapped" here)
        const std::vector<int>& l = own_list_[level];
        b.nslots = (int)l.size();
        for (int k = 0; k < b.nslots; ++k) { b.mod[k] = l[k]; b.pos[k] = l[k] < nq ? l[k] : level + 1 + (l[k] - nq); }
        return;
    }
    b.nslots = level + 1 + np;
    for (int j = 0; j <= level; ++j) { b.mod[j] = j; b.pos[j] = j; }
    for (int j = 0; j < np; ++j) { b.mod[level + 1 + j] = nq + j; b.pos[level + 1 + j] = level + 1 + j; }
}
void Context::slots_q_owned(NttBatch& b, int L) const {
    if (!masked_) { slots_range(b, 0, L); return; }
    b.nslots = 0;
    for (int l : ownq_) if (l < L) { b.mod[b.nslots] = l; b.pos[b.nslots] = l; ++b.nslots; }
}

// limb sharding: which moduli this context computes (mod_idx over Q then P); n == 0 restores "everything"
void Context::set_owned(const int* mod_idx, int n) {
    sync();
    if (is_bfv() || alpha != 1) { if (n) throw Error("mkhe: limb sharding is wired for the mkckks path with one prime per digit only"); }
    own_.assign(mtot, n == 0 ? 1 : 0);
    for (int i = 0; i < n; ++i) { if (mod_idx[i] < 0 || mod_idx[i] >= mtot) throw Error("mkhe: owned modulus index out of range"); own_[mod_idx[i]] = 1; }
    masked_ = n != 0;
    own_list_.assign(nq, {}); own_cnt_.assign(nq, 0); ownq_.clear();
    std::vector<int> map((size_t)nq * mtot, 0);
    for (int l = 0; l < nq; ++l) {
        for (int j = 0; j <= l; ++j) if (own_[j]) own_list_[l].push_back(j);
        for (int j = 0; j < np; ++j) if (own_[nq + j]) own_list_[l].push_back(nq + j);
        own_cnt_[l] = (int)own_list_[l].size();
        for (int k = 0; k < own_cnt_[l]; ++k) map[(size_t)l * mtot + k] = own_list_[l][k];
    }
    for (int j = 0; j < nq; ++j) if (own_[j]) ownq_.push_back(j);
    if (d_map_own) MKHE_HIP(hipFree(d_map_own));
    if (d_ownq) MKHE_HIP(hipFree(d_ownq));
    d_map_own = dev_upload(map); d_ownq = dev_upload(ownq_);
}
// ... or `limbs` consecutive moduli starting at mod_base (plain polynomials)
void Context::slots_range(NttBatch& b, int mod_base, int limbs) const {
    b.nslots = limbs;
    for (int j = 0; j < limbs; ++j) { b.mod[j] = mod_base + j; b.pos[j] = j; }
}

// forward NTT launch: one kernel per modulus class, each with its own timing record
void Context::ntt_fwd_launch(const NttBatch& b_in, bool decompose) {
    NttBatch b = b_in;
    b.psi31 = d_psi31; b.psi31n = d_psi31n; b.psi31c = d_psi31c; b.psi31b = d_psi31b; b.u_mods = u_mods_; b.no_h16 = d_psi31 ? 0 : 1;
    b.psif = d_psif; b.f_mods = f_mods_;
    for (int i = 0; i < mall && i < NTT_MAX_SLOTS; ++i) b.sched[i] = h16_sched_.empty() ? 15 : h16_sched_[i];
    const bool ok32 = ntt32_ok(logN, b), ok16 = ntt16_ok(logN, b);
    NttTune* sampling = nullptr;
    int use32 = ok32 ? 1 : 0;
    if (ok32 && ok16 && ntt32_mode() == 2) use32 = ntt_pick((((long)b.nslots * b.nouter) << 2) | (decompose ? 2 : 0) | (b.src_lazy ? 1 : 0), sampling);
    if (sampling) (void)hipEventRecord(sampling->e0[sampling->slot], s_);
    if (use32) {
        ProfScope ps(this, decompose ? PROF_NTT32_DECOMP : PROF_NTT32_FWD, 16.0 * N * b.nouter * b.nslots);
        NttBatch bt = b; bt.trace = ntt_trace;
        launch_ntt32_fwd(bt, small16_.data(), s_);
    } else if (ok16) {
        ProfScope ps(this, decompose ? PROF_NTT16_DECOMP : PROF_NTT16_FWD, 16.0 * N * b.nouter * b.nslots);
        NttBatch bt = b; bt.trace = ntt_trace;
        launch_ntt16_fwd(bt, small16_.data(), s_, logN);
    }
    if (use32 || ok16) return;
    if (decompose && ntt_fwd_mixed_ok(logN, b, small_q_.data())) {
        // large Decompose launches: both modulus classes in one persistent grid (no ragged tail of the big-modulus class)
        ProfScope ps(this, PROF_NTT_DECOMP_MIXED, 16.0 * N * b.nouter * b.nslots);
        launch_ntt_fwd_mixed(logN, b, small_q_.data(), s_);
        return;
    }
    NttBatch part[2];
    const int n = split_ntt_fwd(b, small_q_.data(), part);
    for (int i = 0; i < n; ++i) part[i].trace = ntt_trace;
    // two classes = two kernels; the second one runs on the side stream so that their partial last
    // waves of workgroups (one workgroup per CU) fill each other's idle CUs
    const bool side = (n == 2) && (s_ == stream);
    for (int i = n - 1; i >= 0; --i) {
        const bool small = part[i].lazy_out != 0;
        const int cls = b.prestaged == 2 ? PROF_NTT14_SPLIT : decompose ? (small ? PROF_NTT_DECOMP : PROF_NTT_DECOMP_BIGQ) : (small ? PROF_NTT_FWD : PROF_NTT_FWD_BIGQ);
        const bool on_side = side && i == 1;
        if (on_side) { fork_side(0); s_ = overlap ? stream2 : stream; }
        {
            ProfScope ps(this, cls, 16.0 * N * part[i].nouter * part[i].nslots);
            launch_ntt_fwd_class(logN, part[i], s_);
        }
        if (on_side) { side_done(0); s_ = stream; }
    }
    if (side) join_side(0);
}

void Context::ntt_reset(NttTune& t, int choice) {
    for (int k = 0; k < 2; ++k) t.n[k] = t.req[k] = t.blk[k] = 0;
    t.head = t.inflight = 0; t.slot = -1; t.seen = 0;       // (the events are kept and re-recorded)
    t.decided = choice;
}
int Context::ntt_pick(long key, NttTune*& sampling) {
    constexpr int WARM = NttTune::WARM, SETTLE = NttTune::SETTLE, TIMED = NttTune::TIMED, RING = NttTune::RING;
    sampling = nullptr;
    { auto it = ntt_tune_.find(key); if (it == ntt_tune_.end() && ntt_forced_ >= 0) return ntt_forced_; }
    NttTune& t = ntt_tune_[key];
    if (t.decided >= 0) return t.decided;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s_, &cs);
    if (cs != hipStreamCaptureStatusNone) return 0;    // (no timing inside a graph capture: the recorded sequence keeps the two-pass kernel)
    // periods between consecutive timed launches of one block, oldest first, as far as the GPU has come
    while (t.inflight >= 2) {
        const int h = t.head, nx = (h + 1) % RING;
        if (t.which[nx] == t.which[h]) {
            float ms = 0.f;
            const hipError_t e = hipEventElapsedTime(&ms, t.e0[h], t.e0[nx]);
            if (e == hipErrorNotReady) { (void)hipGetLastError(); break; }
            if (e == hipSuccess && t.n[t.which[h]] < TIMED) t.t[t.which[h]][t.n[t.which[h]]++] = ms;
            else if (e != hipSuccess) --t.req[t.which[h]];             // (lost sample: one more launch of this kernel is asked for)
            (void)hipGetLastError();
        }
        t.head = nx; --t.inflight;
    }
    if (t.n[0] >= TIMED && t.n[1] >= TIMED) {
        auto med = [](float* v, int n) { std::sort(v, v + n); return v[n / 2]; };
        const float m32 = med(t.t[1], TIMED), m16 = med(t.t[0], TIMED);
        t.decided = m32 <= m16 ? 1 : 0;
        // a smaller shape follows the choice of the largest one decided so far unless the other kernel is ahead by more than 3 %
        long lead = -1;
        for (const auto& kv : ntt_tune_) if (kv.second.decided >= 0 && &kv.second != &t && (kv.first >> 2) > (key >> 2) && (kv.first >> 2) > (lead >> 2)) lead = kv.first;
        if (lead >= 0) {
            const int l = ntt_tune_[lead].decided;
            const float ml = l ? m32 : m16, mo = l ? m16 : m32;
            if (mo > 0.97f * ml) t.decided = l;
        }
        return t.decided;                              // (the events stay until the context goes: release_all)
    }
    // launch number `seen` of this shape: [0, WARM) on the two-pass kernel (whose time inside an operation is the same on every part met: the safe
    // choice for a shape that never gets as far as a decision); then a block of H32 and a block of H16 -- SETTLE launches, then TIMED + 1 launches with
    // a start event (TIMED periods) -- and H16 again until the last period has come back.  The host runs ahead of the GPU by up to hundreds of
    // launches: the blocks are counted in REQUESTED events (the GPU executes the launches in this order whenever it gets to them), the decision waits
    // for the harvest.
    if (t.seen < WARM) { ++t.seen; return 0; }
    const int k = t.req[1] < TIMED + 1 ? 1 : (t.req[0] < TIMED + 1 ? 0 : -1);
    if (k < 0 || t.inflight >= RING) return 0;
    if (t.blk[k]++ < SETTLE) return k;
    const int sl = (t.head + t.inflight) % RING;
    if (!t.e0[sl]) MKHE_HIP(hipEventCreate(&t.e0[sl]));
    t.which[sl] = k; t.slot = sl; ++t.inflight; ++t.req[k];
    sampling = &t;
    return k;
}

void Context::ntt_inv_launch(NttBatch& b) {
    b.psi31 = d_psiinv31; b.inv31c = d_inv31c; b.u_mods = u_mods_; b.small_mods = small_mods_; b.no_h16 = d_psiinv31 ? 0 : 1;
    launch_ntt_inv(logN, b, s_);
}

// ------------------------------------------------------------------ ring level
void Context::ntt(const u64* src, u64* dst, int count, int limbs, int mod_base, bool inverse, bool lazy) {
    if (mod_base < 0 || mod_base + limbs > mall) throw Error("mkhe: ntt modulus range");
    NttBatch b{};
    b.src = src; b.dst = dst; b.mods = d_mods; b.psi = inverse ? d_psiinv : d_psi; b.aux = d_inv_aux;
    if (limbs > NTT_MAX_SLOTS) throw Error("mkhe: too many limbs per polynomial");
    slots_range(b, mod_base, limbs);
    b.src_outer = b.dst_outer = (long)limbs * N; b.src_inner = b.dst_inner = N;
    b.nouter = count; b.lazy_out = lazy ? 1 : 0;
    if (inverse) { ProfScope ps(this, PROF_NTT_INV, 16.0 * N * count * limbs); ntt_inv_launch(b); }
    else ntt_fwd_launch(b, false);
    MKHE_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ Decompose (keyswitch.go:49-73)
void Context::decompose(int level, bool is_ntt, const u64* a, u64* out_swk) {
    check_level(level);
    staged_open_.erase(std::remove(staged_open_.begin(), staged_open_.end(), (const u64*)out_swk), staged_open_.end());
    const u64* ainv = a;
    if (is_ntt) { ntt(a, invntt_, 1, level + 1, 0, true, false); ainv = invntt_; }
    if (alpha != 1) { decompose_batch(level, {ainv}, {out_swk}); return; }
    // alpha = 1 (basis_extension.go:443-451): digit i = limb i, re-read under every active modulus,
    // fused with the forward NTT (DecomposeSingleNTT, keyswitch.go:21-31).
    NttBatch b{};
    b.src = ainv; b.dst = out_swk; b.mods = d_mods; b.psi = d_psi; b.aux = d_inv_aux;
    slots_qp(b, level);
    b.src_outer = N; b.src_inner = 0; b.src_mapped = 0;
    b.dst_outer = (long)mtot * N; b.dst_inner = N; b.dst_mapped = 1;
    b.nouter = beta(level);
    b.reduce_in = 1; b.reduce_src_mod_is_outer = 1;
    ntt_fwd_launch(b, true);
    MKHE_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ ExternalProduct[Hoisted]
// (keyswitch_hoisted.go:10-40): sum over digits, InvNTTLazy on Q and P parts, ModDownQPtoQ.
void Context::ext_core(int level, const u64* ah, const u64* bg, u64* c, bool accumulate) {
    const int nb = beta(level), nslots = nslots_qp(level);
    InnerProductArgs ip{};
    for (int i = 0; i < nb; ++i) { ip.a[i] = bg + (size_t)i * mtot * N; ip.b[i] = ah + (size_t)i * mtot * N; }
    ip.out = c1_; ip.mods = d_mods; ip.map = map_qp(level);
    ip.term_outer = 0; ip.out_outer = 0; ip.nterms = nb; ip.nslots = nslots; ip.nouter = 1; ip.N = N; ip.mform_out = 0;
    { ProfScope ps(this, PROF_INNER, 8.0 * N * nslots * (2.0 * nb + 1)); launch_inner_product(ip, s_); }

    NttBatch b{};
    b.src = c1_; b.dst = c1_; b.mods = d_mods; b.psi = d_psiinv; b.aux = d_inv_aux; slots_qp(b, level);
    b.nouter = 1; b.src_inner = b.dst_inner = N; b.src_mapped = b.dst_mapped = 1;
    b.lazy_out = 1;
    { ProfScope ps(this, PROF_NTT_INV, 16.0 * N * nslots); ntt_inv_launch(b); }

    ModDownArgs md{};
    md.xq = c1_; md.xp = c1_ + (size_t)nq * N; md.dst = c; md.mods_q = d_mods; md.mods_p = d_mods + nq;
    md.t = ModDownTables{d_md_qoverqiinvqi, d_md_qoverqimodp, d_md_vtimes, d_md_down};
    md.level = level; md.np = np; md.N = N; md.accumulate = accumulate ? 1 : 0; md.nbatch = 1;
    { ProfScope ps(this, PROF_MODDOWN, 8.0 * N * ((level + 1) * (accumulate ? 3.0 : 2.0) + np)); launch_moddown(md, s_); }
    MKHE_HIP(hipGetLastError());
}
void Context::external_product_hoisted(int level, const u64* ah, const u64* bg, u64* c, bool accumulate) {
    check_level(level);
    ext_core(level, ah, bg, c, accumulate);
}
void Context::external_product(int level, bool is_ntt, const u64* a, const u64* bg, u64* c, bool accumulate) {
    check_level(level);
    decompose(level, is_ntt, a, swk3_);
    ext_core(level, swk3_, bg, c, accumulate);
}

// ------------------------------------------------------------------ batched forms (one launch for all parties)
// Small launches of the small ring: forward sub-transforms + inner products in one kernel (ExtFusedArgs).  The limit is in limbs of the Decompose
// launch (vectors x digits x limb slots): beyond it the launch fills the chip and the one-pass H16-class kernel + the streaming inner product win.
bool Context::ext_fused_ok(int level, int nvec) const {
    static const int lim = MKHE_AB_INT("MKHE_EXT_FUSED_MAX", 150);
    if (alpha != 1 || logN != 14 || masked_ || is_bfv() || nvec < 1 || nvec > EXTF_MAX_V || mall > 64) return false;
    return nvec * beta(level) * nslots_qp(level) <= lim;
}
void Context::decompose_batch(int level, const std::vector<const u64*>& src, const std::vector<u64*>& dst, bool internal, bool stage_only) {
    check_level(level);
    const int nb = beta(level);
    if (stage_only && (alpha != 1 || !internal || logN != 14 || src.size() > (size_t)NTT_MAX_ITEMS)) throw Error("mkhe: internal: staged Decompose outside its shape");
    if (alpha != 1) {
        // alpha >= 2: CRT-reconstructed digits are spread in the coefficient domain first (DecomposeAndSplit,
        // basis_extension.go:428-535), then NTT'd in place (DecomposeSingleNTT, keyswitch.go:29-30)
        for (size_t base = 0; base < src.size(); base += DEC_MAX_ITEMS) {
            const int n = (int)std::min<size_t>(DEC_MAX_ITEMS, src.size() - base);
            DecompSpreadArgs da{};
            // N = 2^16 (round 3).  When every class part of the NTT launch runs on the H16 kernel, the spread kernel applies the first TWO stages
            // of the transform and the four 2^14-point sub-transforms of every limb are single passes of that kernel, in place: one read and one
            // write of every digit limb.  MKHE_SPREAD_RADIX4=0 is the earlier form: cross-half stage only, the spread digits staged in a separate
            // buffer (5.4 GB for the 16 operand components of an 8-party PN16QP1761 MulRelin) and the 2^15-point sub-transforms out of place from
            // there, two passes each, the second recomputing its cross stage from a second read of the source -- 6.3 GB moved per 2.1 GB Decompose
            // launch against 4.2 GB.
            static const int radix4_env = MKHE_AB_INT("MKHE_SPREAD_RADIX4", 1);
            const size_t item_words = (size_t)beta_max * mtot * N;
            bool h16 = false;
            if (logN == 16 && !masked_ && d_psi31) {
                NttBatch q{};                                  // the launch as ntt_fwd_launch will see it: does every class part run on H16?
                q.mods = d_mods; slots_qp(q, level); q.nouter = n * nb; q.prestaged = 1; q.psi31 = d_psi31; q.no_h16 = 0;
                h16 = ntt_fwd_prestaged_oop_ok(logN, q, small_q_.data());
            }
            bool radix4 = h16 && radix4_env && d_tb30 && d_tw30;
            // (what the H16 sub-transforms then load is below 22.2 p, not 4 p: a U-class modulus -- never reduced -- has to hold that input, 14 stages
            // of growth by 5 p and the 75 p bias below 2^62; the other moduli are brought to |x| < p at the load, Job::red)
            for (int s2 = 0; s2 < level + 1 + np && radix4; ++s2) {
                const int m = s2 <= level ? s2 : nq + (s2 - level - 1);
                if (((u_mods_ >> m) & 1) && (double)moduli[m] * (22.2 + 70 + 75) >= 4611686018427387904.0) radix4 = false;
            }
            const bool oop = h16 && !radix4;
            u64* stage = oop ? scratch(spreadbuf_, spreadbuf_words_, (size_t)n * item_words) : nullptr;
            for (int i = 0; i < n; ++i) { da.src[i] = src[base + i]; da.dst[i] = oop ? stage + (size_t)i * item_words : dst[base + i]; }
            da.mods = d_mods; da.map = map_qp(level); da.ta = d_dec_a; da.tb = d_dec_b; da.tc = d_dec_c;
            for (int d = 0; d < nb; ++d) {
                // decompLvl rule of DecomposeAndSplit (:437-441), digit index d
                const int dl = (level > alpha * (d + 1) - 1) ? alpha - 2 : (level % alpha) - 1;
                da.nd[d] = dl + 2;
            }
            da.alpha = alpha; da.ndigits = nb; da.nslots = level + 1 + np; da.mtot = mtot; da.N = N; da.nitems = n;
            // N = 2^16: the forward NTT always runs split; its cross-half stage is applied by the spread kernel itself
            da.psi = d_psi; da.first_stage = radix4 ? 2 : (logN == 16 && !masked_) ? 1 : 0;
            da.tb30 = d_tb30; da.tw30 = d_tw30;
            { ProfScope ps(this, PROF_SPREAD, 8.0 * N * n * ((level + 1) + (double)nb * da.nslots)); launch_decomp_spread(da, s_); }
            NttBatch b{};
            b.mods = d_mods; b.psi = d_psi; b.aux = d_inv_aux; slots_qp(b, level);
            b.src_outer = b.dst_outer = (long)mtot * N; b.src_inner = b.dst_inner = N; b.src_mapped = b.dst_mapped = 1;
            b.nitems = n; b.outers_per_item = nb;
            for (int i = 0; i < n; ++i) { b.src_items[i] = oop ? stage + (size_t)i * item_words : dst[base + i]; b.dst_items[i] = dst[base + i]; }
            b.nouter = n * nb; b.prestaged = da.first_stage; b.prestaged_oop = oop ? 1 : 0; b.skip_norm = internal ? 1 : 0;
            ntt_fwd_launch(b, false);
        }
        MKHE_HIP(hipGetLastError());
        return;
    }
    for (size_t base = 0; base < src.size(); base += NTT_MAX_ITEMS) {
        const int n = (int)std::min<size_t>(NTT_MAX_ITEMS, src.size() - base);
        NttBatch b{};
        b.mods = d_mods; b.psi = d_psi; b.aux = d_inv_aux; slots_qp(b, level);
        b.src_outer = N; b.src_inner = 0; b.src_mapped = 0;
        b.dst_outer = (long)mtot * N; b.dst_inner = N; b.dst_mapped = 1;
        b.reduce_in = 1; b.reduce_src_mod_is_outer = 1; b.skip_norm = internal ? 1 : 0;
        b.nitems = n; b.outers_per_item = nb;
        for (int i = 0; i < n; ++i) { b.src_items[i] = src[base + i]; b.dst_items[i] = dst[base + i]; }
        b.nouter = n * nb;
        if (stage_only) { ProfScope ps(this, PROF_NTT_DECOMP, 16.0 * N * b.nouter * b.nslots); launch_ntt_cross8_dec(b, logN, s_); }
        else ntt_fwd_launch(b, true);
        for (int i = 0; i < n; ++i) {
            const u64* dp = dst[base + i];
            staged_open_.erase(std::remove(staged_open_.begin(), staged_open_.end(), dp), staged_open_.end());
            if (stage_only) staged_open_.push_back(dp);
        }
    }
    MKHE_HIP(hipGetLastError());
}

// items: independent external products  dst (+)= ModDown( sum_i bg[i] (.) ah[i] )
// front half: inner products over the gadget digits + lazy inverse NTT of up to EXT_MAX_ITEMS items into c1 ([item][mtot][N])
int Context::ext_merge_members(int level) const {
    static const int on = MKHE_AB_INT("MKHE_EXT_MERGE", 1);
    if (!on || masked_ || np > 4) return 0;
    static_assert(VI_MAX == 4 && MD_VI_MAX == 4 && EXT_MAX_ITEMS == 64 && NTT_MAX_ITEMS == 64, "ExtMerge is sized for these");
    // members per virtual item: the padded slot list of the inverse launch has to fit, and the merged multSum adds
    // members * np 128-bit products before its one Montgomery fold
    int M = VI_MAX;
    if (M > 16 / np) M = 16 / np;
    if (M > (NTT_MAX_SLOTS - (level + 1)) / np) M = (NTT_MAX_SLOTS - (level + 1)) / np;
    return M;
}
bool Context::ext_plan_merge(int level, const ExtItem* it, int n, ExtMerge& mp) const {
    const int M = ext_merge_members(level);
    bool any = false;
    for (int i = 0; i < n; ++i) any = any || it[i].qadd != nullptr;
    if (M < 2 || n > 64 || (n < 2 && !any)) {
        if (any) throw Error("mkhe: internal: a batch with an NTT-domain summand has to be merged");
        return false;
    }
    mp = ExtMerge{};
    bool used[64] = {};
    for (int i = 0; i < n; ++i) {
        if (used[i]) continue;
        int first = 1;
        for (int k = i; k < n; ++k) {
            if (used[k] || it[k].dst != it[i].dst) continue;
            used[k] = true;
            // a later product of the group has to ADD onto the destination (anything else is not a sum: leave the batch alone)
            if (k != i && (!it[k].accumulate || it[k].addend || it[k].qadd)) {
                if (any) throw Error("mkhe: internal: a batch with an NTT-domain summand has to be merged");
                return false;
            }
            if (k == i && it[i].qadd && (it[i].accumulate || it[i].addend)) throw Error("mkhe: internal: qadd on an accumulating product");
            if (first || mp.cnt[mp.nvi - 1] == M) {
                const int v = mp.nvi++;
                mp.dst[v] = it[i].dst;
                mp.accumulate[v] = first ? (it[i].accumulate ? 1 : 0) : 1;
                mp.addend[v] = first ? it[i].addend : nullptr;
                mp.qadd[v] = first ? it[i].qadd : nullptr;
                mp.gal[v] = it[i].gal; mp.post[v] = it[i].post;      // (the destination's: every virtual item of the group carries them, the kernel applies post once)
                first = 0;
            }
            const int v = mp.nvi - 1;
            mp.mem[v][mp.cnt[v]++] = (unsigned char)k;
            if (mp.cnt[v] > 1) any = true;
            if (mp.cnt[v] > mp.members_max) mp.members_max = mp.cnt[v];
        }
    }
    return any;
}

// The product kernel of a batch whose F2 items come out of the Decompose NTT of the t_i itself (N = 2^15, ntt16_f2_kernel; ExtItem::f2_party).  Returns whether
// the products arrive in parts (f2_parts[i]: NttBatch::vi_parts of item i -- the inverse NTT adds them at its load).
bool Context::ext_front_f2(int level, const ExtItem* it, int n, u64* c1, ExtInnerArgs& ia, unsigned short* f2_parts) {
    const size_t item_words = (size_t)mtot * N;
    // N = 2^15: the F2 products come out of the Decompose NTT of the t_i itself (ntt16_f2_kernel).  The other items of the batch exist already
    // (step E computed by the F1 kernel: `pre`) or are plain products of stored digits -- the sharded finish, more parties than the F1 kernel's
    // forms take: the inner-product kernel computes those first and skips the F2 items (role 2: "computed elsewhere")
    bool others = false;
    for (int i = 0; i < n; ++i) {
        if (it[i].f2_party >= 0) { if (it[i].pre || it[i].ah2) throw Error("mkhe: internal: fused F2 products in a launch that cannot take them"); ia.pair[i] = 2; }      // (also undoes the pairing of a party's two products above: neither is computed here)
        else others = others || !(it[i].pre && !it[i].pre_src);
    }
    if (others) launch_ext_inner(ia, s_);
    const int np0 = (int)ext_f2_src_.size();
    const F2Sched& sc = f2_schedule(np0, level);
    F2FusedArgs fa{};
    fa.segs = sc.d_segs; fa.nwg = sc.nwg; fa.c1 = c1; fa.item_words = (long)item_words; fa.digit_stride = (long)item_words;
    for (int a = 0; a < np0; ++a) { fa.src[a] = ext_f2_src_[a]; fa.item_v[a] = fa.item_u[a] = -1; }
    int nf2 = 0;
    for (int i = 0; i < n; ++i) {
        const int a = it[i].f2_party;
        if (a < 0) continue;
        if (a >= np0) throw Error("mkhe: internal: fused F2 products in a launch that cannot take them");
        if (it[i].f2_key == 0) { fa.kv[a] = it[i].bg; fa.item_v[a] = i; fa.extra_v[a] = n + nf2 * (sc.parts - 1); }
        else { if (fa.ku && fa.ku != it[i].bg) throw Error("mkhe: internal: fused F2 products with more than one CRS"); fa.ku = it[i].bg; fa.item_u[a] = i; fa.extra_u[a] = n + nf2 * (sc.parts - 1); }
        f2_parts[i] = (unsigned short)(((n + nf2 * (sc.parts - 1)) << 8) | (sc.parts - 1));
        ++nf2;
    }
    for (int a = 0; a < np0; ++a) if (fa.item_v[a] < 0 || fa.item_u[a] < 0 || !fa.kv[a]) throw Error("mkhe: internal: a party without its two F2 products");
    if (n + nf2 * (sc.parts - 1) > 255) throw Error("mkhe: internal: too many product parts");
    fa.mods = d_mods; fa.psi = d_psi; fa.psi31 = d_psi31; fa.psi31n = d_psi31n; fa.u_mods = u_mods_;
    for (int m2 = 0; m2 < mall && m2 < NTT_MAX_SLOTS; ++m2) { if (small16_[m2]) fa.small_mask |= 1ull << m2; fa.sched[m2] = h16_sched_.empty() ? 15 : h16_sched_[m2]; }
    { NttBatch q{}; slots_qp(q, level); for (int s2 = 0; s2 < q.nslots; ++s2) fa.mod[s2] = q.mod[s2]; }
    fa.trace = ntt_trace;
    launch_ntt16_f2(fa, s_);
    return sc.parts > 1;
}

void Context::ext_front(int level, const ExtItem* it, int n, u64* c1, const ExtMerge* mp) {
    const int nb = beta(level), nslots = nslots_qp(level);
    const size_t item_words = (size_t)mtot * N;
    if (n < 1 || nslots < 1) return;
    ExtInnerArgs ia{};
    bool two = false, fused_inv = false, any_parts = false;
    unsigned short f2_parts[EXT_MAX_ITEMS] = {};
    bool xby = ext_xout_ != nullptr && n <= (ext_xout2_ ? 4 : 16);     // (five to sixteen single-gadget items: ext_inner_xwide_kernel)
    const bool xby2 = ext_xout2_ != nullptr;               // mkbfv: both gadgets carry their x
    for (int i = 0; i < n; ++i) {
        ia.ah[i] = it[i].ah; ia.bg[i] = it[i].bg;
        ia.ah2[i] = it[i].ah2; ia.bg2[i] = it[i].bg2;
        ia.xkey[i] = it[i].xkey; ia.xkey2[i] = it[i].xkey2;
        two = two || ia.ah2[i] != nullptr;
        xby = xby && it[i].xkey && it[i].bg == it[0].bg;
        if (xby2) xby = xby && it[i].ah2 && it[i].xkey2 && it[i].bg2 == it[0].bg2; else xby = xby && !it[i].ah2;
    }
    if ((ext_xout_ || ext_xout2_) && !xby) throw Error("mkhe: internal: x by-product requested for a batch that cannot carry it");
    ia.xout = xby ? ext_xout_ : nullptr; ia.xout2 = xby && xby2 ? ext_xout2_ : nullptr; ia.xmform = 1;
    const bool xy = !ext_ykeys_.empty() && ext_xmap_.empty(), xyb = !ext_ykeys_.empty() && !ext_xmap_.empty();
    const int ny = (int)ext_ykeys_.size();
    if (xy && (!xby || mp || n > (xby2 ? 4 : 8) || (n > 4 && ny != n) || ny < 1 || ny > 8 || (n <= 4 && ny > 4) || (int)ext_yh_.size() != ny ||
               (xby2 && ((int)ext_ykeys2_.size() != ny || (int)ext_yh2_.size() != ny)))) throw Error("mkhe: internal: y inside a launch that cannot compute it");
    int xgroups = 0;
    if (!ext_xmap_.empty()) {
        // B inputs' step F1 in one launch (mul_relin_batch): the items that share y_b are input b's, at most four, and carry x_b
        if (xby || two || mp) throw Error("mkhe: internal: per-input x by-products on a batch that cannot carry them");
        for (int i = 0; i < n; ++i) {
            u64* xo = nullptr; int cnt = 0;
            for (const auto& e : ext_xmap_) if (e.first == it[i].bg) xo = e.second;
            for (int k = 0; k < n; ++k) cnt += it[k].bg == it[i].bg;
            if (!xo || !it[i].xkey || cnt > 4) throw Error("mkhe: internal: per-input x by-products on a batch that cannot carry them");
            ia.xkey2[i] = xo;
            bool first = true;
            for (int k = 0; k < i; ++k) first = first && it[k].bg != it[i].bg;
            xgroups += first;
        }
        ia.xout = ext_xmap_[0].second; ia.xmulti = 1;
        xby = true;
    }
    // keys that a single item reads (v_i, rotation keys) are streamed; x, y, u are shared by several items and stay cached
    for (int i = 0; i < n; ++i) {
        int uses = 0;
        for (int k = 0; k < n; ++k) uses += (it[k].bg == it[i].bg);
        ia.bg_once[i] = uses == 1 ? 1 : 0;
    }
    // neighbours that share their digits (step F: <h(t_i), v_i> and <h(t_i), u>) are computed together
    for (int i = 0; i + 1 < n && !xby; ++i)
        if (!ia.pair[i] && ia.ah[i] == ia.ah[i + 1] && !ia.ah2[i] && !ia.ah2[i + 1]) { ia.pair[i] = 1; ia.pair[i + 1] = 2; ++i; }
    for (int i = 0; i < n; ++i) if (it[i].pre) {      // (role 2 = computed elsewhere, in the slot: the kernels skip it; role 3 = computed elsewhere, at pre_src: they copy it)
        if (ia.pair[i]) throw Error("mkhe: internal: a precomputed item inside a pair");
        if (it[i].pre_src) { ia.pair[i] = 3; ia.bg[i] = it[i].pre_src; ia.bg_once[i] = 1; } else ia.pair[i] = 2;
    }
    ia.c1 = c1; ia.mods = d_mods; ia.map = map_qp(level); ia.digit_stride = (long)item_words; ia.c1_item = (long)item_words;
    ia.nitems = n; ia.nb = nb; ia.nslots = nslots; ia.N = N;
    // algorithmic bytes: every DISTINCT digit / key array once (items that share x, y or the CRS u are computed by one thread that loads the
    // shared operand once per coefficient, ext_inner_group_kernel), one output limb per item, and the x by-product's keys and result
    (void)two;
    int distinct = 0;
    {
        const u64* seen[4 * EXT_MAX_ITEMS]; int ns = 0;
        auto add = [&](const u64* p) { if (!p) return; for (int k = 0; k < ns; ++k) if (seen[k] == p) return; seen[ns++] = p; };
        for (int i = 0; i < n; ++i) if (!it[i].pre) { add(ia.ah[i]); add(ia.bg[i]); add(ia.ah2[i]); add(ia.bg2[i]); }
        distinct = ns;
    }
    // (the fused F2 launch: every t_i limb once, every key once, the parts of the products out)
    const bool f2 = !ext_f2_src_.empty();
    const double f2_bytes = f2 ? 8.0 * N * ((double)ext_f2_src_.size() * nb + (double)nb * nslots * (ext_f2_src_.size() + 1.0) + 2.0 * ext_f2_src_.size() * nslots * f2_schedule((int)ext_f2_src_.size(), level).parts) : 0.0;
    { ProfScope ps(this, f2 ? PROF_NTT_F2 : PROF_EXT_INNER, f2 ? f2_bytes : 8.0 * N * nslots * ((double)nb * distinct + n + (xby ? nb * (n + (xgroups ? xgroups : 1.0)) * (xby2 ? 2 : 1) : 0.0) + (xy ? nb * (2.0 * ny - 1.0) : 0.0) + (xyb ? nb * (1.0 * n + ext_ykeys_.size() - xgroups) : 0.0)));
      if (ext_staged_.empty()) {
          for (int i = 0; i < n; ++i)
              if (!it[i].pre && std::find(staged_open_.begin(), staged_open_.end(), it[i].ah) != staged_open_.end())
                  throw Error("mkhe: internal: digits left after the cross stages read as a full transform");
      }
      if (!ext_f2_src_.empty()) {
          if (xby || xy || xyb || !mp || !ext_staged_.empty()) throw Error("mkhe: internal: fused F2 products in a launch that cannot take them");
          any_parts = ext_front_f2(level, it, n, c1, ia, f2_parts);
      } else if (!ext_staged_.empty()) {
          // the items' digit vectors were left after the cross stages (decompose_batch, stage_only): sub-transforms and products in one kernel
          if (xby || xy || xyb || two || (int)ext_staged_.size() > EXTF_MAX_V) throw Error("mkhe: internal: staged digits in a launch that cannot take them");
          ExtFusedArgs fa{};
          fa.nv = (int)ext_staged_.size();
          for (int v = 0; v < fa.nv; ++v) fa.stage[v] = ext_staged_[v];
          for (int v = 0; v < fa.nv; ++v) staged_open_.erase(std::remove(staged_open_.begin(), staged_open_.end(), ext_staged_[v]), staged_open_.end());
          for (int i = 0; i < n; ++i) {
              if (it[i].pre && !it[i].pre_src) continue;                 // (computed before, in its slot)
              int v = -1;
              for (int k = 0; k < fa.nv; ++k) if (ext_staged_[k] == it[i].ah) v = k;
              if (v < 0 || it[i].pre || fa.nk[v] >= 2) throw Error("mkhe: internal: staged digits in a launch that cannot take them");
              fa.bg[v][fa.nk[v]] = it[i].bg; fa.out[v][fa.nk[v]] = c1 + (size_t)i * item_words; ++fa.nk[v];
          }
          for (int v = 0; v < fa.nv; ++v) if (!fa.nk[v]) throw Error("mkhe: internal: a staged digit vector without a product");
          fa.mods = d_mods; fa.map = map_qp(level); fa.psi = d_psi; fa.digit_stride = (long)item_words;
          for (int m2 = 0; m2 < mall && m2 < 64; ++m2) if (small_q_[m2]) fa.small_mask |= 1ull << m2;
          fa.nb = nb; fa.nslots = nslots; fa.N = N; fa.logN = logN;
          // every product of the launch made here, no NTT-domain summand (rotations, conjugations): the inverse sub-transforms as well
          static const int inv_env = MKHE_AB_INT("MKHE_EXT_FUSED_INV", 1);
          fused_inv = inv_env != 0;
          for (int i = 0; i < n; ++i) fused_inv = fused_inv && !it[i].pre && !it[i].qadd;
          if (fused_inv) { fa.inv = 1; fa.psiinv = d_psiinv; fa.aux = d_inv_aux; }
          launch_ext_fused_lds(fa, s_);
      } else if (xy && n > 4) {
          ExtXyWideArgs xa{};
          for (int j = 0; j < n; ++j) { xa.ah[j] = it[j].ah; xa.xkey[j] = it[j].xkey; xa.ykey[j] = ext_ykeys_[j]; xa.yh[j] = ext_yh_[j]; }
          xa.xout = ext_e_slot_ >= 0 ? nullptr : ext_xout_; xa.e_out = ext_e_slot_ >= 0 ? c1 + (size_t)ext_e_slot_ * item_words : nullptr; xa.c1 = c1;
          xa.mods = d_mods; xa.map = map_qp(level); xa.digit_stride = (long)item_words; xa.c1_item = (long)item_words;
          xa.g = n; xa.nb = nb; xa.nslots = nslots; xa.N = N;
          launch_ext_inner_xy_wide(xa, s_);
      } else if (xy) {
          ExtXyArgs xa{};
          for (int j = 0; j < n; ++j) { xa.ah[j] = it[j].ah; xa.xkey[j] = it[j].xkey; }
          for (int j = 0; j < ny; ++j) { xa.ykey[j] = ext_ykeys_[j]; xa.yh[j] = ext_yh_[j]; }
          if (xby2) { for (int j = 0; j < n; ++j) { xa.ah2[j] = it[j].ah2; xa.xkey2[j] = it[j].xkey2; } for (int j = 0; j < ny; ++j) { xa.ykey2[j] = ext_ykeys2_[j]; xa.yh2[j] = ext_yh2_[j]; } }
          xa.xout2 = (xby2 && ext_e_slot_ < 0) ? ext_xout2_ : nullptr;
          xa.xout = ext_e_slot_ >= 0 ? nullptr : ext_xout_; xa.e_out = ext_e_slot_ >= 0 ? c1 + (size_t)ext_e_slot_ * item_words : nullptr; xa.c1 = c1; xa.mods = d_mods; xa.map = map_qp(level); xa.digit_stride = (long)item_words; xa.c1_item = (long)item_words;
          xa.g = n; xa.g1 = ny; xa.nb = nb; xa.nslots = nslots; xa.N = N;
          launch_ext_inner_xy(xa, s_);
      } else if (xyb) {
          // B inputs' step F1 with x_b and y_b in the thread: the items come input by input (g per input, mul_relin_batch), up to XYB_MAX inputs per launch
          const int nin = (int)ext_xmap_.size(), g = nin ? n / nin : 0, g1 = (int)ext_ykeys_.size();
          if (two || mp || g < 1 || g > 4 || g1 < 1 || g1 > 4 || nin * g != n || (int)ext_yh_.size() != nin * g1 || (!ext_eouts_.empty() && (int)ext_eouts_.size() != nin)) throw Error("mkhe: internal: per-input y on a batch that cannot carry it");
          for (int b0 = 0; b0 < nin; b0 += XYB_MAX) {
              ExtXyBatchArgs xa{};
              const int cnt = std::min(XYB_MAX, nin - b0);
              for (int b = 0; b < cnt; ++b) {
                  for (int j = 0; j < g; ++j) {
                      const ExtItem& e = it[(b0 + b) * g + j];
                      if (e.bg != ext_xmap_[b0 + b].first || !e.xkey) throw Error("mkhe: internal: per-input y on a batch that cannot carry it");
                      xa.ah[b][j] = e.ah;
                      if (b == 0) xa.xkey[j] = e.xkey; else if (e.xkey != xa.xkey[j]) throw Error("mkhe: internal: per-input y on a batch that cannot carry it");
                  }
                  for (int j = 0; j < g1; ++j) xa.yh[b][j] = ext_yh_[(b0 + b) * g1 + j];
                  xa.xout[b] = ext_xmap_[b0 + b].second;
                  if (!ext_eouts_.empty()) xa.eout[b] = ext_eouts_[b0 + b];
              }
              for (int j = 0; j < g1; ++j) xa.ykey[j] = ext_ykeys_[j];
              xa.c1 = c1 + (size_t)b0 * g * item_words; xa.mods = d_mods; xa.map = map_qp(level); xa.digit_stride = (long)item_words; xa.c1_item = (long)item_words;
              xa.g = g; xa.g1 = g1; xa.nbatch = cnt; xa.nb = nb; xa.nslots = nslots; xa.N = N;
              launch_ext_inner_xy_batch(xa, s_);
          }
      } else launch_ext_inner(ia, s_); }
    NttBatch b{};
    b.src = c1; b.dst = c1; b.mods = d_mods; b.psi = d_psiinv; b.aux = d_inv_aux;
    b.src_inner = b.dst_inner = N; b.src_mapped = b.dst_mapped = 1;
    b.src_outer = b.dst_outer = (long)item_words; b.lazy_out = 1;
    if (mp) {
        // merged: one job per (virtual item, Q limb) -- the members' limbs are added up at the load -- and one per (member, P limb)
        b.vi = 1; b.vi_q = level + 1; b.vi_np = np; b.nouter = mp->nvi;
        b.nslots = level + 1 + mp->members_max * np;
        for (int j = 0; j <= level; ++j) { b.mod[j] = j; b.pos[j] = j; }
        for (int k = 0; k < mp->members_max; ++k)
            for (int j = 0; j < np; ++j) { b.mod[level + 1 + k * np + j] = nq + j; b.pos[level + 1 + k * np + j] = level + 1 + j; }
        for (int v = 0; v < mp->nvi; ++v) {
            b.vi_cnt[v] = mp->cnt[v]; b.vi_extra[v] = mp->qadd[v];
            for (int k = 0; k < VI_MAX; ++k) b.vi_mem[v] |= (unsigned)mp->mem[v][k] << (8 * k);
            int nsum = mp->qadd[v] ? 1 : 0;
            for (int k = 0; k < mp->cnt[v]; ++k) nsum += 1 + (f2_parts[mp->mem[v][k]] & 255);
            if (nsum > VI_SUMS) throw Error("mkhe: internal: more summands than an inverse job adds up at its load");
        }
        if (any_parts) for (int i = 0; i < n; ++i) b.vi_parts[i] = f2_parts[i];
        b.vi_jobs = mp->nvi * (level + 1) + n * np;
        { ProfScope ps(this, PROF_NTT_INV, 16.0 * N * b.vi_jobs); if (fused_inv) launch_ntt_inv_cross8_sum(b, logN, s_); else ntt_inv_launch(b); }
        return;
    }
    slots_qp(b, level);
    b.nouter = n;
    { ProfScope ps(this, PROF_NTT_INV, 16.0 * N * n * nslots); if (fused_inv) launch_ntt_inv_cross8_sum(b, logN, s_); else ntt_inv_launch(b); }
}
// back half: ModDown of the items in c1 into (or onto) their destinations
void Context::ext_back(int level, const ExtItem* it, int n, const u64* c1, u64 galEl, const ExtMerge* mp) {
    if (n < 1) return;
    const size_t item_words = (size_t)mtot * N;
    if (mp) {
        ModDownMergedArgs md{};
        md.c1 = c1; md.mods_q = d_mods; md.mods_p = d_mods + nq;
        md.t = ModDownTables{d_md_qoverqiinvqi, d_md_qoverqimodp, d_md_vtimes, d_md_down};
        md.c1_item = (long)item_words; md.p_offset = (long)nq * N; md.nvi = mp->nvi; md.level = level; md.np = np; md.N = N;
        double bytes = 0;
        for (int v = 0; v < mp->nvi; ++v) {
            md.dst[v] = mp->dst[v]; md.accumulate[v] = mp->accumulate[v]; md.addend[v] = mp->addend[v]; md.cnt[v] = mp->cnt[v];
            md.gal_v[v] = mp->gal[v]; md.post[v] = mp->post[v];
            for (int k = 0; k < MD_VI_MAX; ++k) md.mem[v] |= (unsigned)mp->mem[v][k] << (8 * k);
            bytes += 8.0 * N * ((level + 1) * (mp->accumulate[v] ? 3.0 : 2.0) + np * mp->cnt[v]);
        }
        md.galEl = galEl; md.logN = logN;
        bool any_gal = galEl != 0;
        for (int v = 0; v < mp->nvi; ++v) any_gal = any_gal || mp->gal[v] || mp->post[v];
        if (!rs_maps_.empty() && !any_gal && level >= 1) {
            // fused Rescale (mul_relin_rescale, mul_relin_batch): possible when this launch is the only writer of every polynomial of the products it
            // touches -- every destination inside a registered product, on a polynomial boundary, written once, all polynomials of the product present
            const size_t PF = (size_t)(level + 1) * N;
            bool ok = true;
            std::vector<int> which(mp->nvi, -1), cover(rs_maps_.size(), 0);
            for (int v = 0; v < mp->nvi && ok; ++v) {
                ok = !mp->accumulate[v] && !mp->addend[v];
                for (size_t m = 0; m < rs_maps_.size() && ok && which[v] < 0; ++m) {
                    const RsMap& r = rs_maps_[m];
                    if (!r.done && r.out_limbs == level && mp->dst[v] >= r.full && mp->dst[v] < r.full + (size_t)r.npolys * PF &&
                        (size_t)(mp->dst[v] - r.full) % PF == 0) which[v] = (int)m;
                }
                ok = ok && which[v] >= 0;
                for (int w = 0; w < v && ok; ++w) ok = mp->dst[w] != mp->dst[v];
                if (ok) ++cover[which[v]];
            }
            for (size_t m = 0; m < rs_maps_.size() && ok; ++m) ok = cover[m] == 0 || cover[m] == rs_maps_[m].npolys;
            if (ok) {
                md.rescale_row = d_rescale + (size_t)(level - 1) * nq;
                md.rescale_h = md.rescale_row + (size_t)nq * nq;
                for (int v = 0; v < mp->nvi; ++v) {
                    RsMap& r = rs_maps_[which[v]];
                    md.rdst[v] = r.out + (size_t)(mp->dst[v] - r.full) / PF * ((size_t)level * N);
                    r.done = true;
                }
            }
        }
        { ProfScope ps(this, PROF_MODDOWN, bytes); launch_moddown_merged(md, s_); }
        return;
    }
    ModDownBatchArgs md{};
    md.c1 = c1; md.mods_q = d_mods; md.mods_p = d_mods + nq;
    md.t = ModDownTables{d_md_qoverqiinvqi, d_md_qoverqimodp, d_md_vtimes, d_md_down};
    md.c1_item = (long)item_words; md.p_offset = (long)nq * N; md.nitems = n; md.level = level; md.np = np; md.N = N;
    if (masked_) { md.qlist = d_ownq; md.nqlist = nq_owned(level); }
    double bytes = 0;
    for (int i = 0; i < n; ++i) {
        md.dst[i] = it[i].dst; md.accumulate[i] = it[i].accumulate ? 1 : 0; md.addend[i] = it[i].addend;
        md.gal_v[i] = it[i].gal; md.post[i] = it[i].post;
        bytes += 8.0 * N * ((level + 1) * (it[i].accumulate ? 3.0 : 2.0) + np);
    }
    md.galEl = galEl; md.logN = logN;
    { ProfScope ps(this, PROF_MODDOWN, bytes); launch_moddown_batch(md, s_); }
}
void Context::ext_batch(int level, const std::vector<ExtItem>& items, int join_before_moddown, int stage, u64 galEl) {
    if (galEl && items.size() > (size_t)EXT_MAX_ITEMS) throw Error("mkhe: too many external products for a fused rotation");
    if (stage != 0 && items.size() > (size_t)EXT_MAX_ITEMS) throw Error("mkhe: too many external products for a staged batch");
    check_level(level);
    const size_t item_words = (size_t)mtot * N;
    for (size_t base = 0; base < items.size(); base += EXT_MAX_ITEMS) {
        const int n = (int)std::min<size_t>(EXT_MAX_ITEMS, items.size() - base);
        int extra = 0;
        if (!ext_f2_src_.empty()) {
            int nf2 = 0;
            for (int i = 0; i < n; ++i) nf2 += items[base + i].f2_party >= 0;
            extra = nf2 * (f2_schedule((int)ext_f2_src_.size(), level).parts - 1);
        }
        u64* c1 = scratch(c1b_, c1b_words_, (size_t)(n + extra) * item_words);
        ExtMerge mp;
        const bool merged = stage == 0 && ext_plan_merge(level, items.data() + base, n, mp);
        if (stage != 2) ext_front(level, items.data() + base, n, c1, merged ? &mp : nullptr);
        if (stage == 1) continue;
        if (join_before_moddown >= 0) { join_side(join_before_moddown); join_before_moddown = -1; }
        ext_back(level, items.data() + base, n, c1, galEl, merged ? &mp : nullptr);
    }
    if (join_before_moddown >= 0) join_side(join_before_moddown);
    MKHE_HIP(hipGetLastError());
}

}  // namespace mkhe
