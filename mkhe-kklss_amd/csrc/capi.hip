// capi.hip -- extern "C" boundary (include/mkhe.h) over mkhe::Context.
#include "../../include/mkhe.h"
#include "engine.h"
#include <string>
#include <vector>

using namespace mkhe;

struct mkhe_ctx { Context* c; };
// handles created in one call (mkhe_ct_create_batch / mkhe_swk_create_batch) are views into ONE pooled block, which goes back to the pool when the last
// of them is destroyed -- behind the uses of all of them
struct mkhe_block { u64* base; size_t words; int refs; HandleUsers users; };
struct mkhe_swk { Swk s; mutable HandleUsers users; mkhe_block* blk = nullptr; };
struct mkhe_ct { Ct c; mutable HandleUsers users; mkhe_block* blk = nullptr; };
static mkhe_block* block_new(Context* c, size_t words, int refs) {
    mkhe_block* b = new mkhe_block();
    b->words = words; b->refs = refs; b->base = nullptr;
    try { b->base = c->pool_alloc(words); } catch (...) { delete b; throw; }
    return b;
}
static void block_release(mkhe_ctx* ctx, mkhe_block* b, const HandleUsers& u) {
    for (const auto& e : u.v) {
        bool found = false;
        for (auto& f : b->users.v) if (f.first == e.first) { if (e.second > f.second) f.second = e.second; found = true; }
        if (!found) b->users.v.push_back(e);
    }
    b->users.exposed = b->users.exposed || u.exposed;
    if (--b->refs > 0) return;
    if (ctx) ctx->c->pool_free(b->base, b->words, &b->users); else (void)hipFree(b->base);
    delete b;
}
struct mkhe_graph { hipGraphExec_t exec; };

static thread_local std::string g_err;

// An exception may leave the context in the middle of a forked chain (active stream still the side stream, a side chain that
// writes into the caller's output never joined, a MulAndRelin plan half executed): recover() puts the context back into its
// idle state and drains both streams before the error is reported, so that the caller may free its handles safely.
static thread_local Context* g_last_ctx = nullptr;
// g_last_ctx is reset first, so that a call that fails before it reaches need() -- mkhe_ctx_create, a null-argument check -- never drains or
// invalidates the plan of whichever context this thread happened to use last (which may even be gone).
#define MKHE_TRY(body) g_last_ctx = nullptr; try { body; if (g_last_ctx) g_last_ctx->mark_enqueued(); return 0; } \
    catch (const std::exception& e) { g_err = e.what(); if (g_last_ctx) g_last_ctx->recover(); return 1; } \
    catch (...) { g_err = "mkhe: unknown error"; if (g_last_ctx) g_last_ctx->recover(); return 1; }

// every entry point goes through this: a null context is an error of the caller, reported like any other
static Context* need(const mkhe_ctx* ctx) {
    if (!ctx || !ctx->c) { g_last_ctx = nullptr; throw Error("mkhe: null context"); }
    g_last_ctx = ctx->c;
    ctx->c->touch();                 // anything this call enqueues is counted (Context::seq_, the pool's ordering clock)
    return ctx->c;
}

// Every entry point names the handles it is about to enqueue work on (Context::note_use: the pool of whichever context the buffer
// is freed into later orders itself behind exactly these uses, csrc/engine.h).
static void mark1(Context* c, const mkhe_ct* h) { if (h) c->note_use(h->users); }
static void mark1(Context* c, const mkhe_swk* h) { if (h) c->note_use(h->users); }
template <class... H> static void mark(const mkhe_ctx* ctx, H... hs) { if (ctx && ctx->c) { (mark1(ctx->c, hs), ...); } }

static std::vector<const Swk*> swk_list(const mkhe_ctx* ctx, const mkhe_swk* const* v, int n) {
    std::vector<const Swk*> r;
    if (!v) return r;
    r.resize(n);
    for (int i = 0; i < n; ++i) { if (!v[i]) throw Error("mkhe: null key handle in a per-party key list"); mark(ctx, v[i]); r[i] = &v[i]->s; }
    return r;
}

extern "C" {

const char* mkhe_last_error(void) { return g_err.c_str(); }

int mkhe_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

int mkhe_ctx_create(mkhe_ctx** out, int logN, const uint64_t* Q, int nQ, const uint64_t* P, int nP,
                    int gamma, const uint64_t* psiQ, const uint64_t* psiP, int device) {
    MKHE_TRY({
        if (!out || !Q || !P) throw Error("mkhe_ctx_create: null argument");
        *out = new mkhe_ctx{new Context(logN, Q, nQ, P, nP, gamma, psiQ, psiP, device)};
    })
}
void mkhe_ctx_destroy(mkhe_ctx* ctx) { if (ctx) { if (g_last_ctx == ctx->c) g_last_ctx = nullptr; delete ctx->c; delete ctx; } }
int mkhe_ctx_sync(mkhe_ctx* ctx) { MKHE_TRY(need(ctx)->sync()) }
int mkhe_capture_begin(mkhe_ctx* ctx) {
    MKHE_TRY({
        MKHE_HIP(hipSetDevice(need(ctx)->device));
        // the engine captures several streams (side stream, forked contexts) that join each other in both directions; the
        // HIP runtime of ROCm 7.0 (e.g. the one bundled with PyTorch 2.10, which a process that imported torch first
        // binds to) recurses without end in hipStreamEndCapture on such a capture.  Refuse instead of crashing.
        int rv = 0;
        MKHE_HIP(hipRuntimeGetVersion(&rv));
        if (rv < 70200000) throw Error("mkhe_capture_begin: the HIP runtime loaded in this process (version " + std::to_string(rv) +
                                       ") cannot end a multi-stream capture; graph capture needs the ROCm >= 7.2 runtime");
        MKHE_HIP(hipStreamBeginCapture(need(ctx)->stream, hipStreamCaptureModeRelaxed));
    })
}
int mkhe_capture_end(mkhe_ctx* ctx, mkhe_graph** out) {
    MKHE_TRY({
        if (!out) throw Error("mkhe_capture_end: null argument");
        hipGraph_t g = nullptr;
        MKHE_HIP(hipStreamEndCapture(need(ctx)->stream, &g));
        hipGraphExec_t ex = nullptr;
        const hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) throw Error(std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        *out = new mkhe_graph{ex};
    })
}
int mkhe_graph_launch(mkhe_ctx* ctx, mkhe_graph* graph) {
    MKHE_TRY({ if (!graph) throw Error("mkhe_graph_launch: null argument"); MKHE_HIP(hipGraphLaunch(graph->exec, need(ctx)->stream)); })
}
void mkhe_graph_destroy(mkhe_graph* graph) { if (graph) { (void)hipGraphExecDestroy(graph->exec); delete graph; } }
int mkhe_ctx_wait_for(mkhe_ctx* ctx, mkhe_ctx* other) {
    MKHE_TRY({ if (!ctx || !other) throw Error("mkhe_ctx_wait_for: null argument"); need(ctx)->wait_for(*need(other)); })
}
int mkhe_ctx_alpha(const mkhe_ctx* ctx) { if (!ctx) return 0; return ctx->c->alpha; }
int mkhe_ctx_beta(const mkhe_ctx* ctx, int level) { if (!ctx) return 0; return ctx->c->beta(level); }
int mkhe_ctx_n(const mkhe_ctx* ctx) { if (!ctx) return 0; return ctx->c->N; }
size_t mkhe_ctx_swk_words(const mkhe_ctx* ctx) { if (!ctx) return 0; return ctx->c->swk_words(); }
uint64_t mkhe_ctx_psi(const mkhe_ctx* ctx, int i) { if (!ctx) return 0; return (i >= 0 && i < ctx->c->mall) ? ctx->c->psi_plain[i] : 0; }
void* mkhe_ctx_stream(mkhe_ctx* ctx) { if (!ctx) return 0; ctx->c->mark_external(); return (void*)ctx->c->stream; }

// ---- switching keys
static void swk_create(mkhe_ctx* ctx, mkhe_swk** out, bool zero) {
    Context* c = need(ctx);
    MKHE_HIP(hipSetDevice(c->device));
    if (!out) throw Error("mkhe_swk_create: null argument");
    mkhe_swk* s = new mkhe_swk();
    try { s->s.d = c->pool_alloc(c->swk_words()); } catch (...) { delete s; throw; }
    if (zero) { c->note_use(s->users); MKHE_HIP(hipMemsetAsync(s->s.d, 0, c->swk_words() * sizeof(u64), c->stream)); }
    *out = s;
}
int mkhe_swk_create(mkhe_ctx* ctx, mkhe_swk** out) { MKHE_TRY(swk_create(ctx, out, true)) }
int mkhe_swk_create_uninit(mkhe_ctx* ctx, mkhe_swk** out) { MKHE_TRY(swk_create(ctx, out, false)) }
void mkhe_swk_destroy(mkhe_ctx* ctx, mkhe_swk* swk) {
    if (!swk) return;
    if (swk->blk) block_release(ctx, swk->blk, swk->users);
    else if (swk->s.d && swk->s.owned) { if (ctx) ctx->c->pool_free(swk->s.d, ctx->c->swk_words(), &swk->users); else (void)hipFree(swk->s.d); }
    delete swk;
}
int mkhe_swk_create_batch(mkhe_ctx* ctx, int count, mkhe_swk** out) {
    MKHE_TRY({
        Context* c = need(ctx);
        if (!out || count < 1) throw Error("mkhe_swk_create_batch: bad argument");
        MKHE_HIP(hipSetDevice(c->device));
        mkhe_block* b = block_new(c, (size_t)count * c->swk_words(), count);
        for (int i = 0; i < count; ++i) { mkhe_swk* s = new mkhe_swk(); s->s.d = b->base + (size_t)i * c->swk_words(); s->s.owned = false; s->blk = b; out[i] = s; }
    })
}
void mkhe_swk_destroy_batch(mkhe_ctx* ctx, int count, mkhe_swk* const* swks) {
    if (!swks) return;
    for (int i = 0; i < count; ++i) mkhe_swk_destroy(ctx, swks[i]);
}
int mkhe_swk_upload(mkhe_ctx* ctx, mkhe_swk* swk, const uint64_t* host) {
    MKHE_TRY({ mark(ctx, swk);
        Context* c = need(ctx);
        if (!swk || !host) throw Error("mkhe_swk_upload: null argument");
        MKHE_HIP(hipMemcpyAsync(swk->s.d, host, c->swk_words() * sizeof(u64), hipMemcpyHostToDevice, c->stream));
        c->sync();
    })
}
int mkhe_swk_upload_limbs(mkhe_ctx* ctx, mkhe_swk* swk, const uint64_t* const* limbs, int ndigits) {
    MKHE_TRY({ mark(ctx, swk);
        Context* c = need(ctx);
        if (!swk || !limbs) throw Error("mkhe_swk_upload_limbs: null argument");
        if (ndigits < 0 || ndigits > c->beta_max) throw Error("mkhe_swk_upload_limbs: bad digit count");
        for (int i = 0; i < ndigits * c->mtot; ++i)
            MKHE_HIP(hipMemcpyAsync(swk->s.d + (size_t)i * c->N, limbs[i], (size_t)c->N * sizeof(u64), hipMemcpyHostToDevice, c->stream));
        c->sync();
    })
}
int mkhe_swk_download(mkhe_ctx* ctx, const mkhe_swk* swk, uint64_t* host) {
    MKHE_TRY({ mark(ctx, swk);
        Context* c = need(ctx);
        if (!swk || !host) throw Error("mkhe_swk_download: null argument");
        MKHE_HIP(hipMemcpyAsync(host, swk->s.d, c->swk_words() * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        c->sync();
    })
}
void* mkhe_swk_devptr(mkhe_swk* swk) { if (swk) swk->users.exposed = true; return swk ? swk->s.d : nullptr; }

// ---- ciphertexts
static void ct_create(mkhe_ctx* ctx, int n, const int* ids, int limbs, bool zero, mkhe_ct** out);
int mkhe_ct_create(mkhe_ctx* ctx, int n, const int* ids, int limbs, mkhe_ct** out) {
    MKHE_TRY(ct_create(ctx, n, ids, limbs, true, out))
}
int mkhe_ct_create_uninit(mkhe_ctx* ctx, int n, const int* ids, int limbs, mkhe_ct** out) {
    MKHE_TRY(ct_create(ctx, n, ids, limbs, false, out))
}
} // extern "C"
static void ct_create(mkhe_ctx* ctx, int n, const int* ids, int limbs, bool zero, mkhe_ct** out) {
    {
        Context* c = need(ctx);
        if (!out || (n > 0 && !ids)) throw Error("mkhe_ct_create: null argument");
        if (n < 0 || n > 32 || limbs < 1 || limbs > c->nq) throw Error("mkhe_ct_create: bad shape");
        MKHE_HIP(hipSetDevice(c->device));
        mkhe_ct* t = new mkhe_ct();
        t->c.n = n; t->c.limbs = limbs; t->c.ids.assign(ids, ids + n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) if (ids[i] == ids[j]) { delete t; throw Error("mkhe_ct_create: repeated id"); }
        const size_t w = (size_t)(1 + n) * limbs * c->N;
        try { t->c.d = c->pool_alloc(w); } catch (...) { delete t; throw; }
        if (zero) { c->note_use(t->users); MKHE_HIP(hipMemsetAsync(t->c.d, 0, w * sizeof(u64), c->stream)); }
        *out = t;
    }
}
extern "C" {
void mkhe_ct_destroy(mkhe_ctx* ctx, mkhe_ct* ct) {
    if (!ct) return;
    if (ct->blk) block_release(ctx, ct->blk, ct->users);
    else if (ct->c.d) { if (ctx) ctx->c->pool_free(ct->c.d, (size_t)(1 + ct->c.n) * ct->c.limbs * ctx->c->N, &ct->users); else (void)hipFree(ct->c.d); }
    delete ct;
}
int mkhe_ct_create_batch(mkhe_ctx* ctx, int nbatch, int n, const int* ids, int limbs, mkhe_ct** out) {
    MKHE_TRY({
        Context* c = need(ctx);
        if (!out || nbatch < 1 || (n > 0 && !ids)) throw Error("mkhe_ct_create_batch: bad argument");
        if (n < 0 || n > 32 || limbs < 1 || limbs > c->nq) throw Error("mkhe_ct_create_batch: bad shape");
        for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) if (ids[i] == ids[j]) throw Error("mkhe_ct_create_batch: repeated id");
        MKHE_HIP(hipSetDevice(c->device));
        const size_t w = (size_t)(1 + n) * limbs * c->N;
        mkhe_block* b = block_new(c, (size_t)nbatch * w, nbatch);
        for (int k = 0; k < nbatch; ++k) {
            mkhe_ct* t = new mkhe_ct();
            t->c.n = n; t->c.limbs = limbs; t->c.ids.assign(ids, ids + n); t->c.d = b->base + (size_t)k * w; t->blk = b;
            out[k] = t;
        }
    })
}
void mkhe_ct_destroy_batch(mkhe_ctx* ctx, int nbatch, mkhe_ct* const* cts) {
    if (!cts) return;
    for (int k = 0; k < nbatch; ++k) mkhe_ct_destroy(ctx, cts[k]);
}
int mkhe_ct_upload(mkhe_ctx* ctx, mkhe_ct* ct, const uint64_t* host) {
    MKHE_TRY({ mark(ctx, ct);
        Context* c = need(ctx);
        if (!ct || !host) throw Error("mkhe_ct_upload: null argument");
        const size_t w = (size_t)(1 + ct->c.n) * ct->c.limbs * c->N;
        MKHE_HIP(hipMemcpyAsync(ct->c.d, host, w * sizeof(u64), hipMemcpyHostToDevice, c->stream));
        c->sync();
    })
}
int mkhe_ct_copy(mkhe_ctx* ctx, const mkhe_ct* in, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, in, out);
        if (!in || !out) throw Error("mkhe_ct_copy: null argument");
        if (in->c.n != out->c.n || in->c.limbs != out->c.limbs || in->c.ids != out->c.ids) throw Error("mkhe_ct_copy: shapes differ");
        const size_t w = (size_t)(1 + in->c.n) * in->c.limbs * need(ctx)->N;
        MKHE_HIP(hipMemcpyAsync(out->c.d, in->c.d, w * sizeof(u64), hipMemcpyDeviceToDevice, need(ctx)->stream));
    })
}
int mkhe_ct_upload_poly_limbs(mkhe_ctx* ctx, mkhe_ct* ct, int slot, const uint64_t* const* limbs) {
    MKHE_TRY({ mark(ctx, ct);
        Context* c = need(ctx);
        if (!ct || !limbs) throw Error("mkhe_ct_upload_poly_limbs: null argument");
        if (slot < 0 || slot > ct->c.n) throw Error("mkhe_ct_upload_poly_limbs: bad slot");
        for (int l = 0; l < ct->c.limbs; ++l)
            MKHE_HIP(hipMemcpyAsync(ct->c.d + ((size_t)slot * ct->c.limbs + l) * c->N, limbs[l], (size_t)c->N * sizeof(u64), hipMemcpyHostToDevice, c->stream));
        c->sync();
    })
}
int mkhe_ct_download(mkhe_ctx* ctx, const mkhe_ct* ct, uint64_t* host) {
    MKHE_TRY({ mark(ctx, ct);
        Context* c = need(ctx);
        if (!ct || !host) throw Error("mkhe_ct_download: null argument");
        const size_t w = (size_t)(1 + ct->c.n) * ct->c.limbs * c->N;
        MKHE_HIP(hipMemcpyAsync(host, ct->c.d, w * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        c->sync();
    })
}
int mkhe_ct_download_poly_limbs(mkhe_ctx* ctx, const mkhe_ct* ct, int slot, uint64_t* const* limbs) {
    MKHE_TRY({ mark(ctx, ct);
        Context* c = need(ctx);
        if (!ct || !limbs) throw Error("mkhe_ct_download_poly_limbs: null argument");
        if (slot < 0 || slot > ct->c.n) throw Error("mkhe_ct_download_poly_limbs: bad slot");
        for (int l = 0; l < ct->c.limbs; ++l)
            MKHE_HIP(hipMemcpyAsync(limbs[l], ct->c.d + ((size_t)slot * ct->c.limbs + l) * c->N, (size_t)c->N * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        c->sync();
    })
}
int mkhe_ct_limbs(const mkhe_ct* ct) { return ct ? ct->c.limbs : 0; }
int mkhe_ct_nparties(const mkhe_ct* ct) { return ct ? ct->c.n : 0; }
void* mkhe_ct_devptr(mkhe_ct* ct) { if (ct) ct->users.exposed = true; return ct ? ct->c.d : nullptr; }

// ---- raw buffers
int mkhe_buf_alloc(mkhe_ctx* ctx, size_t words, void** dev_out) {
    MKHE_TRY({
        MKHE_HIP(hipSetDevice(need(ctx)->device));
        void* d = nullptr;
        MKHE_HIP(hipMalloc(&d, (words ? words : 1) * sizeof(u64)));
        *dev_out = d;
    })
}
void mkhe_buf_free(mkhe_ctx* ctx, void* dev) {
    if (!dev) return;
    if (ctx) (void)hipStreamSynchronize(ctx->c->stream);
    (void)hipFree(dev);
}
int mkhe_buf_upload(mkhe_ctx* ctx, void* dev, const uint64_t* host, size_t words) {
    MKHE_TRY({
        MKHE_HIP(hipMemcpyAsync(dev, host, words * sizeof(u64), hipMemcpyHostToDevice, need(ctx)->stream));
        need(ctx)->sync();
    })
}
int mkhe_buf_download(mkhe_ctx* ctx, const void* dev, uint64_t* host, size_t words) {
    MKHE_TRY({
        MKHE_HIP(hipMemcpyAsync(host, dev, words * sizeof(u64), hipMemcpyDeviceToHost, need(ctx)->stream));
        need(ctx)->sync();
    })
}

// ---- ring level
int mkhe_ntt(mkhe_ctx* ctx, const void* src, void* dst, int count, int limbs, int mod_base, int inverse, int lazy) {
    MKHE_TRY({ if (!src || !dst || count < 1 || limbs < 1) throw Error("mkhe_ntt: bad argument"); need(ctx)->ntt((const u64*)src, (u64*)dst, count, limbs, mod_base, inverse != 0, lazy != 0); })
}

// ---- KeySwitcher
static const u64* ct_slot(const Context* c, const mkhe_ct* ct, int slot, int level) {
    if (!ct) throw Error("mkhe: null ciphertext");
    if (slot < 0 || slot > ct->c.n) throw Error("mkhe: bad ciphertext slot");
    if (ct->c.limbs < level + 1) throw Error("mkhe: ciphertext level below requested level");
    return ct->c.d + (size_t)slot * ct->c.limbs * c->N;
}
int mkhe_decompose(mkhe_ctx* ctx, int level, int is_ntt, const mkhe_ct* ct, int slot, mkhe_swk* out) {
    MKHE_TRY({ mark(ctx, ct, out); if (!out) throw Error("mkhe_decompose: null argument"); need(ctx)->decompose(level, is_ntt != 0, ct_slot(need(ctx), ct, slot, level), out->s.d); })
}
int mkhe_hoisted_form(mkhe_ctx* ctx, int level, const mkhe_ct* ct, mkhe_swk* const* out) {
    MKHE_TRY({ mark(ctx, ct);
        if (!ct || (ct->c.n > 0 && !out)) throw Error("mkhe_hoisted_form: null argument");
        std::vector<const u64*> src; std::vector<u64*> dst;
        for (int i = 0; i < ct->c.n; ++i) {
            if (!out[i]) throw Error("mkhe_hoisted_form: null output handle");
            mark(ctx, out[i]);
            src.push_back(ct_slot(need(ctx), ct, 1 + i, level)); dst.push_back(out[i]->s.d);
        }
        if (!src.empty()) need(ctx)->decompose_batch(level, src, dst, false);
    })
}
int mkhe_external_product(mkhe_ctx* ctx, int level, int is_ntt, const mkhe_ct* a, int slot,
                          const mkhe_swk* bg, mkhe_ct* out, int out_slot) {
    MKHE_TRY({ mark(ctx, a, bg, out);
        if (!out || !bg) throw Error("mkhe_external_product: null argument");
        if (out->c.limbs != level + 1) throw Error("mkhe_external_product: out must have level+1 limbs");
        need(ctx)->external_product(level, is_ntt != 0, ct_slot(need(ctx), a, slot, level), bg->s.d,
                                 const_cast<u64*>(ct_slot(need(ctx), out, out_slot, level)), false);
    })
}
int mkhe_external_product_hoisted(mkhe_ctx* ctx, int level, const mkhe_swk* ah, const mkhe_swk* bg, mkhe_ct* out, int out_slot) {
    MKHE_TRY({ mark(ctx, ah, bg, out);
        if (!out || !ah || !bg) throw Error("mkhe_external_product_hoisted: null argument");
        if (out->c.limbs != level + 1) throw Error("mkhe_external_product_hoisted: out must have level+1 limbs");
        need(ctx)->external_product_hoisted(level, ah->s.d, bg->s.d, const_cast<u64*>(ct_slot(need(ctx), out, out_slot, level)), false);
    })
}
int mkhe_mul_and_relin(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                       const mkhe_swk* const* hoist0, const mkhe_swk* const* hoist1,
                       const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_d0,
                       const mkhe_swk* const* rlk_v0, const mkhe_swk* crs_u, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, op0, op1, crs_u, out);
        if (!op0 || !op1 || !out || !crs_u || !rlk_b1 || !rlk_d0 || !rlk_v0) throw Error("mkhe_mul_and_relin: null argument");
        auto h0 = swk_list(ctx, hoist0, op0->c.n); auto h1 = swk_list(ctx, hoist1, op1->c.n);
        auto b1 = swk_list(ctx, rlk_b1, op1->c.n); auto d0 = swk_list(ctx, rlk_d0, op0->c.n); auto v0 = swk_list(ctx, rlk_v0, op0->c.n);
        const bool same = (op0 == op1) && (hoist0 == hoist1);
        need(ctx)->mul_and_relin(op0->c, same ? op0->c : op1->c, hoist0 ? h0.data() : nullptr,
                              hoist1 ? (same ? h0.data() : h1.data()) : nullptr,
                              b1.data(), d0.data(), v0.data(), crs_u->s, out->c);
    })
}
int mkhe_mul_relin_rescale(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                           const mkhe_swk* const* hoist0, const mkhe_swk* const* hoist1,
                           const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_d0,
                           const mkhe_swk* const* rlk_v0, const mkhe_swk* crs_u, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, op0, op1, crs_u, out);
        if (!op0 || !op1 || !out || !crs_u || !rlk_b1 || !rlk_d0 || !rlk_v0) throw Error("mkhe_mul_relin_rescale: null argument");
        auto h0 = swk_list(ctx, hoist0, op0->c.n); auto h1 = swk_list(ctx, hoist1, op1->c.n);
        auto b1 = swk_list(ctx, rlk_b1, op1->c.n); auto d0 = swk_list(ctx, rlk_d0, op0->c.n); auto v0 = swk_list(ctx, rlk_v0, op0->c.n);
        const bool same = (op0 == op1) && (hoist0 == hoist1);
        need(ctx)->mul_relin_rescale(op0->c, same ? op0->c : op1->c, hoist0 ? h0.data() : nullptr,
                                     hoist1 ? (same ? h0.data() : h1.data()) : nullptr,
                                     b1.data(), d0.data(), v0.data(), crs_u->s, out->c);
    })
}
int mkhe_mr_partial(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                    const mkhe_swk* const* hoist0, const mkhe_swk* const* hoist1,
                    const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_d0,
                    int with_c0, mkhe_ct* out, mkhe_swk* x_part, mkhe_swk* y_part) {
    MKHE_TRY({ mark(ctx, op0, op1, out, x_part, y_part);
        if (!op0 || !op1 || !out || !x_part || !y_part || !rlk_b1 || !rlk_d0) throw Error("mkhe_mr_partial: null argument");
        auto h0 = swk_list(ctx, hoist0, op0->c.n); auto h1 = swk_list(ctx, hoist1, op1->c.n);
        auto b1 = swk_list(ctx, rlk_b1, op1->c.n); auto d0 = swk_list(ctx, rlk_d0, op0->c.n);
        need(ctx)->mr_prepare(op0->c, op1->c, hoist0 ? h0.data() : nullptr, hoist1 ? h1.data() : nullptr, with_c0 != 0, out->c);
        need(ctx)->mr_xy(b1.data(), d0.data(), x_part->s.d, y_part->s.d, false);
    })
}
int mkhe_swk_fold(mkhe_ctx* ctx, mkhe_swk* swk, int level, int mform) {
    MKHE_TRY({ mark(ctx, swk); if (!swk) throw Error("mkhe_swk_fold: null argument"); need(ctx)->fold(swk->s.d, true, level, need(ctx)->beta(level), (long)need(ctx)->mtot * need(ctx)->N, mform != 0); })
}
int mkhe_swk_fold_pieces(mkhe_ctx* ctx, const void* pieces, int npieces, long piece_stride_words, long first_limb, long nlimbs, int level, int mform, void* dst) {
    MKHE_TRY({ mark(ctx); need(ctx)->fold_pieces((const u64*)pieces, npieces, piece_stride_words, first_limb, nlimbs, level, mform != 0, (u64*)dst); })
}
int mkhe_mr_finish(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, const mkhe_swk* x, const mkhe_swk* y,
                   const mkhe_swk* const* rlk_v0, const mkhe_swk* crs_u, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, op0, op1, x, y, crs_u, out);
        if (!op0 || !op1 || !x || !y || !rlk_v0 || !crs_u || !out) throw Error("mkhe_mr_finish: null argument");
        auto v0 = swk_list(ctx, rlk_v0, op0->c.n);
        need(ctx)->mr_finish(op0->c, op1->c, x->s.d, y->s.d, v0.data(), crs_u->s, out->c);
    })
}
int mkhe_mr_finish_head(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, const mkhe_swk* y, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, op0, op1, y, out);
        if (!op0 || !op1 || !y || !out) throw Error("mkhe_mr_finish_head: null argument");
        need(ctx)->mr_finish_head(op0->c, op1->c, y->s.d, out->c);
    })
}
int mkhe_mr_finish_tail(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, const mkhe_swk* x,
                        const mkhe_swk* const* rlk_v0, const mkhe_swk* crs_u, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, op0, op1, x, crs_u, out);
        if (!op0 || !op1 || !x || !rlk_v0 || !crs_u || !out) throw Error("mkhe_mr_finish_tail: null argument");
        auto v0 = swk_list(ctx, rlk_v0, op0->c.n);
        need(ctx)->mr_finish_tail(op0->c, op1->c, x->s.d, v0.data(), crs_u->s, out->c);
    })
}
int mkhe_ct_fold(mkhe_ctx* ctx, mkhe_ct* ct) {
    MKHE_TRY({ mark(ctx, ct); if (!ct) throw Error("mkhe_ct_fold: null argument"); need(ctx)->fold(ct->c.d, false, ct->c.limbs - 1, 1 + ct->c.n, (long)ct->c.limbs * need(ctx)->N, false); })
}
int mkhe_rotate(mkhe_ctx* ctx, uint64_t galEl, const mkhe_ct* in, const mkhe_swk* const* hoist,
                const mkhe_swk* const* rk, const mkhe_swk* crs, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, in, crs, out);
        if (!in || !out || !rk || !crs) throw Error("mkhe_rotate: null argument");
        auto h = swk_list(ctx, hoist, in->c.n); auto r = swk_list(ctx, rk, in->c.n);
        need(ctx)->rotate(galEl, in->c, hoist ? h.data() : nullptr, r.data(), crs->s, out->c);
    })
}
int mkhe_ctx_set_owned(mkhe_ctx* ctx, const int* mod_idx, int n) {
    MKHE_TRY({ if (n < 0 || (n > 0 && !mod_idx)) throw Error("mkhe_ctx_set_owned: bad argument"); need(ctx)->set_owned(mod_idx, n); })
}
int mkhe_lsh_phase(mkhe_ctx* ctx, int phase, const mkhe_ct* op0, const mkhe_ct* op1,
                   const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_d0, const mkhe_swk* const* rlk_v0,
                   const mkhe_swk* crs_u, mkhe_ct* out, void* dev_stage, size_t* words_out) {
    MKHE_TRY({ mark(ctx, op0, op1, crs_u, out);
        if (!op0 || !op1 || !out || !words_out) throw Error("mkhe_lsh_phase: null argument");
        auto b1 = swk_list(ctx, rlk_b1, op1->c.n); auto d0 = swk_list(ctx, rlk_d0, op0->c.n); auto v0 = swk_list(ctx, rlk_v0, op0->c.n);
        *words_out = need(ctx)->lsh_phase(phase, op0->c, op1->c, rlk_b1 ? b1.data() : nullptr, rlk_d0 ? d0.data() : nullptr,
                                       rlk_v0 ? v0.data() : nullptr, crs_u ? &crs_u->s : nullptr, out->c, (u64*)dev_stage);
    })
}
int mkhe_rotate_partial(mkhe_ctx* ctx, const mkhe_ct* in, const mkhe_swk* const* hoist,
                        const mkhe_swk* const* rk, const mkhe_swk* crs, int with_c0, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, in, crs, out);
        if (!in || !out || !rk || !crs) throw Error("mkhe_rotate_partial: null argument");
        auto h = swk_list(ctx, hoist, in->c.n); auto r = swk_list(ctx, rk, in->c.n);
        need(ctx)->rotate_partial(in->c, hoist ? h.data() : nullptr, r.data(), crs->s, with_c0 != 0, out->c);
    })
}
int mkhe_ct_automorphism(mkhe_ctx* ctx, uint64_t galEl, const mkhe_ct* in, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, in, out); if (!in || !out) throw Error("mkhe_ct_automorphism: null argument"); need(ctx)->automorphism(galEl, in->c, out->c); })
}
int mkhe_conjugate(mkhe_ctx* ctx, uint64_t galEl, const mkhe_ct* in, const mkhe_swk* const* ck,
                   const mkhe_swk* crs, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, in, crs, out);
        if (!in || !out || !ck || !crs) throw Error("mkhe_conjugate: null argument");
        auto k = swk_list(ctx, ck, in->c.n);
        need(ctx)->conjugate(galEl, in->c, k.data(), crs->s, out->c);
    })
}
int mkhe_rescale(mkhe_ctx* ctx, const mkhe_ct* in, int nb, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, in, out); if (!in || !out) throw Error("mkhe_rescale: null argument"); need(ctx)->rescale(in->c, nb, out->c); })
}

int mkhe_ct_add(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, op0, op1, out); if (!op0 || !op1 || !out) throw Error("mkhe_ct_add: null argument"); need(ctx)->ct_binary(0, op0->c, op1->c, out->c); })
}
int mkhe_ct_sub(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, op0, op1, out); if (!op0 || !op1 || !out) throw Error("mkhe_ct_sub: null argument"); need(ctx)->ct_binary(1, op0->c, op1->c, out->c); })
}
int mkhe_ct_sum(mkhe_ctx* ctx, int n, const mkhe_ct* const* in, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, out);
        if (n < 1 || !in || !out) throw Error("mkhe_ct_sum: bad argument");
        std::vector<const Ct*> v(n);
        for (int i = 0; i < n; ++i) { if (!in[i]) throw Error("mkhe_ct_sum: null ciphertext"); mark(ctx, in[i]); v[i] = &in[i]->c; }
        need(ctx)->ct_sum(v, out->c);
    })
}
int mkhe_ct_mul_const(mkhe_ctx* ctx, const mkhe_ct* in, const uint64_t* c_first, const uint64_t* c_second, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, in, out); if (!in || !out || !c_first || !c_second) throw Error("mkhe_ct_mul_const: null argument"); need(ctx)->ct_mul_const(in->c, c_first, c_second, out->c); })
}
int mkhe_ct_mul_ptxt(mkhe_ctx* ctx, const mkhe_ct* in, const void* dev_pt, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, in, out); if (!in || !out || !dev_pt) throw Error("mkhe_ct_mul_ptxt: null argument"); need(ctx)->ct_mul_ptxt(in->c, (const u64*)dev_pt, out->c); })
}

// ---- B operations of one shape per call (batch.hip)
static std::vector<const Ct*> ct_list(const mkhe_ctx* ctx, const mkhe_ct* const* v, int n, const char* what) {
    if (!v) throw Error(std::string(what) + ": null ciphertext list");
    std::vector<const Ct*> r(n);
    for (int i = 0; i < n; ++i) { if (!v[i]) throw Error(std::string(what) + ": null ciphertext in the batch"); mark(ctx, v[i]); r[i] = &v[i]->c; }
    return r;
}
static std::vector<Ct*> ct_list_out(const mkhe_ctx* ctx, mkhe_ct* const* v, int n, const char* what) {
    if (!v) throw Error(std::string(what) + ": null output list");
    std::vector<Ct*> r(n);
    for (int i = 0; i < n; ++i) {
        if (!v[i]) throw Error(std::string(what) + ": null output in the batch");
        for (int k = 0; k < i; ++k) if (v[k] == v[i]) throw Error(std::string(what) + ": the outputs of a batch must be distinct");
        mark(ctx, v[i]); r[i] = &v[i]->c;
    }
    return r;
}
static std::vector<const Swk*> swk_flat(const mkhe_ctx* ctx, const mkhe_swk* const* v, size_t n) {
    std::vector<const Swk*> r;
    if (!v) return r;
    r.resize(n);
    for (size_t i = 0; i < n; ++i) { if (!v[i]) throw Error("mkhe: null hoisted form in a batch"); mark(ctx, v[i]); r[i] = &v[i]->s; }
    return r;
}
int mkhe_hoisted_form_batch(mkhe_ctx* ctx, int level, int nbatch, const mkhe_ct* const* cts, mkhe_swk* const* out) {
    MKHE_TRY({
        if (nbatch < 1) throw Error("mkhe_hoisted_form_batch: empty batch");
        auto c = ct_list(ctx, cts, nbatch, "mkhe_hoisted_form_batch");
        const size_t n = (size_t)nbatch * c[0]->n;
        if (n && !out) throw Error("mkhe_hoisted_form_batch: null argument");
        std::vector<Swk*> o(n);
        for (size_t i = 0; i < n; ++i) { if (!out[i]) throw Error("mkhe_hoisted_form_batch: null output handle"); mark(ctx, out[i]); o[i] = &out[i]->s; }
        need(ctx)->hoisted_form_batch(level, c, o);
    })
}
int mkhe_rotate_batch(mkhe_ctx* ctx, uint64_t galEl, int nbatch, const mkhe_ct* const* in, const mkhe_swk* const* hoist,
                      const mkhe_swk* const* rk, const mkhe_swk* crs, mkhe_ct* const* out) {
    MKHE_TRY({ mark(ctx, crs);
        if (nbatch < 1 || !rk || !crs) throw Error("mkhe_rotate_batch: bad argument");
        auto i = ct_list(ctx, in, nbatch, "mkhe_rotate_batch");
        auto o = ct_list_out(ctx, out, nbatch, "mkhe_rotate_batch");
        auto h = swk_flat(ctx, hoist, (size_t)nbatch * i[0]->n);
        auto k = swk_list(ctx, rk, i[0]->n);
        need(ctx)->rotate_batch(galEl, i, h, k.data(), crs->s, o);
    })
}
int mkhe_rotate_multi(mkhe_ctx* ctx, int nbatch, const uint64_t* galEl, const mkhe_ct* const* in, const mkhe_swk* const* hoist,
                      const mkhe_swk* const* rk, const mkhe_swk* const* crs, const mkhe_ct* const* post_add, mkhe_ct* const* out) {
    MKHE_TRY({
        if (nbatch < 1 || !galEl || !rk || !crs) throw Error("mkhe_rotate_multi: bad argument");
        auto i = ct_list(ctx, in, nbatch, "mkhe_rotate_multi");
        auto o = ct_list_out(ctx, out, nbatch, "mkhe_rotate_multi");
        auto h = swk_flat(ctx, hoist, (size_t)nbatch * i[0]->n);
        auto k = swk_flat(ctx, rk, (size_t)nbatch * i[0]->n);
        auto c = swk_flat(ctx, crs, (size_t)nbatch);
        std::vector<const Ct*> p;
        if (post_add) p = ct_list(ctx, post_add, nbatch, "mkhe_rotate_multi");
        need(ctx)->rotate_multi(std::vector<u64>(galEl, galEl + nbatch), i, h, k, c, p, o);
    })
}
int mkhe_mul_relin_batch(mkhe_ctx* ctx, int nbatch, const mkhe_ct* const* op0, const mkhe_ct* const* op1,
                         const mkhe_swk* const* hoist0, const mkhe_swk* const* hoist1,
                         const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_d0, const mkhe_swk* const* rlk_v0,
                         const mkhe_swk* crs_u, int rescale, mkhe_ct* const* out) {
    MKHE_TRY({ mark(ctx, crs_u);
        if (nbatch < 1 || !crs_u || !rlk_b1 || !rlk_d0 || !rlk_v0) throw Error("mkhe_mul_relin_batch: bad argument");
        auto a = ct_list(ctx, op0, nbatch, "mkhe_mul_relin_batch");
        auto b = ct_list(ctx, op1, nbatch, "mkhe_mul_relin_batch");
        auto o = ct_list_out(ctx, out, nbatch, "mkhe_mul_relin_batch");
        auto h0 = swk_flat(ctx, hoist0, (size_t)nbatch * a[0]->n); auto h1 = swk_flat(ctx, hoist1, (size_t)nbatch * b[0]->n);
        auto b1 = swk_list(ctx, rlk_b1, b[0]->n); auto d0 = swk_list(ctx, rlk_d0, a[0]->n); auto v0 = swk_list(ctx, rlk_v0, a[0]->n);
        need(ctx)->mul_relin_batch(a, b, h0, h1, b1.data(), d0.data(), v0.data(), crs_u->s, rescale != 0, o);
    })
}
int mkhe_ct_mul_ptxt_batch(mkhe_ctx* ctx, int nbatch, const mkhe_ct* const* in, const void* dev_pt, int nb_rescale, mkhe_ct* const* out) {
    MKHE_TRY({
        if (nbatch < 1 || !dev_pt) throw Error("mkhe_ct_mul_ptxt_batch: bad argument");
        auto i = ct_list(ctx, in, nbatch, "mkhe_ct_mul_ptxt_batch");
        auto o = ct_list_out(ctx, out, nbatch, "mkhe_ct_mul_ptxt_batch");
        need(ctx)->ct_mul_ptxt_batch(i, (const u64*)dev_pt, nb_rescale, o);
    })
}
int mkhe_ct_binary_batch(mkhe_ctx* ctx, int op, int nbatch, const mkhe_ct* const* op0, const mkhe_ct* const* op1, mkhe_ct* const* out) {
    MKHE_TRY({
        if (nbatch < 1 || (op != 0 && op != 1)) throw Error("mkhe_ct_binary_batch: bad argument");
        auto a = ct_list(ctx, op0, nbatch, "mkhe_ct_binary_batch");
        auto b = ct_list(ctx, op1, nbatch, "mkhe_ct_binary_batch");
        auto o = ct_list_out(ctx, out, nbatch, "mkhe_ct_binary_batch");
        need(ctx)->ct_binary_batch(op, a, b, o);
    })
}

// ---- key generation / CRS expansion
int mkhe_keygen_secret(mkhe_ctx* ctx, const int32_t* s, void* dev_sk) {
    MKHE_TRY({ if (!s || !dev_sk) throw Error("mkhe_keygen_secret: null argument"); need(ctx)->keygen_secret(s, (u64*)dev_sk); })
}
int mkhe_keygen_switching_key(mkhe_ctx* ctx, const void* dev_sk, const int32_t* e, mkhe_swk* out) {
    MKHE_TRY({ mark(ctx, out); if (!dev_sk || !e || !out) throw Error("mkhe_keygen_switching_key: null argument"); need(ctx)->keygen_switching_key((const u64*)dev_sk, e, out->s.d); })
}
int mkhe_keygen_public_key(mkhe_ctx* ctx, const void* dev_sk, const int32_t* e, const mkhe_swk* crs_a, void* dev_pk) {
    MKHE_TRY({ mark(ctx, crs_a); if (!dev_sk || !e || !crs_a || !dev_pk) throw Error("mkhe_keygen_public_key: null argument"); need(ctx)->keygen_public_key((const u64*)dev_sk, e, crs_a->s.d, (u64*)dev_pk); })
}
int mkhe_keygen_relin_key(mkhe_ctx* ctx, const void* dev_sk, const void* dev_r, const int32_t* e,
                          const mkhe_swk* crs_a, const mkhe_swk* crs_u, mkhe_swk* b, mkhe_swk* d, mkhe_swk* v) {
    MKHE_TRY({ mark(ctx, crs_a, crs_u, b, d, v);
        if (!dev_sk || !dev_r || !e || !crs_a || !crs_u || !b || !d || !v) throw Error("mkhe_keygen_relin_key: null argument");
        need(ctx)->keygen_relin_key((const u64*)dev_sk, (const u64*)dev_r, e, crs_a->s.d, crs_u->s.d, b->s.d, d->s.d, v->s.d);
    })
}
int mkhe_keygen_rotation_key(mkhe_ctx* ctx, uint64_t galEl, const void* dev_sk, const int32_t* e, const mkhe_swk* crs, mkhe_swk* out) {
    MKHE_TRY({ mark(ctx, crs, out); if (!dev_sk || !e || !crs || !out) throw Error("mkhe_keygen_rotation_key: null argument"); need(ctx)->keygen_rotation_key(galEl, (const u64*)dev_sk, e, crs->s.d, out->s.d); })
}
int mkhe_keygen_conjugation_key(mkhe_ctx* ctx, const void* dev_sk, const int32_t* e, const mkhe_swk* crs, mkhe_swk* out) {
    MKHE_TRY({ mark(ctx, crs, out); if (!dev_sk || !e || !crs || !out) throw Error("mkhe_keygen_conjugation_key: null argument"); need(ctx)->keygen_conjugation_key((const u64*)dev_sk, e, crs->s.d, out->s.d); })
}
int mkhe_bfv_keygen_switching_key(mkhe_ctx* ctx, const void* dev_sk, const uint64_t* g, const int32_t* e, mkhe_swk* out) {
    MKHE_TRY({ mark(ctx, out); if (!dev_sk || !g || !e || !out) throw Error("mkhe_bfv_keygen_switching_key: null argument"); need(ctx)->bfv_keygen_switching_key((const u64*)dev_sk, g, e, out->s.d); })
}
int mkhe_bfv_keygen_relin_key(mkhe_ctx* ctx, const void* dev_sk, const void* dev_r, const uint64_t* g1, const uint64_t* g2,
                              const int32_t* e, const mkhe_swk* a1, const mkhe_swk* a2, const mkhe_swk* u,
                              mkhe_swk* b1, mkhe_swk* b2, mkhe_swk* d1, mkhe_swk* d2, mkhe_swk* v) {
    MKHE_TRY({ mark(ctx, a1, a2, u, b1, b2, d1, d2, v);
        if (!dev_sk || !dev_r || !g1 || !g2 || !e || !a1 || !a2 || !u || !b1 || !b2 || !d1 || !d2 || !v) throw Error("mkhe_bfv_keygen_relin_key: null argument");
        need(ctx)->bfv_keygen_relin_key((const u64*)dev_sk, (const u64*)dev_r, g1, g2, e, a1->s.d, a2->s.d, u->s.d,
                                     b1->s.d, b2->s.d, d1->s.d, d2->s.d, v->s.d);
    })
}
int mkhe_crs_expand(mkhe_ctx* ctx, uint64_t seed, int32_t idx, mkhe_swk* out) {
    MKHE_TRY({ mark(ctx, out); if (!out) throw Error("mkhe_crs_expand: null argument"); need(ctx)->crs_expand(seed, idx, out->s.d); })
}

// ---- mkbfv
int mkhe_ctx_create_bfv(mkhe_ctx** out, int logN, const uint64_t* Q, const uint64_t* QMul, int nQ,
                        const uint64_t* P, int nP, int gamma, uint64_t T, int device) {
    MKHE_TRY({
        if (!out || !Q || !QMul || !P) throw Error("mkhe_ctx_create_bfv: null argument");
        *out = new mkhe_ctx{new Context(logN, Q, nQ, P, nP, gamma, nullptr, nullptr, device, QMul, nQ, T)};
    })
}
int mkhe_bfv_modup_q_to_r(mkhe_ctx* ctx, const void* q, void* r, int npolys) {
    MKHE_TRY({ if (!q || !r || npolys < 1) throw Error("mkhe_bfv_modup_q_to_r: bad argument"); need(ctx)->bfv_modup_q_to_r((const u64*)q, (u64*)r, npolys); })
}
int mkhe_bfv_rescale(mkhe_ctx* ctx, const void* q, void* r, int npolys) {
    MKHE_TRY({ if (!q || !r || npolys < 1) throw Error("mkhe_bfv_rescale: bad argument"); need(ctx)->bfv_rescale((const u64*)q, (u64*)r, npolys); })
}
int mkhe_bfv_quantize(mkhe_ctx* ctx, const void* r, void* q, int npolys) {
    MKHE_TRY({ if (!q || !r || npolys < 1) throw Error("mkhe_bfv_quantize: bad argument"); need(ctx)->bfv_quantize((const u64*)r, (u64*)q, npolys); })
}
int mkhe_bfv_ntt_r(mkhe_ctx* ctx, const void* src, void* dst, int count, int inverse) {
    MKHE_TRY({ if (!src || !dst || count < 1) throw Error("mkhe_bfv_ntt_r: bad argument"); need(ctx)->ntt_r((const u64*)src, (u64*)dst, count, inverse != 0); })
}
int mkhe_bfv_decompose(mkhe_ctx* ctx, const void* polyr, mkhe_swk* ad1, mkhe_swk* ad2) {
    MKHE_TRY({ mark(ctx, ad1, ad2);
        if (!polyr || !ad1 || !ad2) throw Error("mkhe_bfv_decompose: null argument");
        need(ctx)->bfv_decompose_batch({(const u64*)polyr}, {ad1->s.d}, {ad2->s.d});
    })
}
int mkhe_bfv_external_product(mkhe_ctx* ctx, const void* dev_polyr, const mkhe_swk* bg1, const mkhe_swk* bg2, void* dev_c) {
    MKHE_TRY({ mark(ctx, bg1, bg2);
        if (!dev_polyr || !bg1 || !bg2 || !dev_c) throw Error("mkhe_bfv_external_product: null argument");
        need(ctx)->bfv_external_product((const u64*)dev_polyr, bg1->s.d, bg2->s.d, (u64*)dev_c);
    })
}
int mkhe_bfv_external_product_hoisted(mkhe_ctx* ctx, const mkhe_swk* ah1, const mkhe_swk* ah2,
                                      const mkhe_swk* bg1, const mkhe_swk* bg2, void* dev_c) {
    MKHE_TRY({ mark(ctx, ah1, ah2, bg1, bg2);
        if (!ah1 || !ah2 || !bg1 || !bg2 || !dev_c) throw Error("mkhe_bfv_external_product_hoisted: null argument");
        need(ctx)->bfv_external_product_hoisted(ah1->s.d, ah2->s.d, bg1->s.d, bg2->s.d, (u64*)dev_c);
    })
}
int mkhe_bfv_mul_relin(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                       const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_b2,
                       const mkhe_swk* const* rlk_d1, const mkhe_swk* const* rlk_d2,
                       const mkhe_swk* const* rlk_v, const mkhe_swk* crs_u, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, op0, op1, crs_u, out);
        if (!op0 || !op1 || !out || !crs_u || !rlk_b1 || !rlk_b2 || !rlk_d1 || !rlk_d2 || !rlk_v) throw Error("mkhe_bfv_mul_relin: null argument");
        auto b1 = swk_list(ctx, rlk_b1, op1->c.n); auto b2 = swk_list(ctx, rlk_b2, op1->c.n);
        auto d1 = swk_list(ctx, rlk_d1, op0->c.n); auto d2 = swk_list(ctx, rlk_d2, op0->c.n); auto v = swk_list(ctx, rlk_v, op0->c.n);
        need(ctx)->bfv_mul_relin(op0->c, op1->c, b1.data(), b2.data(), d1.data(), d2.data(), v.data(), crs_u->s, out->c);
    })
}

int mkhe_bfv_mul_relin_unhoisted(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                                 const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_b2,
                                 const mkhe_swk* const* rlk_d1, const mkhe_swk* const* rlk_d2,
                                 const mkhe_swk* const* rlk_v, const mkhe_swk* crs_u, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, op0, op1, crs_u, out);
        if (!op0 || !op1 || !out || !crs_u || !rlk_b1 || !rlk_b2 || !rlk_d1 || !rlk_d2 || !rlk_v) throw Error("mkhe_bfv_mul_relin_unhoisted: null argument");
        auto b1 = swk_list(ctx, rlk_b1, op1->c.n); auto b2 = swk_list(ctx, rlk_b2, op1->c.n);
        auto d1 = swk_list(ctx, rlk_d1, op0->c.n); auto d2 = swk_list(ctx, rlk_d2, op0->c.n); auto v = swk_list(ctx, rlk_v, op0->c.n);
        need(ctx)->bfv_mul_relin_unhoisted(op0->c, op1->c, b1.data(), b2.data(), d1.data(), d2.data(), v.data(), crs_u->s, out->c);
    })
}

int mkhe_bfv_mr_partial(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1,
                        const mkhe_swk* const* rlk_b1, const mkhe_swk* const* rlk_b2,
                        const mkhe_swk* const* rlk_d1, const mkhe_swk* const* rlk_d2, int with_c0, mkhe_ct* out,
                        mkhe_swk* x1, mkhe_swk* x2, mkhe_swk* y1, mkhe_swk* y2) {
    MKHE_TRY({ mark(ctx, op0, op1, out, x1, x2, y1, y2);
        if (!op0 || !op1 || !out || !rlk_b1 || !rlk_b2 || !rlk_d1 || !rlk_d2 || !x1 || !x2 || !y1 || !y2) throw Error("mkhe_bfv_mr_partial: null argument");
        auto b1 = swk_list(ctx, rlk_b1, op1->c.n); auto b2 = swk_list(ctx, rlk_b2, op1->c.n);
        auto d1 = swk_list(ctx, rlk_d1, op0->c.n); auto d2 = swk_list(ctx, rlk_d2, op0->c.n);
        need(ctx)->bfv_mr_partial(op0->c, op1->c, b1.data(), b2.data(), d1.data(), d2.data(), with_c0 != 0, false, out->c,
                                  x1->s.d, x2->s.d, y1->s.d, y2->s.d);
    })
}
int mkhe_bfv_mr_finish(mkhe_ctx* ctx, const mkhe_ct* op0, const mkhe_ct* op1, const mkhe_swk* x1, const mkhe_swk* x2,
                       const mkhe_swk* y1, const mkhe_swk* y2, const mkhe_swk* const* rlk_v, const mkhe_swk* crs_u, mkhe_ct* out) {
    MKHE_TRY({ mark(ctx, op0, op1, x1, x2, y1, y2, crs_u, out);
        if (!op0 || !op1 || !out || !x1 || !x2 || !y1 || !y2 || !rlk_v || !crs_u) throw Error("mkhe_bfv_mr_finish: null argument");
        auto v = swk_list(ctx, rlk_v, op0->c.n);
        need(ctx)->bfv_mr_finish(op0->c, op1->c, x1->s.d, x2->s.d, y1->s.d, y2->s.d, v.data(), crs_u->s, out->c);
    })
}

int mkhe_set_overlap(mkhe_ctx* ctx, int on) { MKHE_TRY({ need(ctx)->sync(); need(ctx)->overlap = on != 0; }) }
int mkhe_ntt_trace(mkhe_ctx* ctx, void* dev_buf) { need(ctx)->ntt_trace = (u64*)dev_buf; return 0; }
int mkhe_prof_enable(mkhe_ctx* ctx, int on) { MKHE_TRY(need(ctx)->prof_enable(on != 0)) }
int mkhe_ntt_choice(mkhe_ctx* ctx, long limbs, int decompose) {
    g_last_ctx = nullptr;
    try {
        Context* c = need(ctx);
        for (int lazy = 0; lazy < 2; ++lazy) {
            auto it = c->ntt_tune_.find((limbs << 2) | (decompose ? 2 : 0) | lazy);
            if (it != c->ntt_tune_.end() && it->second.decided >= 0) return it->second.decided;
        }
        return c->ntt_forced_ >= 0 ? c->ntt_forced_ : -1;
    }
    catch (const std::exception& e) { g_err = e.what(); return -2; }
    catch (...) { g_err = "mkhe: unknown error"; return -2; }
}
int mkhe_ctx_set_ntt_choice(mkhe_ctx* ctx, long limbs, int decompose, int choice) {
    MKHE_TRY({
        Context* c = need(ctx);
        if (choice < -1 || choice > 1) throw Error("mkhe: ntt choice must be -1 (measure), 0 (two-pass) or 1 (single-pass)");
        if (limbs <= 0) { c->ntt_forced_ = choice; for (auto& kv : c->ntt_tune_) c->ntt_reset(kv.second, choice); }
        else for (int lazy = 0; lazy < 2; ++lazy) c->ntt_reset(c->ntt_tune_[(limbs << 2) | (decompose ? 2 : 0) | lazy], choice);
    })
}
int mkhe_ctx_set_batch_lanes(mkhe_ctx* ctx, long min_limbs) { MKHE_TRY({ need(ctx)->batch_lanes_min_ = min_limbs; }) }
long long mkhe_pool_held_bytes(mkhe_ctx* ctx) {
    g_last_ctx = nullptr;
    try { return (long long)(need(ctx)->pool_held_words() * sizeof(u64)); }
    catch (const std::exception& e) { g_err = e.what(); return -1; }
    catch (...) { g_err = "mkhe: unknown error"; return -1; }
}
int mkhe_pool_trim(mkhe_ctx* ctx) { MKHE_TRY(need(ctx)->pool_trim_device()) }
int mkhe_f2_schedule_probe(int parties, int nb, int nslots, const long* weights, int grid, unsigned char* segs, int* parts) {
    static_assert(sizeof(F2Seg) == 8 && F2_SEGS == 3, "include/mkhe.h documents the record");
    int p = 0;
    if (!weights || !segs || grid == 0 || grid < -65536 || grid > 65536) { if (parts) *parts = 0; return 0; }
    try {
        const int n = grid > 0 ? f2_build_schedule(parties, nb, nslots, weights, grid, reinterpret_cast<F2Seg*>(segs), &p)
                               : f2_plan_schedule(parties, nb, nslots, weights, -grid, reinterpret_cast<F2Seg*>(segs), &p);
        if (parts) *parts = p;
        return n;
    } catch (...) { if (parts) *parts = 0; return 0; }
}
int mkhe_prof_nclass(void) { return Context::PROF_NCLASS; }
const char* mkhe_prof_name(int cls) {
    static const char* names[] = {"ntt_fwd_kernel<N,1,true>  (Decompose, q<2^57)", "ntt_fwd_kernel<N,0,true>  (Decompose, q>=2^57)",
                                  "ntt_fwd_kernel<N,2,true>  (Decompose, both modulus classes in one persistent launch)",
                                  "ntt16_fwd_kernel<true>  (Decompose, 16 coefficients per thread, two workgroups per CU)", "ntt16_fwd_kernel<false>",
                                  "ntt32_fwd_kernel<true>  (Decompose, ONE pass per limb: 32 coefficients per thread, one workgroup per CU)", "ntt32_fwd_kernel<false>",
                                  "ntt14_fwd_split_kernel  (N = 2^16 Decompose after decomp_spread4_kernel: four one-pass 2^14-point sub-transforms per limb)",
                                  "ntt_fwd_kernel<N,1,false>", "ntt_fwd_kernel<N,0,false>", "ntt_inv_kernel<N>",
                                  "inner_product_kernel", "ext_inner_kernel", "moddown[_batch]_kernel", "tensor_kernel", "basis_conv_kernel", "decomp_spread_kernel",
                                  "ntt16_f2_kernel  (Decompose of the t_i with the step-F2 products in registers)", "other"};
    static_assert(sizeof(names) / sizeof(names[0]) == Context::PROF_NCLASS, "one name per timing class");
    return (cls >= 0 && cls < Context::PROF_NCLASS) ? names[cls] : "";
}
int mkhe_prof_collect(mkhe_ctx* ctx, double* ms, long* launches, double* alg_bytes) {
    MKHE_TRY(need(ctx)->prof_collect(ms, launches, alg_bytes))
}

}  // extern "C"
