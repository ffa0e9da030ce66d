// poly_kernels.hip -- see poly_kernels.h.
#include "poly_kernels.h"
#include <cstdlib>
#include <stdexcept>

namespace mkhe {

constexpr int PW_THREADS = 256;

// ------------------------------------------------------------------ inner product
// HBM-bound (2 * nterms + 1 words moved per output word): every lane moves 16 bytes per load (two coefficients), all the loads of
// a thread are issued before the first product (the term loop is unrolled: NT is a template argument), and the grid covers the
// limb exactly (no grid-stride loop, no block cap).
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
// Operands that are read exactly once in a launch (the relinearization keys, the hoisted digits) are loaded with the
// non-temporal hint: they stream past L2 / Infinity Cache instead of evicting what IS re-read (x, y, the CRS u, twiddles).
// Measured on MI355X: inner_product_kernel 102 -> 88 us per launch (5.1 -> 6.0 TB/s), the step 1.23 -> 1.217 ms.
__device__ __forceinline__ u64x2 ld_stream(const u64* p) { return __builtin_nontemporal_load((const u64x2*)p); }
__device__ __forceinline__ u64x2 ld_cached(const u64* p) { return *(const u64x2*)p; }
// Wave-uniform table entries (per-modulus constants, base-conversion tables) are read through the constant address space, i.e. with SCALAR
// loads: indexed by a value that came out of a vector load (a slot -> modulus map) or by a loop counter the compiler does not prove uniform,
// they were fetched per lane -- one dependent vector round trip per entry inside the limb loops of the latency-bound kernels (round 3).
typedef const __attribute__((address_space(4))) u64* sc_u64;
typedef const __attribute__((address_space(4))) int* sc_int;
typedef const __attribute__((address_space(4))) Mod* sc_mod;
__device__ __forceinline__ Mod load_mod(sc_mod p) {
    Mod m;
    m.q = p->q; m.q2 = p->q2; m.ninv32 = p->ninv32; m.finv = p->finv; m.qinv = p->qinv; m.r1 = p->r1; m.r2 = p->r2; m.qs = p->qs; m.r1s = p->r1s;
    return m;
}
// table[v] for a per-lane v in [0, n]: the n + 1 entries are wave-uniform (scalar loads), the lane selects
template <int NMAX> __device__ __forceinline__ u64 select_entry(sc_u64 row, int n, u64 v) {
    u64 r = row[0];
#pragma unroll
    for (int i = 1; i <= NMAX; ++i) if (i <= n) r = v == (u64)i ? row[i] : r;
    return r;
}
template <int NT>
__global__ void __launch_bounds__(PW_THREADS) inner_product_kernel(InnerProductArgs a) {
    const int s = blockIdx.y;                 // active-limb slot
    const int o = blockIdx.z;                 // outer item (gadget digit or 0)
    const int m = a.map[s];
    const Mod md = a.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const long base = (long)o * a.term_outer + (long)m * a.N;
    const long obase = (long)o * a.out_outer + (long)m * a.N;
    const int n = 2 * (blockIdx.x * PW_THREADS + threadIdx.x);
    if (n >= a.N) return;
    const int nt = NT ? NT : a.nterms;
    u64 acc0 = 0, acc1 = 0;
    if (a.addend) { const u64x2 p = ld_cached(a.addend + obase + n); acc0 = p.x; acc1 = p.y; }      // canonical: < q < 2q
    if constexpr (NT != 0) {
        u64x2 x[NT], y[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) { x[t] = ld_stream(a.a[t] + base + n); y[t] = ld_stream(a.b[t] + base + n); }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc0 = csub(acc0 + mont_mul_lazy(x[t].x, y[t].x, q, ninv), q2);
            acc1 = csub(acc1 + mont_mul_lazy(x[t].y, y[t].y, q, ninv), q2);
        }
    } else {
        for (int t = 0; t < nt; ++t) {
            const u64x2 x = ld_stream(a.a[t] + base + n), y = ld_stream(a.b[t] + base + n);
            acc0 = csub(acc0 + mont_mul_lazy(x.x, y.x, q, ninv), q2);
            acc1 = csub(acc1 + mont_mul_lazy(x.y, y.y, q, ninv), q2);
        }
    }
    acc0 = csub(acc0, q); acc1 = csub(acc1, q);
    if (a.mform_out) { acc0 = mont_mul(acc0, md.r2, q, ninv); acc1 = mont_mul(acc1, md.r2, q, ninv); }
    u64x2 r; r.x = acc0; r.y = acc1;
    *(u64x2*)(a.out + obase + n) = r;
}

typedef const __attribute__((address_space(4))) InnerProductBatchArgs* ipb_kargs;
__global__ void __launch_bounds__(PW_THREADS) inner_product_batch_kernel(InnerProductBatchArgs a) {
    ipb_kargs ka = (ipb_kargs)__builtin_amdgcn_kernarg_segment_ptr();      // per-input pointer lists: scalar loads
    const int s = blockIdx.y;
    const int bi = blockIdx.z / a.nouter, o = blockIdx.z - bi * a.nouter;
    const int m = a.map[s];
    const Mod md = a.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const long base = (long)o * a.term_outer + (long)m * a.N;
    const long obase = (long)o * a.out_outer + (long)m * a.N;
    const int n = 2 * (blockIdx.x * PW_THREADS + threadIdx.x);
    if (n >= a.N) return;
    u64 acc0 = 0, acc1 = 0;
    for (int t = 0; t < a.nterms; ++t) {
        const u64x2 x = ld_cached(ka->a[t] + base + n), y = ld_stream(ka->b[bi][t] + base + n);      // (the keys are re-read by every input)
        acc0 = csub(acc0 + mont_mul_lazy(x.x, y.x, q, ninv), q2);
        acc1 = csub(acc1 + mont_mul_lazy(x.y, y.y, q, ninv), q2);
    }
    acc0 = csub(acc0, q); acc1 = csub(acc1, q);
    if (a.mform_out) { acc0 = mont_mul(acc0, md.r2, q, ninv); acc1 = mont_mul(acc1, md.r2, q, ninv); }
    u64x2 r; r.x = acc0; r.y = acc1;
    *(u64x2*)(ka->out[bi] + obase + n) = r;
}
void launch_inner_product_batch(const InnerProductBatchArgs& a, hipStream_t st) {
    const dim3 grid((a.N / 2 + PW_THREADS - 1) / PW_THREADS, a.nslots, a.nouter * a.nbatch), blk(PW_THREADS);
    hipLaunchKernelGGL(inner_product_batch_kernel, grid, blk, 0, st, a);
}

void launch_inner_product(const InnerProductArgs& a, hipStream_t st) {
    const dim3 grid((a.N / 2 + PW_THREADS - 1) / PW_THREADS, a.nslots, a.nouter), blk(PW_THREADS);
    switch (a.nterms) {
        case 1: hipLaunchKernelGGL(inner_product_kernel<1>, grid, blk, 0, st, a); break;
        case 2: hipLaunchKernelGGL(inner_product_kernel<2>, grid, blk, 0, st, a); break;
        case 3: hipLaunchKernelGGL(inner_product_kernel<3>, grid, blk, 0, st, a); break;
        case 4: hipLaunchKernelGGL(inner_product_kernel<4>, grid, blk, 0, st, a); break;
        case 8: hipLaunchKernelGGL(inner_product_kernel<8>, grid, blk, 0, st, a); break;
        default: hipLaunchKernelGGL(inner_product_kernel<0>, grid, blk, 0, st, a); break;
    }
}

// ------------------------------------------------------------------ ModDown
// reconstructRNS + multSum + the MRed tail, restated literally per coefficient
// (basis_extension.go:192-232, 337-357, 537-646).  The float64 correction index is evaluated
// as sequential IEEE divide/add (no contraction: see the -ffp-contract=off build flag).
__global__ void __launch_bounds__(PW_THREADS) moddown_kernel(ModDownArgs a) {
    const int bi = blockIdx.z;
    const u64* xq = a.xq + (long)bi * a.xq_batch;
    const u64* xp = a.xp + (long)bi * a.xp_batch;
    u64* dst = a.dst + (long)bi * a.dst_batch;
    const int n = blockIdx.x * PW_THREADS + threadIdx.x;
    if (n >= a.N) return;
    u64 y[MAXP];
    double vi = 0.0;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        if (i < a.np) {
            const Mod mp = a.mods_p[i];
            y[i] = mont_mul(xp[(long)i * a.N + n], a.t.qoverqiinvqi[i], mp.q, mp.ninv32);
            vi = vi + (double)y[i] / (double)mp.q;
        }
    }
    const u64 v = (u64)vi;
    // limbs are split over blockIdx.y to fill the chip
    for (int j = blockIdx.y; j <= a.level; j += gridDim.y) {
        const Mod mq = a.mods_q[j];
        u64 rlo = 0, rhi = 0;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            if (i < a.np) {
                u64 mhi, mlo;
                mul64x64(y[i], a.t.qoverqimodp[(long)j * a.np + i], mhi, mlo);
                u64 sum = rlo + mlo;
                rhi += mhi + (sum < rlo ? 1 : 0);
                rlo = sum;
            }
        }
        const u64 hhi = mulhi64(rlo * mq.qinv, mq.q);
        const u64 lift = rhi - hhi + mq.q + a.t.vtimesqmodp[(long)j * (a.np + 1) + v];
        const u64 x = xq[(long)j * a.N + n];
        u64 z = mont_mul(lift + mq.q2 - x, a.t.downparam[j], mq.q, mq.ninv32);
        if (a.accumulate) z = csub(dst[(long)j * a.N + n] + z, mq.q);
        dst[(long)j * a.N + n] = z;
    }
}

void launch_moddown(const ModDownArgs& a, hipStream_t st) {
    const int bx = (a.N + PW_THREADS - 1) / PW_THREADS;
    int by = a.level + 1;
    if (by > 4) by = 4;
    hipLaunchKernelGGL(moddown_kernel, dim3(bx, by, a.nbatch), dim3(PW_THREADS), 0, st, a);
}

// ------------------------------------------------------------------ batched ExternalProduct halves
// HBM-bound like the inner product: 16-byte lanes (two coefficients per thread), the digit loop unrolled by four so that
// 8 (12 for a pair) loads are in flight per thread, the grid covers the limb exactly.
typedef const __attribute__((address_space(4))) ExtInnerArgs* ext_kargs;
__device__ __forceinline__ void ext_store(u64* out, u64 x0, u64 x1, u64 q) { u64x2 r; r.x = csub(x0, q); r.y = csub(x1, q); *(u64x2*)out = r; }
// G single items that share bg: out_m = sum_i bg[i] (.) ah_m[i]
template <int G>
__device__ __forceinline__ void ext_group_singles(const ExtInnerArgs& a, ext_kargs ka, int leader, long off, const Mod& md) {
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const long ds = a.digit_stride;
    const u64* ah[G]; int idx[G];
    int it = leader;
#pragma unroll
    for (int m = 0; m < G; ++m) { idx[m] = it; ah[m] = ka->ah[it] + off; it = ka->gnext[it]; }
    const u64* bg = ka->bg[leader] + off;
    const bool once = ka->bg_once[leader] != 0;
    u64 acc[G][2];
#pragma unroll
    for (int m = 0; m < G; ++m) { acc[m][0] = 0; acc[m][1] = 0; }
    if (a.xout) {
        // with the x by-product: per digit the G keys d_i as well, x[d] stored as soon as it is complete
        const u64* xk[G];
#pragma unroll
        for (int m = 0; m < G; ++m) xk[m] = ka->xkey[idx[m]] + off;
        u64* xo = (a.xmulti ? (u64*)ka->xkey2[leader] : a.xout) + off;
#pragma unroll 1
        for (int i = 0; i < a.nb; ++i) {
            const u64x2 g = once ? ld_stream(bg + i * ds) : ld_cached(bg + i * ds);
            u64x2 h[G], k[G];
#pragma unroll
            for (int m = 0; m < G; ++m) { h[m] = ld_stream(ah[m] + i * ds); k[m] = ld_stream(xk[m] + i * ds); }
            u64 x0 = 0, x1 = 0;
#pragma unroll
            for (int m = 0; m < G; ++m) {
                acc[m][0] = csub(acc[m][0] + mont_mul_lazy(g.x, h[m].x, q, ninv), q2);
                acc[m][1] = csub(acc[m][1] + mont_mul_lazy(g.y, h[m].y, q, ninv), q2);
                x0 = csub(x0 + mont_mul_lazy(k[m].x, h[m].x, q, ninv), q2);
                x1 = csub(x1 + mont_mul_lazy(k[m].y, h[m].y, q, ninv), q2);
            }
            x0 = csub(x0, q); x1 = csub(x1, q);
            if (a.xmform) { x0 = mont_mul(x0, md.r2, q, ninv); x1 = mont_mul(x1, md.r2, q, ninv); }
            u64x2 r; r.x = x0; r.y = x1;
            *(u64x2*)(xo + i * ds) = r;
        }
    } else {
#pragma unroll 2
    for (int i = 0; i < a.nb; ++i) {
        const u64x2 g = once ? ld_stream(bg + i * ds) : ld_cached(bg + i * ds);
        u64x2 h[G];
#pragma unroll
        for (int m = 0; m < G; ++m) h[m] = ld_stream(ah[m] + i * ds);
#pragma unroll
        for (int m = 0; m < G; ++m) {
            acc[m][0] = csub(acc[m][0] + mont_mul_lazy(g.x, h[m].x, q, ninv), q2);
            acc[m][1] = csub(acc[m][1] + mont_mul_lazy(g.y, h[m].y, q, ninv), q2);
        }
    }
    }
    if (!a.xmulti && ka->ah2[leader]) {          // second gadget (mkbfv: the QMul digits), same shape
        const u64* ah2[G];
#pragma unroll
        for (int m = 0; m < G; ++m) ah2[m] = ka->ah2[idx[m]] + off;
        const u64* bg2 = ka->bg2[leader] + off;
        if (a.xout2) {
            const u64* xk[G];
#pragma unroll
            for (int m = 0; m < G; ++m) xk[m] = ka->xkey2[idx[m]] + off;
            u64* xo = a.xout2 + off;
#pragma unroll 1
            for (int i = 0; i < a.nb; ++i) {
                const u64x2 g = once ? ld_stream(bg2 + i * ds) : ld_cached(bg2 + i * ds);
                u64x2 h[G], k[G];
#pragma unroll
                for (int m = 0; m < G; ++m) { h[m] = ld_stream(ah2[m] + i * ds); k[m] = ld_stream(xk[m] + i * ds); }
                u64 x0 = 0, x1 = 0;
#pragma unroll
                for (int m = 0; m < G; ++m) {
                    acc[m][0] = csub(acc[m][0] + mont_mul_lazy(g.x, h[m].x, q, ninv), q2);
                    acc[m][1] = csub(acc[m][1] + mont_mul_lazy(g.y, h[m].y, q, ninv), q2);
                    x0 = csub(x0 + mont_mul_lazy(k[m].x, h[m].x, q, ninv), q2);
                    x1 = csub(x1 + mont_mul_lazy(k[m].y, h[m].y, q, ninv), q2);
                }
                x0 = csub(x0, q); x1 = csub(x1, q);
                if (a.xmform) { x0 = mont_mul(x0, md.r2, q, ninv); x1 = mont_mul(x1, md.r2, q, ninv); }
                u64x2 r; r.x = x0; r.y = x1;
                *(u64x2*)(xo + i * ds) = r;
            }
        } else {
#pragma unroll 2
        for (int i = 0; i < a.nb; ++i) {
            const u64x2 g = once ? ld_stream(bg2 + i * ds) : ld_cached(bg2 + i * ds);
            u64x2 h[G];
#pragma unroll
            for (int m = 0; m < G; ++m) h[m] = ld_stream(ah2[m] + i * ds);
#pragma unroll
            for (int m = 0; m < G; ++m) {
                acc[m][0] = csub(acc[m][0] + mont_mul_lazy(g.x, h[m].x, q, ninv), q2);
                acc[m][1] = csub(acc[m][1] + mont_mul_lazy(g.y, h[m].y, q, ninv), q2);
            }
        }
        }
    }
#pragma unroll
    for (int m = 0; m < G; ++m) ext_store(a.c1 + (long)idx[m] * a.c1_item + off, acc[m][0], acc[m][1], q);
}
// G pairs (item, item + 1) that share ah within the pair and bg[item + 1] (the CRS u) across the pairs
template <int G>
__device__ __forceinline__ void ext_group_pairs(const ExtInnerArgs& a, ext_kargs ka, int leader, long off, const Mod& md) {
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const long ds = a.digit_stride;
    const u64* ah[G]; const u64* bg[G]; int idx[G];
    int it = leader;
#pragma unroll
    for (int m = 0; m < G; ++m) { idx[m] = it; ah[m] = ka->ah[it] + off; bg[m] = ka->bg[it] + off; it = ka->gnext[it]; }
    const u64* bgn = ka->bg[leader + 1] + off;
    const bool once_n = ka->bg_once[leader + 1] != 0;
    u64 av[G][2], bv[G][2];
#pragma unroll
    for (int m = 0; m < G; ++m) { av[m][0] = av[m][1] = bv[m][0] = bv[m][1] = 0; }
#pragma unroll 1
    for (int i = 0; i < a.nb; ++i) {
        const u64x2 gn = once_n ? ld_stream(bgn + i * ds) : ld_cached(bgn + i * ds);
        u64x2 h[G], g[G];
#pragma unroll
        for (int m = 0; m < G; ++m) { h[m] = ld_stream(ah[m] + i * ds); g[m] = ld_stream(bg[m] + i * ds); }
#pragma unroll
        for (int m = 0; m < G; ++m) {
            av[m][0] = csub(av[m][0] + mont_mul_lazy(g[m].x, h[m].x, q, ninv), q2); av[m][1] = csub(av[m][1] + mont_mul_lazy(g[m].y, h[m].y, q, ninv), q2);
            bv[m][0] = csub(bv[m][0] + mont_mul_lazy(gn.x, h[m].x, q, ninv), q2); bv[m][1] = csub(bv[m][1] + mont_mul_lazy(gn.y, h[m].y, q, ninv), q2);
        }
    }
#pragma unroll
    for (int m = 0; m < G; ++m) {
        u64* out = a.c1 + (long)idx[m] * a.c1_item + off;
        ext_store(out, av[m][0], av[m][1], q);
        ext_store(out + a.c1_item, bv[m][0], bv[m][1], q);
    }
}
// an item (or a pair of neighbours that share their digits) per thread: launches without groups (few registers, full occupancy)
__global__ void __launch_bounds__(PW_THREADS) ext_inner_kernel(ExtInnerArgs a) {
    const int s = blockIdx.y, item = blockIdx.z;
    const int role = a.pair[item];
    if (role == 2) return;                    // computed by its leader
    const int n = 2 * (blockIdx.x * PW_THREADS + threadIdx.x);
    if (n >= a.N) return;
    const int m = a.map[s];
    if (role == 3) {                          // computed before (step E inside a batch's F1 kernel): into the slot
        *(u64x2*)(a.c1 + (long)item * a.c1_item + (long)m * a.N + n) = ld_stream(a.bg[item] + (long)m * a.N + n);
        return;
    }
    const Mod md = a.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const u64* ah = a.ah[item] + (long)m * a.N + n;
    const u64* bg = a.bg[item] + (long)m * a.N + n;
    u64* out = a.c1 + (long)item * a.c1_item + (long)m * a.N + n;
    const long ds = a.digit_stride;
    const bool once = a.bg_once[item] != 0;
    u64x2 r;
    if (role == 1) {
        const u64* bgn = a.bg[item + 1] + (long)m * a.N + n;
        u64 a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll 4
        for (int i = 0; i < a.nb; ++i) {
            const u64x2 h = ld_stream(ah + i * ds), g = once ? ld_stream(bg + i * ds) : ld_cached(bg + i * ds), gn = ld_cached(bgn + i * ds);
            a0 = csub(a0 + mont_mul_lazy(g.x, h.x, q, ninv), q2); a1 = csub(a1 + mont_mul_lazy(g.y, h.y, q, ninv), q2);
            b0 = csub(b0 + mont_mul_lazy(gn.x, h.x, q, ninv), q2); b1 = csub(b1 + mont_mul_lazy(gn.y, h.y, q, ninv), q2);
        }
        r.x = csub(a0, q); r.y = csub(a1, q);
        *(u64x2*)out = r;
        r.x = csub(b0, q); r.y = csub(b1, q);
        *(u64x2*)(out + a.c1_item) = r;
        return;
    }
    u64 a0 = 0, a1 = 0;
#pragma unroll 4
    for (int i = 0; i < a.nb; ++i) {
        const u64x2 h = ld_stream(ah + i * ds), g = once ? ld_stream(bg + i * ds) : ld_cached(bg + i * ds);
        a0 = csub(a0 + mont_mul_lazy(g.x, h.x, q, ninv), q2); a1 = csub(a1 + mont_mul_lazy(g.y, h.y, q, ninv), q2);
    }
    if (a.ah2[item]) {
        const u64* ah2 = a.ah2[item] + (long)m * a.N + n;
        const u64* bg2 = a.bg2[item] + (long)m * a.N + n;
#pragma unroll 4
        for (int i = 0; i < a.nb; ++i) {
            const u64x2 h = *(const u64x2*)(ah2 + i * ds), g = *(const u64x2*)(bg2 + i * ds);
            a0 = csub(a0 + mont_mul_lazy(g.x, h.x, q, ninv), q2); a1 = csub(a1 + mont_mul_lazy(g.y, h.y, q, ninv), q2);
        }
    }
    r.x = csub(a0, q); r.y = csub(a1, q);
    *(u64x2*)out = r;
}
// launches with groups: the leaders run the grouped forms above, everything else the per-item form
__global__ void __launch_bounds__(PW_THREADS) ext_inner_group_kernel(ExtInnerArgs a) {
    ext_kargs ka = (ext_kargs)__builtin_amdgcn_kernarg_segment_ptr();
    const int s = blockIdx.y, item = blockIdx.z;
    const int role = ka->pair[item], grp = ka->grp[item];
    if (role == 2 || grp == 2) return;                    // computed by its (pair / group) leader
    const int n = 2 * (blockIdx.x * PW_THREADS + threadIdx.x);
    if (n >= a.N) return;
    const int m = a.map[s];
    if (role == 3) {                                      // computed before (step E inside a batch's F1 kernel): into the slot
        *(u64x2*)(a.c1 + (long)item * a.c1_item + (long)m * a.N + n) = ld_stream(ka->bg[item] + (long)m * a.N + n);
        return;
    }
    const Mod md = a.mods[m];
    if (grp == 1) {
        const long off = (long)m * a.N + n;
        int G = 1;
        for (int it = ka->gnext[item]; it != 255; it = ka->gnext[it]) ++G;
        if (role == 1) { if (G == 2) ext_group_pairs<2>(a, ka, item, off, md); else if (G == 3) ext_group_pairs<3>(a, ka, item, off, md); else ext_group_pairs<4>(a, ka, item, off, md); }
        else { if (G == 1) ext_group_singles<1>(a, ka, item, off, md); else if (G == 2) ext_group_singles<2>(a, ka, item, off, md); else if (G == 3) ext_group_singles<3>(a, ka, item, off, md); else ext_group_singles<4>(a, ka, item, off, md); }
        return;
    }
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const u64* ah = a.ah[item] + (long)m * a.N + n;
    const u64* bg = a.bg[item] + (long)m * a.N + n;
    u64* out = a.c1 + (long)item * a.c1_item + (long)m * a.N + n;
    const long ds = a.digit_stride;
    const bool once = a.bg_once[item] != 0;
    u64x2 r;
    if (role == 1) {
        const u64* bgn = a.bg[item + 1] + (long)m * a.N + n;
        u64 a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll 4
        for (int i = 0; i < a.nb; ++i) {
            const u64x2 h = ld_stream(ah + i * ds), g = once ? ld_stream(bg + i * ds) : ld_cached(bg + i * ds), gn = ld_cached(bgn + i * ds);
            a0 = csub(a0 + mont_mul_lazy(g.x, h.x, q, ninv), q2); a1 = csub(a1 + mont_mul_lazy(g.y, h.y, q, ninv), q2);
            b0 = csub(b0 + mont_mul_lazy(gn.x, h.x, q, ninv), q2); b1 = csub(b1 + mont_mul_lazy(gn.y, h.y, q, ninv), q2);
        }
        r.x = csub(a0, q); r.y = csub(a1, q);
        *(u64x2*)out = r;
        r.x = csub(b0, q); r.y = csub(b1, q);
        *(u64x2*)(out + a.c1_item) = r;
        return;
    }
    u64 a0 = 0, a1 = 0;
#pragma unroll 4
    for (int i = 0; i < a.nb; ++i) {
        const u64x2 h = ld_stream(ah + i * ds), g = once ? ld_stream(bg + i * ds) : ld_cached(bg + i * ds);
        a0 = csub(a0 + mont_mul_lazy(g.x, h.x, q, ninv), q2); a1 = csub(a1 + mont_mul_lazy(g.y, h.y, q, ninv), q2);
    }
    if (a.ah2[item]) {
        const u64* ah2 = a.ah2[item] + (long)m * a.N + n;
        const u64* bg2 = a.bg2[item] + (long)m * a.N + n;
#pragma unroll 4
        for (int i = 0; i < a.nb; ++i) {
            const u64x2 h = *(const u64x2*)(ah2 + i * ds), g = *(const u64x2*)(bg2 + i * ds);
            a0 = csub(a0 + mont_mul_lazy(g.x, h.x, q, ninv), q2); a1 = csub(a1 + mont_mul_lazy(g.y, h.y, q, ninv), q2);
        }
    }
    r.x = csub(a0, q); r.y = csub(a1, q);
    *(u64x2*)out = r;
}
// Step F1 of a MulAndRelin with FIVE TO SIXTEEN parties (round 3): all items share bg = y and every item's digits meet its own key d_i as well, so
// that x = sum_i d_i (.) h(c0_i) comes out of the same pass over the h(c0_i) as the external products <h(c0_i), y> -- what ext_group_singles<G> does
// for up to four parties, here with the members of a digit taken four at a time (8, 12 or 16 accumulator pairs, eight loads in flight).  Without it
// x is a separate inner_product_kernel launch that reads the 8 x 331 MB of hoisted digits of an 8-party PN16QP1761 MulRelin a second time.
template <int G4>
__device__ __forceinline__ void ext_x_wide(const ExtInnerArgs& a, ext_kargs ka, long off, const Mod& md) {
    constexpr int GM = 4 * G4;                  // accumulators held; the launch has G = a.nitems <= GM members (wave-uniform guards)
    const int G = a.nitems;
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const long ds = a.digit_stride;
    const u64* bg = ka->bg[0] + off;
    u64* xo = a.xout + off;
    u64 acc[GM][2];
#pragma unroll
    for (int m = 0; m < GM; ++m) { acc[m][0] = 0; acc[m][1] = 0; }
#pragma unroll 1
    for (int i = 0; i < a.nb; ++i) {
        const long io = off + i * ds;
        const u64x2 g = ld_stream(bg + i * ds);
        u64 x0 = 0, x1 = 0;
#pragma unroll
        for (int m0 = 0; m0 < GM; m0 += 4) {
            if (m0 < G) {
                u64x2 h[4], k[4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (m0 + m < G) { h[m] = ld_stream(ka->ah[m0 + m] + io); k[m] = ld_stream(ka->xkey[m0 + m] + io); }
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    if (m0 + m < G) {
                        acc[m0 + m][0] = csub(acc[m0 + m][0] + mont_mul_lazy(g.x, h[m].x, q, ninv), q2);
                        acc[m0 + m][1] = csub(acc[m0 + m][1] + mont_mul_lazy(g.y, h[m].y, q, ninv), q2);
                        x0 = csub(x0 + mont_mul_lazy(k[m].x, h[m].x, q, ninv), q2);
                        x1 = csub(x1 + mont_mul_lazy(k[m].y, h[m].y, q, ninv), q2);
                    }
            }
        }
        x0 = csub(x0, q); x1 = csub(x1, q);
        if (a.xmform) { x0 = mont_mul(x0, md.r2, q, ninv); x1 = mont_mul(x1, md.r2, q, ninv); }
        u64x2 r; r.x = x0; r.y = x1;
        *(u64x2*)(xo + i * ds) = r;
    }
#pragma unroll
    for (int m = 0; m < GM; ++m)
        if (m < G) ext_store(a.c1 + (long)m * a.c1_item + off, acc[m][0], acc[m][1], q);
}
template <int G4>
__global__ void __launch_bounds__(PW_THREADS) ext_inner_xwide_kernel(ExtInnerArgs a) {
    ext_kargs ka = (ext_kargs)__builtin_amdgcn_kernarg_segment_ptr();
    const int n = 2 * (blockIdx.x * PW_THREADS + threadIdx.x);
    if (n >= a.N) return;
    const int m = a.map[blockIdx.y];
    const Mod md = a.mods[m];
    ext_x_wide<G4>(a, ka, (long)m * a.N + n, md);
}
// Step F1 of a single-device MulAndRelin with as many parties in op1 as in op0 (G <= 4): t_i = <h(c0_i), y>, x = MForm(sum_i d_i (.) h(c0_i)) AND
// y = MForm(sum_j b_j (.) h(c1_j)) in ONE pass -- the thread that forms the products at a coefficient needs y there and nowhere else, so y is neither
// a launch of its own nor written and read back (2 x 59 MB at PN15QP880).  Per digit and thread: 4 G sixteen-byte loads, y[d] in registers, x[d]
// stored.  Same operations as inner_product_kernel (mform_out) + ext_group_singles<G> with the x by-product: the same integers.
// one gadget's pass of ext_inner_xy_kernel: per digit y[d] (G1 terms), x[d] and the F1 products (G0 items) and (E) the G1 step-E products
template <int G0, int G1, bool E>
__device__ __forceinline__ void xy_gadget(const u64* const (&ah)[4], const u64* const (&xkey)[4], const u64* const (&ykey)[4], const u64* const (&yh)[4], u64* xout,
                                          u64 (&acc)[G0][2], u64 (&ace)[E ? G1 : 1][2], const long off, const long ds, const int nb, const Mod& md) {
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
#pragma unroll 1
    for (int i = 0; i < nb; ++i) {
        u64x2 h[G0], k[G0], b[G1], c[G1];
#pragma unroll
        for (int g = 0; g < G1; ++g) { b[g] = ld_stream(ykey[g] + off + i * ds); c[g] = ld_stream(yh[g] + off + i * ds); }
#pragma unroll
        for (int g = 0; g < G0; ++g) { h[g] = ld_stream(ah[g] + off + i * ds); k[g] = ld_stream(xkey[g] + off + i * ds); }
        u64 y0 = 0, y1 = 0;
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            y0 = csub(y0 + mont_mul_lazy(b[g].x, c[g].x, q, ninv), q2);
            y1 = csub(y1 + mont_mul_lazy(b[g].y, c[g].y, q, ninv), q2);
        }
        y0 = mont_mul(csub(y0, q), md.r2, q, ninv); y1 = mont_mul(csub(y1, q), md.r2, q, ninv);
        u64 x0 = 0, x1 = 0;
#pragma unroll
        for (int g = 0; g < G0; ++g) {
            acc[g][0] = csub(acc[g][0] + mont_mul_lazy(y0, h[g].x, q, ninv), q2);
            acc[g][1] = csub(acc[g][1] + mont_mul_lazy(y1, h[g].y, q, ninv), q2);
            x0 = csub(x0 + mont_mul_lazy(k[g].x, h[g].x, q, ninv), q2);
            x1 = csub(x1 + mont_mul_lazy(k[g].y, h[g].y, q, ninv), q2);
        }
        x0 = mont_mul(csub(x0, q), md.r2, q, ninv); x1 = mont_mul(csub(x1, q), md.r2, q, ninv);
        if constexpr (E) {
            // step E from what the thread holds: <h(c1_j), x> (ext_inner_kernel: acc += mont_mul_lazy(x[d], h(c1_j)[d]))
#pragma unroll
            for (int g = 0; g < G1; ++g) {
                ace[g][0] = csub(ace[g][0] + mont_mul_lazy(x0, c[g].x, q, ninv), q2);
                ace[g][1] = csub(ace[g][1] + mont_mul_lazy(x1, c[g].y, q, ninv), q2);
            }
        }
        if (xout) { u64x2 r; r.x = x0; r.y = x1; *(u64x2*)(xout + off + i * ds) = r; }
    }
}
template <int G0, int G1, bool E>
__global__ void __launch_bounds__(PW_THREADS) ext_inner_xy_kernel(ExtXyArgs a) {
    const int s = blockIdx.y;
    const int n = 2 * (blockIdx.x * PW_THREADS + threadIdx.x);
    if (n >= a.N) return;
    const int m = a.map[s];
    const Mod md = a.mods[m];
    const long off = (long)m * a.N + n;
    u64 acc[G0][2], ace[E ? G1 : 1][2];
#pragma unroll
    for (int g = 0; g < G0; ++g) { acc[g][0] = 0; acc[g][1] = 0; }
    if constexpr (E) {
#pragma unroll
        for (int g = 0; g < G1; ++g) { ace[g][0] = 0; ace[g][1] = 0; }
    }
    xy_gadget<G0, G1, E>(a.ah, a.xkey, a.ykey, a.yh, a.xout, acc, ace, off, a.digit_stride, a.nb, md);
    // mkbfv: the second gadget (QMul digits h2, keys d2 / b2, sums x2 / y2) adds its products to the same sums (keyswitch_hoisted.go:20-28)
    if (a.ah2[0]) xy_gadget<G0, G1, E>(a.ah2, a.xkey2, a.ykey2, a.yh2, a.xout2, acc, ace, off, a.digit_stride, a.nb, md);
#pragma unroll
    for (int g = 0; g < G0; ++g) ext_store(a.c1 + (long)g * a.c1_item + off, acc[g][0], acc[g][1], md.q);
    if constexpr (E) {
#pragma unroll
        for (int g = 0; g < G1; ++g) ext_store(a.e_out + (long)g * a.c1_item + off, ace[g][0], ace[g][1], md.q);
    }
}
template <int G0, int G1> void launch_xy_e(const ExtXyArgs& a, dim3 grid, dim3 blk, hipStream_t st) {
    if (a.e_out) hipLaunchKernelGGL((ext_inner_xy_kernel<G0, G1, true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((ext_inner_xy_kernel<G0, G1, false>), grid, blk, 0, st, a);
}
template <int G0> void launch_xy_g1(const ExtXyArgs& a, dim3 grid, dim3 blk, hipStream_t st) {
    switch (a.g1) { case 1: launch_xy_e<G0, 1>(a, grid, blk, st); break; case 2: launch_xy_e<G0, 2>(a, grid, blk, st); break;
                    case 3: launch_xy_e<G0, 3>(a, grid, blk, st); break; default: launch_xy_e<G0, 4>(a, grid, blk, st); break; }
}
void launch_ext_inner_xy(const ExtXyArgs& a, hipStream_t st) {
    if (a.g < 1 || a.g > 4 || a.g1 < 1 || a.g1 > 4) throw std::runtime_error("mkhe: internal: ext_inner_xy_kernel takes one to four parties per operand");
    const int bx = (a.N / 2 + PW_THREADS - 1) / PW_THREADS;
    const dim3 grid(bx, a.nslots, 1), blk(PW_THREADS);
    switch (a.g) { case 1: launch_xy_g1<1>(a, grid, blk, st); break; case 2: launch_xy_g1<2>(a, grid, blk, st); break;
                   case 3: launch_xy_g1<3>(a, grid, blk, st); break; default: launch_xy_g1<4>(a, grid, blk, st); break; }
}
// Five to eight parties per operand (PN16QP1761 with 8 parties: y was a 0.95 ms inner_product_kernel<8> launch, step E a second read of 8 x 331 MB of
// digits): the same pass with the 4 G loads of a digit taken four at a time -- the h(c1_j)[d] stay in registers for step E (8 x 4 VGPRs), the keys
// and the h(c0_i)[d] pass through.
typedef const __attribute__((address_space(4))) ExtXyWideArgs* xyw_kargs;
template <int G, bool E>
__global__ void __launch_bounds__(PW_THREADS) ext_inner_xy_wide_kernel(ExtXyWideArgs a) {
    xyw_kargs ka = (xyw_kargs)__builtin_amdgcn_kernarg_segment_ptr();
    const int s = blockIdx.y;
    const int n = 2 * (blockIdx.x * PW_THREADS + threadIdx.x);
    if (n >= a.N) return;
    const int m = a.map[s];
    const Mod md = a.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const long off = (long)m * a.N + n, ds = a.digit_stride;
    u64 acc[G][2], ace[E ? G : 1][2];
#pragma unroll
    for (int g = 0; g < G; ++g) { acc[g][0] = 0; acc[g][1] = 0; if (E) { ace[g][0] = 0; ace[g][1] = 0; } }
#pragma unroll 1
    for (int i = 0; i < a.nb; ++i) {
        const long o = off + i * ds;
        u64x2 c[G];
        u64 y0 = 0, y1 = 0;
#pragma unroll
        for (int g0 = 0; g0 < G; g0 += 4) {
            u64x2 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) if (g0 + j < G) { b[j] = ld_stream(ka->ykey[g0 + j] + o); c[g0 + j] = ld_stream(ka->yh[g0 + j] + o); }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (g0 + j < G) {
                y0 = csub(y0 + mont_mul_lazy(b[j].x, c[g0 + j].x, q, ninv), q2);
                y1 = csub(y1 + mont_mul_lazy(b[j].y, c[g0 + j].y, q, ninv), q2);
            }
        }
        y0 = mont_mul(csub(y0, q), md.r2, q, ninv); y1 = mont_mul(csub(y1, q), md.r2, q, ninv);
        u64 x0 = 0, x1 = 0;
#pragma unroll
        for (int g0 = 0; g0 < G; g0 += 4) {
            u64x2 h[4], k[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) if (g0 + j < G) { h[j] = ld_stream(ka->ah[g0 + j] + o); k[j] = ld_stream(ka->xkey[g0 + j] + o); }
#pragma unroll
            for (int j = 0; j < 4; ++j) if (g0 + j < G) {
                acc[g0 + j][0] = csub(acc[g0 + j][0] + mont_mul_lazy(y0, h[j].x, q, ninv), q2);
                acc[g0 + j][1] = csub(acc[g0 + j][1] + mont_mul_lazy(y1, h[j].y, q, ninv), q2);
                x0 = csub(x0 + mont_mul_lazy(k[j].x, h[j].x, q, ninv), q2);
                x1 = csub(x1 + mont_mul_lazy(k[j].y, h[j].y, q, ninv), q2);
            }
        }
        x0 = mont_mul(csub(x0, q), md.r2, q, ninv); x1 = mont_mul(csub(x1, q), md.r2, q, ninv);
        if constexpr (E) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                ace[g][0] = csub(ace[g][0] + mont_mul_lazy(x0, c[g].x, q, ninv), q2);
                ace[g][1] = csub(ace[g][1] + mont_mul_lazy(x1, c[g].y, q, ninv), q2);
            }
        }
        if (a.xout) { u64x2 r; r.x = x0; r.y = x1; *(u64x2*)(a.xout + o) = r; }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        ext_store(a.c1 + (long)g * a.c1_item + off, acc[g][0], acc[g][1], q);
        if constexpr (E) ext_store(a.e_out + (long)g * a.c1_item + off, ace[g][0], ace[g][1], q);
    }
}
void launch_ext_inner_xy_wide(const ExtXyWideArgs& a, hipStream_t st) {
    if (a.g < 5 || a.g > 8) throw std::runtime_error("mkhe: internal: ext_inner_xy_wide_kernel takes five to eight parties");
    const int bx = (a.N / 2 + PW_THREADS - 1) / PW_THREADS;
    const dim3 grid(bx, a.nslots, 1), blk(PW_THREADS);
#define MKHE_XYW(GV) do { if (a.e_out) hipLaunchKernelGGL((ext_inner_xy_wide_kernel<GV, true>), grid, blk, 0, st, a); \
                          else hipLaunchKernelGGL((ext_inner_xy_wide_kernel<GV, false>), grid, blk, 0, st, a); } while (0)
    switch (a.g) { case 5: MKHE_XYW(5); break; case 6: MKHE_XYW(6); break; case 7: MKHE_XYW(7); break; default: MKHE_XYW(8); break; }
#undef MKHE_XYW
}
typedef const __attribute__((address_space(4))) ExtXyBatchArgs* xyb_kargs;
template <int G0, int G1, bool E>
__global__ void __launch_bounds__(PW_THREADS) ext_inner_xy_batch_kernel(ExtXyBatchArgs a) {
    xyb_kargs ka = (xyb_kargs)__builtin_amdgcn_kernarg_segment_ptr();      // per-input pointer lists: scalar loads
    const int s = blockIdx.y, bi = blockIdx.z;
    const int n = 2 * (blockIdx.x * PW_THREADS + threadIdx.x);
    if (n >= a.N) return;
    const int m = a.map[s];
    const Mod md = a.mods[m];
    const u64 q = md.q, q2 = md.q2;
    const u32 ninv = md.ninv32;
    const long off = (long)m * a.N + n, ds = a.digit_stride;
    const u64* ah[G0]; const u64* xk[G0]; const u64* yh[G1]; const u64* yk[G1];
#pragma unroll
    for (int g = 0; g < G0; ++g) { ah[g] = ka->ah[bi][g] + off; xk[g] = ka->xkey[g] + off; }
#pragma unroll
    for (int g = 0; g < G1; ++g) { yh[g] = ka->yh[bi][g] + off; yk[g] = ka->ykey[g] + off; }
    u64* xo = E ? nullptr : ka->xout[bi] + off;
    u64 acc[G0][2], ace[E ? G1 : 1][2];
#pragma unroll
    for (int g = 0; g < G0; ++g) { acc[g][0] = 0; acc[g][1] = 0; }
    if constexpr (E) {
#pragma unroll
        for (int g = 0; g < G1; ++g) { ace[g][0] = 0; ace[g][1] = 0; }
    }
#pragma unroll 1
    for (int i = 0; i < a.nb; ++i) {
        u64x2 h[G0], k[G0], b[G1], c[G1];
#pragma unroll
        for (int g = 0; g < G1; ++g) { b[g] = ld_cached(yk[g] + i * ds); c[g] = ld_stream(yh[g] + i * ds); }          // (the keys are re-read by every input)
#pragma unroll
        for (int g = 0; g < G0; ++g) { h[g] = ld_stream(ah[g] + i * ds); k[g] = ld_cached(xk[g] + i * ds); }
        u64 y0 = 0, y1 = 0;
#pragma unroll
        for (int g = 0; g < G1; ++g) {
            y0 = csub(y0 + mont_mul_lazy(b[g].x, c[g].x, q, ninv), q2);
            y1 = csub(y1 + mont_mul_lazy(b[g].y, c[g].y, q, ninv), q2);
        }
        y0 = mont_mul(csub(y0, q), md.r2, q, ninv); y1 = mont_mul(csub(y1, q), md.r2, q, ninv);
        u64 x0 = 0, x1 = 0;
#pragma unroll
        for (int g = 0; g < G0; ++g) {
            acc[g][0] = csub(acc[g][0] + mont_mul_lazy(y0, h[g].x, q, ninv), q2);
            acc[g][1] = csub(acc[g][1] + mont_mul_lazy(y1, h[g].y, q, ninv), q2);
            x0 = csub(x0 + mont_mul_lazy(k[g].x, h[g].x, q, ninv), q2);
            x1 = csub(x1 + mont_mul_lazy(k[g].y, h[g].y, q, ninv), q2);
        }
        x0 = mont_mul(csub(x0, q), md.r2, q, ninv); x1 = mont_mul(csub(x1, q), md.r2, q, ninv);
        if constexpr (E) {
#pragma unroll
            for (int g = 0; g < G1; ++g) {
                ace[g][0] = csub(ace[g][0] + mont_mul_lazy(x0, c[g].x, q, ninv), q2);
                ace[g][1] = csub(ace[g][1] + mont_mul_lazy(x1, c[g].y, q, ninv), q2);
            }
        } else {
            u64x2 r; r.x = x0; r.y = x1;
            *(u64x2*)(xo + i * ds) = r;
        }
    }
#pragma unroll
    for (int g = 0; g < G0; ++g) ext_store(a.c1 + (long)(bi * G0 + g) * a.c1_item + off, acc[g][0], acc[g][1], q);
    if constexpr (E) {
        u64* eo = ka->eout[bi] + off;
#pragma unroll
        for (int g = 0; g < G1; ++g) ext_store(eo + (long)g * a.c1_item, ace[g][0], ace[g][1], q);
    }
}
template <int G0, int G1> void launch_xyb_e(const ExtXyBatchArgs& a, dim3 grid, dim3 blk, hipStream_t st) {
    if (a.eout[0]) hipLaunchKernelGGL((ext_inner_xy_batch_kernel<G0, G1, true>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((ext_inner_xy_batch_kernel<G0, G1, false>), grid, blk, 0, st, a);
}
template <int G0> void launch_xyb_g1(const ExtXyBatchArgs& a, dim3 grid, dim3 blk, hipStream_t st) {
    switch (a.g1) { case 1: launch_xyb_e<G0, 1>(a, grid, blk, st); break; case 2: launch_xyb_e<G0, 2>(a, grid, blk, st); break;
                    case 3: launch_xyb_e<G0, 3>(a, grid, blk, st); break; default: launch_xyb_e<G0, 4>(a, grid, blk, st); break; }
}
void launch_ext_inner_xy_batch(const ExtXyBatchArgs& a, hipStream_t st) {
    if (a.g < 1 || a.g > 4 || a.g1 < 1 || a.g1 > 4 || a.nbatch < 1 || a.nbatch > XYB_MAX) throw std::runtime_error("mkhe: internal: ext_inner_xy_batch_kernel out of its range");
    const int bx = (a.N / 2 + PW_THREADS - 1) / PW_THREADS;
    const dim3 grid(bx, a.nslots, a.nbatch), blk(PW_THREADS);
    switch (a.g) { case 1: launch_xyb_g1<1>(a, grid, blk, st); break; case 2: launch_xyb_g1<2>(a, grid, blk, st); break;
                   case 3: launch_xyb_g1<3>(a, grid, blk, st); break; default: launch_xyb_g1<4>(a, grid, blk, st); break; }
}
void launch_ext_inner(const ExtInnerArgs& a_in, hipStream_t st) {
    ExtInnerArgs a = a_in;
    if (a.xout && !a.xmulti && a.nitems > 4) {
        // (checked by the caller: single items, one shared key, one gadget, five to sixteen of them)
        if (a.nitems > 16 || a.xout2) throw std::runtime_error("mkhe: internal: wide x by-product outside its range");
        const int bx = (a.N / 2 + PW_THREADS - 1) / PW_THREADS;
        const dim3 grid(bx, a.nslots, 1), blk(PW_THREADS);
        if (a.nitems <= 8) hipLaunchKernelGGL(ext_inner_xwide_kernel<2>, grid, blk, 0, st, a);
        else if (a.nitems <= 12) hipLaunchKernelGGL(ext_inner_xwide_kernel<3>, grid, blk, 0, st, a);
        else hipLaunchKernelGGL(ext_inner_xwide_kernel<4>, grid, blk, 0, st, a);
        return;
    }
    static const int grouping = MKHE_AB_INT("MKHE_EXT_GROUP", 1);
    for (int i = 0; i < a.nitems; ++i) { a.grp[i] = 0; a.gnext[i] = 255; }
    if (a.xout && !a.xmulti) {
        // the x by-product needs every item in ONE group (checked by the caller: single items, one shared key, at most four)
        for (int i = 0; i < a.nitems; ++i) { a.grp[i] = i ? 2 : 1; a.gnext[i] = i + 1 < a.nitems ? (unsigned char)(i + 1) : 255; }
        a.bg_once[0] = 1;
    } else if (grouping || a.xmulti) {
        // (xmulti: the items that share a key are one input's step F1 -- at most four, checked by the caller -- and form a group even alone:
        // the group form is the one that carries the x by-product)
        for (int i = 0; i < a.nitems; ++i) {
            if (a.grp[i] || a.pair[i] >= 2) continue;
            const bool pr = a.pair[i] == 1;            // (pairs never carry a second gadget: see Context::ext_front)
            int last = i, cnt = 1;
            for (int k = i + 1; k < a.nitems && cnt < 4; ++k) {
                if (a.grp[k] || a.pair[k] != a.pair[i] || (a.ah2[k] != nullptr) != (a.ah2[i] != nullptr)) continue;
                if (pr ? (a.bg[k + 1] != a.bg[i + 1]) : (a.bg[k] != a.bg[i] || a.bg2[k] != a.bg2[i])) continue;
                a.gnext[last] = (unsigned char)k; a.grp[k] = 2; last = k; ++cnt;
            }
            if (cnt > 1 || a.xmulti) {
                a.grp[i] = 1;
                // the shared operand is read once per coefficient by this group: stream it past the caches when no other group uses it
                const u64* sh = pr ? a.bg[i + 1] : a.bg[i];
                int uses = 0;
                for (int k = 0; k < a.nitems; ++k) uses += (a.bg[k] == sh);
                if (uses == cnt) a.bg_once[pr ? i + 1 : i] = 1;
            }
        }
    }
    const int bx = (a.N / 2 + PW_THREADS - 1) / PW_THREADS;
    bool any = false;
    for (int i = 0; i < a.nitems; ++i) any = any || a.grp[i] == 1;
    if (any) hipLaunchKernelGGL(ext_inner_group_kernel, dim3(bx, a.nslots, a.nitems), dim3(PW_THREADS), 0, st, a);
    else hipLaunchKernelGGL(ext_inner_kernel, dim3(bx, a.nslots, a.nitems), dim3(PW_THREADS), 0, st, a);
}

typedef const __attribute__((address_space(4))) ModDownBatchArgs* mdb_kargs;
__global__ void __launch_bounds__(PW_THREADS) moddown_batch_kernel(ModDownBatchArgs a) {
    mdb_kargs ka = (mdb_kargs)__builtin_amdgcn_kernarg_segment_ptr();
    const int n = blockIdx.x * PW_THREADS + threadIdx.x;
    if (n >= a.N) return;
    const int g = blockIdx.z;
    for (int k = ka->gstart[g]; k < ka->gstart[g + 1]; ++k) {
        const int item = ka->order[k];
        const u64* xq = a.c1 + (long)item * a.c1_item;
        const u64* xp = xq + a.p_offset;
        u64* dst = ka->dst[item];
        const int acc = ka->accumulate[item];
        const u64 gal = ka->gal_v[item] ? (u64)ka->gal_v[item] : a.galEl;
        const u64* post = (k == ka->gstart[g + 1] - 1) ? ka->post[item] : nullptr;
        u64 y[MAXP];
        double vi = 0.0;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            if (i < a.np) {
                const Mod mp = load_mod((sc_mod)a.mods_p + i);
                y[i] = mont_mul(xp[(long)i * a.N + n], ((sc_u64)a.t.qoverqiinvqi)[i], mp.q, mp.ninv32);
                vi = vi + (double)y[i] / (double)mp.q;
            }
        }
        const u64 v = (u64)vi;
        const int nj = a.qlist ? a.nqlist : a.level + 1;
        for (int jj = blockIdx.y; jj < nj; jj += gridDim.y) {
            const int j = a.qlist ? ((sc_int)a.qlist)[jj] : jj;
            const Mod mq = load_mod((sc_mod)a.mods_q + j);
            u64 rlo = 0, rhi = 0;
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                if (i < a.np) {
                    u64 mhi, mlo;
                    mul64x64(y[i], ((sc_u64)a.t.qoverqimodp)[(long)j * a.np + i], mhi, mlo);
                    u64 sum = rlo + mlo;
                    rhi += mhi + (sum < rlo ? 1 : 0);
                    rlo = sum;
                }
            }
            const u64 hhi = mulhi64(rlo * mq.qinv, mq.q);
            const u64 lift = rhi - hhi + mq.q + select_entry<MAXP>((sc_u64)a.t.vtimesqmodp + (long)j * (a.np + 1), a.np, v);
            const u64 x = xq[(long)j * a.N + n];
            u64 z = mont_mul(lift + mq.q2 - x, ((sc_u64)a.t.downparam)[j], mq.q, mq.ninv32);
            long pos = (long)j * a.N + n;
            bool flip = false;
            if (gal) {
                const u64 raw = (u64)n * gal;
                pos = (long)j * a.N + (long)(raw & (u64)(a.N - 1));
                flip = ((raw >> a.logN) & 1) != 0;
            }
            if (acc) {
                const u64* add = ka->addend[item];
                if (add) z = csub(add[(long)j * a.N + n] + z, mq.q);
                else if (flip) z = csub((mq.q - dst[pos]) + z, mq.q);       // the stored value is q - (sum so far), in (0, q]
                else z = csub(dst[pos] + z, mq.q);
            }
            u64 outv = flip ? mq.q - z : z;
            if (post) outv = csub(post[pos] + outv, mq.q);                  // ring.Add(post, rotated): outv in [0, q] (q for a flipped zero), the sum canonical
            dst[pos] = outv;
        }
    }
}
void launch_moddown_batch(const ModDownBatchArgs& a_in, hipStream_t st) {
    // group the items by destination: different destinations are independent, equal ones are applied in order
    ModDownBatchArgs a = a_in;
    int pos = 0; a.ngroups = 0;
    bool used[EXT_MAX_ITEMS] = {};
    for (int i = 0; i < a.nitems; ++i) {
        if (used[i]) continue;
        a.gstart[a.ngroups++] = (unsigned char)pos;
        for (int k = i; k < a.nitems; ++k) if (!used[k] && a.dst[k] == a.dst[i]) { a.order[pos++] = (unsigned char)k; used[k] = true; }
    }
    a.gstart[a.ngroups] = (unsigned char)pos;
    if (a.ngroups < 1) return;
    const int bx = (a.N + PW_THREADS - 1) / PW_THREADS;
    // the P-limb part (two Montgomery products and two float64 divisions per coefficient) is shared by all Q limbs a
    // thread produces: few limb slices per coefficient, the parallelism comes from the destination groups
    const int nj = a.qlist ? a.nqlist : a.level + 1;
    const int cap = 4;
    int by = nj < cap ? nj : cap;
    if (a.ngroups * by < 8) by = nj < 8 ? nj : 8;          // a single external product: spread over the limbs instead
    if (by < 1) return;
    hipLaunchKernelGGL(moddown_batch_kernel, dim3(bx, by, a.ngroups), dim3(PW_THREADS), 0, st, a);
}

// merged form: see ModDownMergedArgs.  NPT = number of special primes (compile time: the members' y live in registers)
typedef const __attribute__((address_space(4))) ModDownMergedArgs* mdm_kargs;
template <int NPT>
__global__ void __launch_bounds__(PW_THREADS) moddown_merged_kernel(ModDownMergedArgs a) {
    mdm_kargs ka = (mdm_kargs)__builtin_amdgcn_kernarg_segment_ptr();
    const int n = blockIdx.x * PW_THREADS + threadIdx.x;
    if (n >= a.N) return;
    const int g = blockIdx.z;
    for (int kk = ka->gstart[g]; kk < ka->gstart[g + 1]; ++kk) {
        const int vi = ka->order[kk];
        const int cnt = (int)ka->cnt[vi];
        const u32 members = ka->mem[vi];
        const u64* xq = a.c1 + (long)(members & 255u) * a.c1_item;
        u64* dst = ka->dst[vi];
        const int acc = (int)ka->accumulate[vi];
        const u64 gal = ka->gal_v[vi] ? (u64)ka->gal_v[vi] : a.galEl;
        const u64* post = (kk == ka->gstart[g + 1] - 1) ? ka->post[vi] : nullptr;
        u64 y[MD_VI_MAX][NPT];
        u32 v[MD_VI_MAX];
#pragma unroll
        for (int k = 0; k < MD_VI_MAX; ++k) {
            v[k] = 0;
#pragma unroll
            for (int i = 0; i < NPT; ++i) y[k][i] = 0;
            if (k < cnt) {
                const u64* xp = a.c1 + (long)((members >> (8 * k)) & 255u) * a.c1_item + a.p_offset;
                double vi_ = 0.0;
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    const Mod mp = load_mod((sc_mod)a.mods_p + i);
                    y[k][i] = mont_mul(xp[(long)i * a.N + n], ((sc_u64)a.t.qoverqiinvqi)[i], mp.q, mp.ninv32);
                    vi_ = vi_ + (double)y[k][i] / (double)mp.q;
                }
                v[k] = (u32)(u64)vi_;
            }
        }
        // The members' y under one special prime meet the same constant in the 128-bit multSum: sum_k y_k[i] * c[j][i] = (sum_k y_k[i]) * c[j][i] as
        // INTEGERS (at most four summands below 2^60: below 2^62; the product below 2^122), so the accumulator -- and with it every bit of the result --
        // is the same with NPT products per limb instead of members x NPT (round 5: the kernel is bound by its instruction count, 1169 per wave)
        u64 ys[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            ys[i] = y[0][i];
#pragma unroll
            for (int k = 1; k < MD_VI_MAX; ++k) ys[i] += y[k][i];          // (y[k][i] = 0 for k >= cnt)
        }
        // the ModDown result of limb j for this coefficient (canonical)
        auto down = [&](int j, const Mod& mq) -> u64 {
            u64 rlo = 0, rhi = 0, vt = 0;
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                u64 mhi, mlo;
                mul64x64(ys[i], ((sc_u64)a.t.qoverqimodp)[(long)j * NPT + i], mhi, mlo);
                const u64 sum = rlo + mlo;
                rhi += mhi + (sum < rlo ? 1 : 0);
                rlo = sum;
            }
#pragma unroll
            for (int k = 0; k < MD_VI_MAX; ++k)
                if (k < cnt) vt = csub(vt + select_entry<NPT>((sc_u64)a.t.vtimesqmodp + (long)j * (NPT + 1), NPT, (u64)v[k]), mq.q);
            const u64 hhi = mulhi64(rlo * mq.qinv, mq.q);
            const u64 lift = rhi - hhi + mq.q + vt;                      // = sum_k (the reference's per-product lift) mod q
            const u64 x = xq[(long)j * a.N + n];                         // lazy, < 2q
            return mont_mul(lift + mq.q2 - x, ((sc_u64)a.t.downparam)[j], mq.q, mq.ninv32);
        };
        if (a.rescale_row) {
            // DivRoundByLastModulus (lattigo, as div_round_last_kernel restates it) of the result, limb by limb: the dropped limb first
            const Mod mL = load_mod((sc_mod)a.mods_q + a.level);
            const u64 qL = mL.q, h = (qL - 1) >> 1;
            const u64 t = csub(down(a.level, mL) + h, qL);
            u64* rd = ka->rdst[vi];
            for (int j = blockIdx.y; j < a.level; j += gridDim.y) {
                const Mod mq = load_mod((sc_mod)a.mods_q + j);
                const u64 hr = ((sc_u64)a.rescale_h)[j];                   // BRedAdd(h, q_j) = h mod q_j (div_round_last_kernel computes it in place)
                const u64 z = down(j, mq);
                rd[(long)j * a.N + n] = mont_mul(t + (mq.q - hr) + mq.q2 - z, mq.q - ((sc_u64)a.rescale_row)[j], mq.q, mq.ninv32);
            }
            continue;
        }
        for (int j = blockIdx.y; j <= a.level; j += gridDim.y) {
            const Mod mq = load_mod((sc_mod)a.mods_q + j);
            u64 z = down(j, mq);
            long pos = (long)j * a.N + n;
            bool flip = false;
            if (gal) {
                const u64 raw = (u64)n * gal;
                pos = (long)j * a.N + (long)(raw & (u64)(a.N - 1));
                flip = ((raw >> a.logN) & 1) != 0;
            }
            if (acc) {
                const u64* add = ka->addend[vi];
                if (add) z = csub(add[(long)j * a.N + n] + z, mq.q);
                else if (flip) z = csub((mq.q - dst[pos]) + z, mq.q);       // the stored value is q - (sum so far), in (0, q]
                else z = csub(dst[pos] + z, mq.q);
            }
            u64 outv = flip ? mq.q - z : z;
            if (post) outv = csub(post[pos] + outv, mq.q);                  // ring.Add(post, rotated): outv in [0, q] (q for a flipped zero), the sum canonical
            dst[pos] = outv;
        }
    }
}
void launch_moddown_merged(const ModDownMergedArgs& a_in, hipStream_t st) {
    ModDownMergedArgs a = a_in;
    int pos = 0; a.ngroups = 0;
    bool used[EXT_MAX_ITEMS] = {};
    for (int i = 0; i < a.nvi; ++i) {
        if (used[i]) continue;
        a.gstart[a.ngroups++] = (unsigned char)pos;
        for (int k = i; k < a.nvi; ++k) if (!used[k] && a.dst[k] == a.dst[i]) { a.order[pos++] = (unsigned char)k; used[k] = true; }
    }
    a.gstart[a.ngroups] = (unsigned char)pos;
    if (a.ngroups < 1) return;
    const int bx = (a.N + PW_THREADS - 1) / PW_THREADS;
    const int nj = a.level + 1;
    // limb slices per coefficient: every slice repeats the P part (y, v of every member: two Montgomery products and two float64 divisions per member and
    // special prime).  Round 5, after the members' y are summed before the multSum products (the per-limb work halved): at N = 2^15 two slices beat four
    // (36.5 against 40.2 us per step for the two ModDown launches of the headline MulRelin, same call; 1: 47.9, 3: 37.4, 7: 45.5); the small rings too, by less
    // (cnn 1.910 against 1.918 ms with four, PN14QP439 equal: four alternating pairs in two calls).
    const int cap = 2;
    int by = nj < cap ? nj : cap;
    if (a.ngroups * by < 8) by = nj < 8 ? nj : 8;
    const dim3 grid(bx, by, a.ngroups), blk(PW_THREADS);
    switch (a.np) {
        case 1: hipLaunchKernelGGL(moddown_merged_kernel<1>, grid, blk, 0, st, a); break;
        case 2: hipLaunchKernelGGL(moddown_merged_kernel<2>, grid, blk, 0, st, a); break;
        case 3: hipLaunchKernelGGL(moddown_merged_kernel<3>, grid, blk, 0, st, a); break;
        case 4: hipLaunchKernelGGL(moddown_merged_kernel<4>, grid, blk, 0, st, a); break;
        default: break;        // (Context::ext_batch merges only for np <= 4)
    }
}

// ------------------------------------------------------------------ tensor (step D)
typedef const __attribute__((address_space(4))) TensorArgs* tensor_kargs;
__global__ void __launch_bounds__(PW_THREADS) tensor_kernel(TensorArgs a) {
    tensor_kargs ka = (tensor_kargs)__builtin_amdgcn_kernarg_segment_ptr();    // per-slot lists: scalar loads
    const int l = a.limbs ? a.limbs[blockIdx.y] : blockIdx.y;
    const Mod md = a.mods[a.map ? a.map[l] : l];
    const u64 q = md.q;
    const u32 ninv = md.ninv32;
    const bool scaled = a.scale != nullptr;
    const u64 sc = scaled ? a.scale[l] : 0;
    const long P = (long)a.L * a.N;          // words per output poly
    const long ib = (long)blockIdx.z * a.in_batch, ob = (long)blockIdx.z * a.out_batch;      // product blockIdx.z of a batched launch
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < a.N; n += gridDim.x * PW_THREADS) {
        const long e = (long)l * a.N + n;
        const u64 a0m = mont_mul(a.a0[ib + e], md.r2, q, ninv);    // MForm(NTT(c0_0))
        const u64 b0 = a.b0[ib + e];
        const u64 b0m = mont_mul(b0, md.r2, q, ninv);         // MForm(NTT(c1_0))
        u64 r0 = a.with_c0 ? mont_mul(a0m, b0, q, ninv) : 0;
        if (scaled) r0 = mont_mul(r0, sc, q, ninv);
        a.out[ob + e] = r0;
        for (int o = 1; o <= a.nout; ++o) {
            u64 r = 0;
            const u64* pa = ka->a[o];
            const u64* pb = ka->b[o];
            if (pa) r = mont_mul(b0m, pa[ib + (long)l * ka->a_ls[o] + n], q, ninv);
            if (pb) r = csub(r + mont_mul(a0m, pb[ib + (long)l * ka->b_ls[o] + n], q, ninv), q);
            if (scaled) r = mont_mul(r, sc, q, ninv);
            a.out[ob + (long)o * P + e] = r;
        }
    }
}
void launch_tensor(const TensorArgs& a, hipStream_t st) {
    int bx = (a.N + PW_THREADS - 1) / PW_THREADS;
    const int by = a.limbs ? a.nlimbs : a.L;
    if (by < 1) return;
    hipLaunchKernelGGL(tensor_kernel, dim3(bx, by, a.nbatch > 1 ? a.nbatch : 1), dim3(PW_THREADS), 0, st, a);
}

// ------------------------------------------------------------------ add
__global__ void __launch_bounds__(PW_THREADS) add_kernel(u64* dst, const u64* x, const u64* y, const Mod* mods, int N) {
    const int l = blockIdx.y;
    const u64 q = mods[l].q;
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < N; n += gridDim.x * PW_THREADS) {
        const long e = (long)l * N + n;
        dst[e] = csub(x[e] + y[e], q);
    }
}
void launch_add(u64* dst, const u64* a, const u64* b, const Mod* mods, int L, int N, hipStream_t st) {
    int bx = (N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(add_kernel, dim3(bx, L), dim3(PW_THREADS), 0, st, dst, a, b, mods, N);
}

__global__ void __launch_bounds__(PW_THREADS) sub_kernel(u64* dst, const u64* x, const u64* y, const Mod* mods, int N) {
    const int l = blockIdx.y;
    const u64 q = mods[l].q;
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < N; n += gridDim.x * PW_THREADS) {
        const long e = (long)l * N + n;
        dst[e] = csub(x[e] + q - y[e], q);
    }
}
void launch_sub(u64* dst, const u64* a, const u64* b, const Mod* mods, int L, int N, hipStream_t st) {
    int bx = (N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(sub_kernel, dim3(bx, L), dim3(PW_THREADS), 0, st, dst, a, b, mods, N);
}
typedef const __attribute__((address_space(4))) CtBinArgs* ctbin_kargs;
__global__ void __launch_bounds__(PW_THREADS) ct_binary_kernel(CtBinArgs a) {
    ctbin_kargs ka = (ctbin_kargs)__builtin_amdgcn_kernarg_segment_ptr();      // per-component lists: scalar loads, no scratch copy
    const int l = blockIdx.y, c = blockIdx.z;
    const u64 q = a.mods[l].q;
    const u64* x = ka->a[c]; const u64* y = ka->b[c]; u64* dst = ka->dst[c];
    const int mode = ka->mode[c];
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < a.N; n += gridDim.x * PW_THREADS) {
        const long e = (long)l * a.N + n;
        u64 v;
        if (mode == 0) v = csub(x[e] + y[e], q);
        else if (mode == 1) v = csub(x[e] + q - y[e], q);
        else if (mode == 2) v = x[e];
        else if (mode == 3) v = y[e];
        else v = q - y[e];
        dst[e] = v;
    }
}
void launch_ct_binary(const CtBinArgs& a, hipStream_t st) {
    int bx = (a.N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(ct_binary_kernel, dim3(bx, a.L, a.ncomp), dim3(PW_THREADS), 0, st, a);
}
typedef const __attribute__((address_space(4))) CtSumArgs* ctsum_kargs;
__global__ void __launch_bounds__(PW_THREADS) ct_sum_kernel(CtSumArgs a) {
    ctsum_kargs ka = (ctsum_kargs)__builtin_amdgcn_kernarg_segment_ptr();
    const int l = blockIdx.y;
    const u64 q = a.mods[l].q;
    const long base = ((long)blockIdx.z * a.L + l) * a.N;
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < a.N; n += gridDim.x * PW_THREADS) {
        u64 v = ka->in[0][base + n];
        for (int k = 1; k < a.n; ++k) v = csub(v + ka->in[k][base + n], q);
        a.dst[base + n] = v;
    }
}
void launch_ct_sum(const CtSumArgs& a, hipStream_t st) {
    if (a.n < 1 || a.npolys < 1 || a.L < 1) return;
    const int bx = (a.N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(ct_sum_kernel, dim3(bx, a.L, a.npolys), dim3(PW_THREADS), 0, st, a);
}
// ring.Neg writes q - a, i.e. q for a = 0 (lattigo ring_operations.go Neg), kept literally
__global__ void __launch_bounds__(PW_THREADS) neg_kernel(u64* dst, const u64* x, const Mod* mods, int N) {
    const int l = blockIdx.y;
    const u64 q = mods[l].q;
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < N; n += gridDim.x * PW_THREADS) {
        const long e = (long)l * N + n;
        dst[e] = q - x[e];
    }
}
void launch_neg(u64* dst, const u64* a, const Mod* mods, int L, int N, hipStream_t st) {
    int bx = (N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(neg_kernel, dim3(bx, L), dim3(PW_THREADS), 0, st, dst, a, mods, N);
}

__global__ void __launch_bounds__(PW_THREADS) mul_const_kernel(u64* dst, const u64* src, const Mod* mods, const int* map, const u64* consts,
                                                                int N, long poly_stride) {
    const int l = blockIdx.y;
    const Mod md = mods[map ? map[l] : l];
    const u64 c = consts[l];
    const long base = (long)blockIdx.z * poly_stride + (long)l * N;
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < N; n += gridDim.x * PW_THREADS)
        dst[base + n] = mont_mul(src[base + n], c, md.q, md.ninv32);
}
void launch_mul_const(u64* dst, const u64* src, const Mod* mods, const int* map, const u64* consts, int L, int N, int npolys,
                      long poly_stride, hipStream_t st) {
    int bx = (N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(mul_const_kernel, dim3(bx, L, npolys), dim3(PW_THREADS), 0, st, dst, src, mods, map, consts, N, poly_stride);
}

typedef const __attribute__((address_space(4))) MulConstArgs* mulconst_kargs;
__global__ void __launch_bounds__(PW_THREADS) mul_const_halves_kernel(MulConstArgs a) {
    mulconst_kargs ka = (mulconst_kargs)__builtin_amdgcn_kernarg_segment_ptr();
    const int l = blockIdx.y;
    const Mod md = a.mods[l];
    const u64 c0 = ka->c[0][l], c1 = ka->c[1][l];
    const u64* s = a.src + (long)blockIdx.z * a.src_poly + (long)l * a.N;
    u64* d = a.dst + (long)blockIdx.z * a.dst_poly + (long)l * a.N;
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < a.N; n += gridDim.x * PW_THREADS)
        d[n] = mont_mul(s[n], n < (a.N >> 1) ? c0 : c1, md.q, md.ninv32);
}
void launch_mul_const_halves(const MulConstArgs& a, hipStream_t st) {
    int bx = (a.N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(mul_const_halves_kernel, dim3(bx, a.L, a.npolys), dim3(PW_THREADS), 0, st, a);
}
__global__ void __launch_bounds__(PW_THREADS) mul_by_poly_kernel(u64* dst, const u64* a, const u64* b, const Mod* mods, int L, int N) {
    const int l = blockIdx.y;
    const Mod md = mods[l];
    const long base = ((long)blockIdx.z * L + l) * N;
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < N; n += gridDim.x * PW_THREADS) {
        const u64 bm = mont_mul(b[(long)l * N + n], md.r2, md.q, md.ninv32);       // MFormLvl(pt)
        dst[base + n] = mont_mul(a[base + n], bm, md.q, md.ninv32);
    }
}
void launch_mul_by_poly(u64* dst, const u64* a, const u64* b, const Mod* mods, int L, int N, int npolys, hipStream_t st) {
    int bx = (N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(mul_by_poly_kernel, dim3(bx, L, npolys), dim3(PW_THREADS), 0, st, dst, a, b, mods, L, N);
}

// ------------------------------------------------------------------ generic basis conversion (mkbfv)
__global__ void __launch_bounds__(PW_THREADS) basis_conv_kernel(BasisConvArgs a) {
    const int n = blockIdx.x * PW_THREADS + threadIdx.x;
    if (n >= a.N) return;
    const int pi = blockIdx.z;
    const u64* src = a.src + (long)pi * a.src_poly;
    u64 y[BC_MAXS];
    double vi = 0.0;
    const bool copy = a.copy_dst != nullptr && blockIdx.y == 0;
#pragma unroll
    for (int i = 0; i < BC_MAXS; ++i) {
        if (i < a.ns) {
            const Mod ms = a.mods_s[i];
            u64 x = src[(long)i * a.N + n];
            if (copy) a.copy_dst[(long)pi * a.copy_poly + (long)i * a.N + n] = x;
            if (a.prescale) x = mont_mul(x, a.prescale[i], ms.q, ms.ninv32);
            y[i] = mont_mul(x, a.t.qoverqiinvqi[i], ms.q, ms.ninv32);
            vi = vi + (double)y[i] / (double)ms.q;
        }
    }
    const u64 v = (u64)vi;
    for (int j = blockIdx.y; j < a.nt; j += gridDim.y) {
        const Mod mt = a.mods_t[j];
        // multSum (basis_extension.go:587-646): the exact 128-bit sum of the ns products y_i * t_i, formed column by column -- y_i = y1 2^32 + y0,
        // t_i = t1 2^32 + t0 (wave-uniform: SGPR operands): c0 = sum y0 t0, c1 = sum (y0 t1 + y1 t0), c2 = sum y1 t1 as 64-bit multiply-adds
        // whose carry-outs are counted (k0, k1; c2 < ns 2^56 cannot carry), 7 instructions per term where the 128-bit product + 128-bit
        // addition took 12.  Same integer, hence the same (rlo, rhi) as the reference's sequential accumulation.
        u64 c0 = 0, c1 = 0, c2 = 0;
        u32 k0 = 0, k1 = 0;
#pragma unroll
        for (int i = 0; i < BC_MAXS; ++i) {
            if (i < a.ns) {
                const u64 t = a.t.qoverqimodp[(long)j * a.ns + i];
                const u32 y0 = lo32(y[i]), y1 = hi32(y[i]);
                u32 t0 = lo32(t), t1 = hi32(t);
                asm("v_mad_u64_u32 %0, vcc, %5, %7, %0\n\t"
                    "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
                    "v_mad_u64_u32 %1, vcc, %5, %8, %1\n\t"
                    "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
                    "v_mad_u64_u32 %1, vcc, %6, %7, %1\n\t"
                    "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
                    "v_mad_u64_u32 %2, vcc, %6, %8, %2"
                    : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(k0), "+v"(k1)
                    : "v"(y0), "v"(y1), "s"(t0), "s"(t1)
                    : "vcc");
            }
        }
        // total = c0 + k0 2^64 + (c1 + k1 2^64) 2^32 + c2 2^64
        const u64 rlo = c0 + (c1 << 32);
        const u64 rhi = (u64)k0 + (c1 >> 32) + ((u64)k1 << 32) + c2 + (rlo < c0 ? 1 : 0);
        const u64 hhi = mulhi64(rlo * mt.qinv, mt.q);
        u64 z = rhi - hhi + mt.q + a.t.vtimesqmodp[(long)j * (a.ns + 1) + v];
        if (a.downparam) {
            const u64 x = a.xsub ? a.xsub[(long)pi * a.xsub_poly + (long)j * a.N + n] : 0;
            z = mont_mul(z + mt.q2 - x, a.downparam[j], mt.q, mt.ninv32);
        }
        a.dst[(long)pi * a.dst_poly + (long)j * a.N + n] = z;
    }
}
void launch_basis_conv(const BasisConvArgs& a, hipStream_t st) {
    const int bx = (a.N + PW_THREADS - 1) / PW_THREADS;
    const int per = 7;                       // target limbs per thread: the y[] set-up (ns products + ns float64 divisions) is amortised over them
                                             // (measured at PN15, nt = 14: 7 -> 44.6 us per launch, 4 -> 54.8, 14 -> 54.2, 2 -> 68.7)
    int by = (a.nt + per - 1) / per;
    if (by < 1) by = 1;
    hipLaunchKernelGGL(basis_conv_kernel, dim3(bx, by, a.npolys), dim3(PW_THREADS), 0, st, a);
}

// ------------------------------------------------------------------ gadget digit spread, alpha >= 2
typedef const __attribute__((address_space(4))) DecompSpreadArgs* dspread_kargs;
// value of coefficient n of digit d under target modulus m (lazy, < 4q): the literal reconstructRNS / multSum sequence
struct SpreadCoeff {
    u64 y[DEC_MAXA];
    u64 v;
};
__device__ __forceinline__ void spread_prepare(const DecompSpreadArgs& a, const u64* src, const u64* ta, int start, int nd, int n, SpreadCoeff& c) {
    double vi = 0.0;
#pragma unroll
    for (int i = 0; i < DEC_MAXA; ++i) {
        if (i < nd) {
            const Mod ms = a.mods[start + i];
            c.y[i] = mont_mul(src[(long)i * a.N + n], ta[i], ms.q, ms.ninv32);
            vi = vi + (double)c.y[i] / (double)ms.q;
        }
    }
    c.v = (u64)vi;
}
// The per-slot constants are wave-uniform: they are read with SCALAR loads (constant address space) into SlotConsts once per slot and shared by
// the four coefficients of a thread -- round 2 fetched them with per-lane vector loads (every lane the same address) inside spread_value, one
// dependent global round trip per coefficient and slot at four waves per SIMD, which is what the kernel was bound by.  vtimesqmodp[v] (v <= nd <= 4,
// per coefficient) is selected from the nd + 1 scalar entries instead of gathered.
struct SlotConsts { u64 tb[DEC_MAXA]; u64 tc[DEC_MAXA + 1]; };
__device__ __forceinline__ void slot_consts(SlotConsts& k, const u64* tb, const u64* tc, int m, int nd) {
    sc_u64 b = (sc_u64)tb + (long)m * DEC_MAXA, c = (sc_u64)tc + (long)m * (DEC_MAXA + 1);
#pragma unroll
    for (int i = 0; i < DEC_MAXA; ++i) k.tb[i] = i < nd ? b[i] : 0;
#pragma unroll
    for (int i = 0; i <= DEC_MAXA; ++i) k.tc[i] = i <= nd ? c[i] : 0;
}
__device__ __forceinline__ u64 spread_value(const SpreadCoeff& c, const SlotConsts& k, int nd, const Mod& mt) {
    u64 rlo = 0, rhi = 0;
#pragma unroll
    for (int i = 0; i < DEC_MAXA; ++i) {
        if (i < nd) {
            u64 mhi, mlo;
            mul64x64(c.y[i], k.tb[i], mhi, mlo);
            u64 sum = rlo + mlo;
            rhi += mhi + (sum < rlo ? 1 : 0);
            rlo = sum;
        }
    }
    const u64 hhi = mulhi64(rlo * mt.qinv, mt.q);
    u64 vt = k.tc[0];
#pragma unroll
    for (int i = 1; i <= DEC_MAXA; ++i) vt = (i <= nd && c.v == (u64)i) ? k.tc[i] : vt;
    return rhi - hhi + mt.q + vt;
}
// first Cooley-Tukey stage on the pair (lo, hi) = coefficients (n, n + N/2), both < 4q: same arithmetic as ntt_split_fwd_kernel
__device__ __forceinline__ void spread_first_stage(u64& lo, u64& hi, u64 w, const Mod& mt) {
    const u64 U = csub(lo, mt.q2);
    const u64 Tm = mont_mul_sdu(hi, w, mt.qs, mt.q, mt.ninv32);
    lo = U + Tm;
    hi = U + (mt.q2 - Tm);
}
// Two adjacent coefficients per thread (16-byte lanes: the kernel writes beta * m limbs per source digit and is bound by its stores).
__device__ __forceinline__ void st2(u64* p, u64 x0, u64 x1) {
    u64x2 r; r.x = x0; r.y = x1;
    __builtin_nontemporal_store(r, (u64x2*)p);          // written once, gigabytes per launch, read by the NTT that follows: past the caches
}
// the first FS Cooley-Tukey stages of the N-point transform on the 2^FS values v[k] = coefficient n + k N / 2^FS (all < 4q, and < 4q again
// afterwards): FS = 1: the pair (0, 1) with psi[1]  (FS = 2 is decomp_spread4_kernel below)
template <int FS>
__device__ __forceinline__ void spread_stages(u64* v, const u64* w, const Mod& mt) {
    if constexpr (FS == 1) spread_first_stage(v[0], v[1], w[1], mt);
}
// FS = DecompSpreadArgs::first_stage: a thread produces the coefficients n, n + 1 at each of the 2^FS points n + k N / 2^FS and applies the first
// FS stages of the forward NTT to them before they are stored
template <int FS>
__global__ void __launch_bounds__(PW_THREADS) decomp_spread_kernel(DecompSpreadArgs a) {
    constexpr int NP = 1 << FS;
    dspread_kargs ka = (dspread_kargs)__builtin_amdgcn_kernarg_segment_ptr();
    const int n = 2 * (blockIdx.x * PW_THREADS + threadIdx.x);
    const int S = a.N >> FS;
    if (n >= S) return;
    const int d = blockIdx.y, item = blockIdx.z;
    const int start = d * a.alpha, nd = ka->nd[d];
    const u64* src = ka->src[item] + (long)start * a.N;
    u64* dst = ka->dst[item] + (long)d * a.mtot * a.N;
    if (nd == 1) {
        u64x2 x[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) x[k] = *(const u64x2*)(src + n + k * S);
        const u64 qs = a.mods[start].q;
        for (int s = 0; s < a.nslots; ++s) {
            const int m = a.map[s];
            const Mod mt = a.mods[m];
            const bool red = qs > 4 * mt.q;
            u64 v0[NP], v1[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                v0[k] = red ? mont_mul_lazy(x[k].x, mt.r1, mt.q, mt.ninv32) : x[k].x;
                v1[k] = red ? mont_mul_lazy(x[k].y, mt.r1, mt.q, mt.ninv32) : x[k].y;
            }
            if constexpr (FS > 0) {
                u64 w[4];
#pragma unroll
                for (int i = 1; i < NP; ++i) w[i] = a.psi[(long)m * a.N + i];
                spread_stages<FS>(v0, w, mt);
                spread_stages<FS>(v1, w, mt);
            }
#pragma unroll
            for (int k = NP - 1; k >= 0; --k) st2(dst + (long)m * a.N + n + k * S, v0[k], v1[k]);
        }
        return;
    }
    const long tsel = (long)d * (a.alpha - 1) + (nd - 2);
    const u64* ta = a.ta + tsel * DEC_MAXA;
    const u64* tb = a.tb + tsel * a.mtot * DEC_MAXA;
    const u64* tc = a.tc + tsel * a.mtot * (DEC_MAXA + 1);
    SpreadCoeff c0[NP], c1[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) { spread_prepare(a, src, ta, start, nd, n + k * S, c0[k]); spread_prepare(a, src, ta, start, nd, n + k * S + 1, c1[k]); }
    for (int s = 0; s < a.nslots; ++s) {
        const int m = ((sc_int)a.map)[s];
        const Mod mt = load_mod((sc_mod)a.mods + m);
        SlotConsts kc;
        slot_consts(kc, tb, tc, m, nd);
        u64 v0[NP], v1[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) { v0[k] = spread_value(c0[k], kc, nd, mt); v1[k] = spread_value(c1[k], kc, nd, mt); }
        if constexpr (FS > 0) {
            u64 w[4];
#pragma unroll
            for (int i = 1; i < NP; ++i) w[i] = ((sc_u64)a.psi)[(long)m * a.N + i];
            spread_stages<FS>(v0, w, mt);
            spread_stages<FS>(v1, w, mt);
        }
#pragma unroll
        for (int k = NP - 1; k >= 0; --k) st2(dst + (long)m * a.N + n + k * S, v0[k], v1[k]);
    }
}
// ---- the radix-4 form (first_stage = 2, round 3): digits of one or two limbs, every modulus below 2^57
// The digit value and the two butterfly stages run on one-round products of radix 2^30 with every operand NON-NEGATIVE (the all-unsigned
// sibling of ntt16_kernels.hip mm30u): a data word y = hi 2^32 + lo meets a constant t as the pre-reduced pair u = t 2^30 mod p, v = t 2^62 mod p
// (both in [0, p), radix-2^30 digits), and
//   C = sum lo u0 + hi v0 ; m = lo30(C * -p^-1) ; T = ((C + m p0) >> 30) + sum lo u1 + hi v1 + m p1 = (sum (lo u + hi v) + m p) / 2^30 = sum y t  (mod p)
// with ONE reduction round for the whole sum: 13 multiplier-class instructions for a two-limb digit where the 128-bit multSum + 64-bit
// Montgomery fold of spread_value takes 21, 9 for a butterfly product.  Only the residue class of a spread digit reaches the results (the
// NTT that follows reduces), so the representative is free: 0 <= T < (2^32 nd + 2^25 nd + 2^30) p / 2^30 < 9.1 p for nd = 2, a butterfly
// product of V < 2^62 is < 6p and its outputs are U + T and U + (6p - T): below 22.2 p after the two stages, < 2^62 for p < 2^57.
struct Pair30 { u32 u0, u1, v0, v1; };
__device__ __forceinline__ Pair30 load_pair30(sc_u64 p) {
    const u64 u = p[0], v = p[1];
    Pair30 r{lo32(u), hi32(u), lo32(v), hi32(v)};
    asm("" : "+s"(r.u0), "+s"(r.u1), "+s"(r.v0), "+s"(r.v1));          // opaque 32-bit scalars (else they are multiplied as halves of a 64-bit constant)
    return r;
}
struct Mod30 { u32 p0, p1, ninv; u64 six; };
__device__ __forceinline__ u64 fold30(u64 c0, u64 c1, const Mod30& k) {
    const u32 m = (lo32(c0) * k.ninv) & 0x3fffffffu;
    return ((c0 + (u64)m * k.p0) >> 30) + c1 + (u64)m * k.p1;
}
__device__ __forceinline__ u64 prod30(u64 y, const Pair30& t, const Mod30& k) {
    const u32 lo = lo32(y), hi = hi32(y);
    return fold30((u64)lo * t.u0 + (u64)hi * t.v0, (u64)lo * t.u1 + (u64)hi * t.v1, k);
}
__device__ __forceinline__ u64 sum30(u64 y0, u64 y1, const Pair30& t0, const Pair30& t1, const Mod30& k) {
    const u32 l0 = lo32(y0), h0 = hi32(y0), l1 = lo32(y1), h1 = hi32(y1);
    return fold30((u64)l0 * t0.u0 + (u64)h0 * t0.v0 + (u64)l1 * t1.u0 + (u64)h1 * t1.v0,
                  (u64)l0 * t0.u1 + (u64)h0 * t0.v1 + (u64)l1 * t1.u1 + (u64)h1 * t1.v1, k);
}
__device__ __forceinline__ void bfly30(u64& U, u64& V, const Pair30& w, const Mod30& k) {
    const u64 T = prod30(V, w, k);
    V = U + (k.six - T);
    U = U + T;
}
__device__ __forceinline__ void stages30(u64* v, const Pair30* w, const Mod30& k) {
    bfly30(v[0], v[2], w[0], k);
    bfly30(v[1], v[3], w[0], k);
    bfly30(v[0], v[1], w[1], k);
    bfly30(v[2], v[3], w[2], k);
}
__global__ void __launch_bounds__(PW_THREADS) decomp_spread4_kernel(DecompSpreadArgs a) {
    dspread_kargs ka = (dspread_kargs)__builtin_amdgcn_kernarg_segment_ptr();
    const int n = 2 * (blockIdx.x * PW_THREADS + threadIdx.x);
    const int S = a.N >> 2;
    if (n >= S) return;
    const int d = blockIdx.y, item = blockIdx.z;
    const int start = d * a.alpha, nd = ka->nd[d];
    const u64* src = ka->src[item] + (long)start * a.N;
    u64* dst = ka->dst[item] + (long)d * a.mtot * a.N;
    u64 y0[4][2], y1[4][2];                  // [point][adjacent coefficient]: the y_i of the digit's two limbs (nd = 1: the coefficient itself in y0)
    u32 cv[4][2];
    if (nd == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { const u64x2 x = *(const u64x2*)(src + n + k * S); y0[k][0] = x.x; y0[k][1] = x.y; y1[k][0] = y1[k][1] = 0; cv[k][0] = cv[k][1] = 0; }
    } else {
        const u64* ta = a.ta + (long)d * (a.alpha - 1) * DEC_MAXA;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                SpreadCoeff c;
                spread_prepare(a, src, ta, start, 2, n + k * S + e, c);
                y0[k][e] = c.y[0]; y1[k][e] = c.y[1]; cv[k][e] = (u32)c.v;
            }
    }
    sc_u64 t30 = (sc_u64)a.tb30 + (long)d * a.mtot * 4;
    sc_u64 tcs = (sc_u64)a.tc + (long)d * (a.alpha - 1) * a.mtot * (DEC_MAXA + 1);
    const u64 qs = ((sc_mod)a.mods)[start].q;
    for (int s = 0; s < a.nslots; ++s) {
        const int m = ((sc_int)a.map)[s];
        sc_mod mp = (sc_mod)a.mods + m;
        const u64 q = mp->q;
        Mod30 k30{lo32(q) & 0x3fffffffu, (u32)(q >> 30), mp->ninv32, 6 * q};
        asm("" : "+s"(k30.p0), "+s"(k30.p1), "+s"(k30.ninv));
        Pair30 w[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i] = load_pair30((sc_u64)a.tw30 + ((long)m * 4 + 1 + i) * 2);
        u64 v0[4], v1[4];
        if (nd == 1) {
            // copy path (:443-451): the limb itself under every modulus; one product by 2^30-form 1 (= a reduction below 6p) when its bound exceeds 4p
            const bool red = qs > 4 * q;
            const Pair30 one = load_pair30((sc_u64)a.tw30 + (long)m * 4 * 2);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v0[k] = red ? prod30(y0[k][0], one, k30) : y0[k][0]; v1[k] = red ? prod30(y0[k][1], one, k30) : y0[k][1]; }
        } else {
            const Pair30 t0 = load_pair30(t30 + (long)m * 4), t1 = load_pair30(t30 + (long)m * 4 + 2);
            const u64 c1 = tcs[(long)m * (DEC_MAXA + 1) + 1], c2 = tcs[(long)m * (DEC_MAXA + 1) + 2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v0[k] = sum30(y0[k][0], y1[k][0], t0, t1, k30) + (cv[k][0] == 0 ? 0 : cv[k][0] == 1 ? c1 : c2);
                v1[k] = sum30(y0[k][1], y1[k][1], t0, t1, k30) + (cv[k][1] == 0 ? 0 : cv[k][1] == 1 ? c1 : c2);
            }
        }
        stages30(v0, w, k30);
        stages30(v1, w, k30);
#pragma unroll
        for (int k = 3; k >= 0; --k) st2(dst + (long)m * a.N + n + k * S, v0[k], v1[k]);
    }
}
void launch_decomp_spread(const DecompSpreadArgs& a, hipStream_t st) {
    const int cnt = (a.N >> a.first_stage) / 2;          // two adjacent coefficients per thread at each of its points
    const int bx = (cnt + PW_THREADS - 1) / PW_THREADS;
    const dim3 grid(bx, a.ndigits, a.nitems);
    if (a.first_stage == 2) {
        if (!a.tb30 || !a.tw30 || a.alpha != 2) throw std::runtime_error("mkhe: internal: radix-4 digit spread without its tables");
        hipLaunchKernelGGL(decomp_spread4_kernel, grid, dim3(PW_THREADS), 0, st, a);
    }
    else if (a.first_stage == 1) hipLaunchKernelGGL(decomp_spread_kernel<1>, grid, dim3(PW_THREADS), 0, st, a);
    else hipLaunchKernelGGL(decomp_spread_kernel<0>, grid, dim3(PW_THREADS), 0, st, a);
}

// ------------------------------------------------------------------ automorphism
__global__ void __launch_bounds__(PW_THREADS) automorphism_kernel(u64* dst, const u64* src, const Mod* mods, int L, int logN, u64 galEl) {
    const int l = blockIdx.y, pidx = blockIdx.z;
    const int N = 1 << logN;
    const u64 q = mods[l].q;
    const long base = ((long)pidx * L + l) * N;
    for (int i = blockIdx.x * PW_THREADS + threadIdx.x; i < N; i += gridDim.x * PW_THREADS) {
        const u64 raw = (u64)i * galEl;
        const int idx = (int)(raw & (u64)(N - 1));
        const u64 v = src[base + i];
        dst[base + idx] = ((raw >> logN) & 1) ? q - v : v;     // 0 with a sign flip is written as q, like the reference
    }
}
void launch_automorphism(u64* dst, const u64* src, const Mod* mods, int L, int logN, u64 galEl, int npolys, hipStream_t st) {
    const int N = 1 << logN;
    int bx = (N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(automorphism_kernel, dim3(bx, L, npolys), dim3(PW_THREADS), 0, st, dst, src, mods, L, logN, galEl);
}

// ------------------------------------------------------------------ rescale step
__global__ void __launch_bounds__(PW_THREADS) div_round_last_kernel(u64* dst, const u64* src, const Mod* mods, const u64* rescale_row,
                                                                     int level, int N, long src_poly, long dst_poly) {
    const int i = blockIdx.y, pidx = blockIdx.z;
    const Mod mi = mods[i];
    const u64 qL = mods[level].q, h = (qL - 1) >> 1;
    // BRedAdd(h, q_i): h < 2^60, q_i > 2^20 -> plain remainder (canonical either way).  The 64-bit `%` is a long software loop that
    // every thread would run for one coefficient's worth of work: the quotient (< 2^40) is estimated in float64 (off by at most one)
    // and the remainder fixed up -- same value.
    u64 hr;
    {
        const u64 k = (u64)((double)h / (double)mi.q);
        hr = h - k * mi.q;
        if ((i64)hr < 0) hr += mi.q;
        if (hr >= mi.q) hr -= mi.q;
    }
    const u64 hneg = mi.q - hr;
    const u64 rp = mi.q - rescale_row[i];
    const u64* s = src + (long)pidx * src_poly;
    u64* d = dst + (long)pidx * dst_poly;
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < N; n += gridDim.x * PW_THREADS) {
        const u64 t = csub(s[(long)level * N + n] + h, qL);
        d[(long)i * N + n] = mont_mul(t + hneg + mi.q2 - s[(long)i * N + n], rp, mi.q, mi.ninv32);
    }
}
void launch_div_round_last(u64* dst, const u64* src, const Mod* mods, const u64* rescale_row, int level, int N, int npolys,
                           long src_poly, long dst_poly, hipStream_t st) {
    if (level < 1) return;
    int bx = (N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(div_round_last_kernel, dim3(bx, level, npolys), dim3(PW_THREADS), 0, st, dst, src, mods, rescale_row, level, N, src_poly, dst_poly);
}

typedef const __attribute__((address_space(4))) DivRoundListArgs* drl_kargs;
__global__ void __launch_bounds__(PW_THREADS) div_round_last_list_kernel(DivRoundListArgs a) {
    drl_kargs ka = (drl_kargs)__builtin_amdgcn_kernarg_segment_ptr();
    const int i = blockIdx.y, pidx = blockIdx.z, g = pidx / a.per_group, p = pidx - g * a.per_group;
    const Mod mi = a.mods[i];
    const u64 qL = a.mods[a.level].q, h = (qL - 1) >> 1;
    u64 hr;
    {
        const u64 k = (u64)((double)h / (double)mi.q);             // BRedAdd(h, q_i) as in div_round_last_kernel
        hr = h - k * mi.q;
        if ((i64)hr < 0) hr += mi.q;
        if (hr >= mi.q) hr -= mi.q;
    }
    const u64 hneg = mi.q - hr;
    const u64 rp = mi.q - a.rescale_row[i];
    const u64* s = a.src + (long)pidx * a.src_poly;
    u64* d = ka->dst[g] + (long)p * a.dst_poly;
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < a.N; n += gridDim.x * PW_THREADS) {
        const u64 t = csub(s[(long)a.level * a.N + n] + h, qL);
        d[(long)i * a.N + n] = mont_mul(t + hneg + mi.q2 - s[(long)i * a.N + n], rp, mi.q, mi.ninv32);
    }
}
void launch_div_round_last_list(const DivRoundListArgs& a, hipStream_t st) {
    if (a.level < 1 || a.ngroups < 1) return;
    const int bx = (a.N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(div_round_last_list_kernel, dim3(bx, a.level, a.ngroups * a.per_group), dim3(PW_THREADS), 0, st, a);
}

// ------------------------------------------------------------------ fold (reduction epilogue)
__global__ void __launch_bounds__(PW_THREADS) fold_kernel(FoldArgs a) {
    const int m = a.map[blockIdx.y];
    const Mod md = a.mods[m];
    u64* p = a.buf + (long)blockIdx.z * a.poly_stride + (long)m * a.N;
    const u64 w = a.mform ? md.r2 : md.r1;       // x*R2/R = x*R (MForm) ; x*R1/R = x
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < a.N; n += gridDim.x * PW_THREADS)
        p[n] = mont_mul(p[n], w, md.q, md.ninv32);
}
void launch_fold(const FoldArgs& a, hipStream_t st) {
    int bx = (a.N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(fold_kernel, dim3(bx, a.nslots, a.npolys), dim3(PW_THREADS), 0, st, a);
}

// The same epilogue for a limb RANGE whose summands arrive as separate pieces (the reduce-scatter half of a mesh exchange, mkhe_kklss_amd/dist.py): limb l of
// the range = sum over the pieces, folded [and MForm'ed], written to dst.  Limb l of a switching-key buffer belongs to digit l / mtot and modulus l % mtot;
// limbs outside the active digits / moduli of the level are left alone.
__global__ void __launch_bounds__(PW_THREADS) fold_pieces_kernel(FoldPiecesArgs a) {
    const long l = a.first_limb + blockIdx.y;
    const int digit = (int)(l / a.mtot), m = (int)(l % a.mtot);
    if (digit >= a.ndigits || !((a.active >> m) & 1)) return;
    const Mod md = a.mods[m];
    const u64 w = a.mform ? md.r2 : md.r1;
    const u64* src = a.pieces + (long)blockIdx.y * a.N;
    u64* dst = a.dst + (long)blockIdx.y * a.N;
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < a.N; n += gridDim.x * PW_THREADS) {
        u64 v = 0;
        for (int p = 0; p < a.npieces; ++p) v += src[(long)p * a.piece_stride + n];
        dst[n] = mont_mul(v, w, md.q, md.ninv32);
    }
}
void launch_fold_pieces(const FoldPiecesArgs& a, hipStream_t st) {
    if (a.nlimbs < 1) return;
    int bx = (a.N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(fold_pieces_kernel, dim3(bx, a.nlimbs), dim3(PW_THREADS), 0, st, a);
}

// ------------------------------------------------------------------ mform
__global__ void __launch_bounds__(PW_THREADS) mform_kernel(u64* dst, const u64* src, const Mod* mods, const int* map, int N) {
    const int m = map[blockIdx.y];
    const Mod md = mods[m];
    for (int n = blockIdx.x * PW_THREADS + threadIdx.x; n < N; n += gridDim.x * PW_THREADS) {
        const long e = (long)m * N + n;
        dst[e] = mont_mul(src[e], md.r2, md.q, md.ninv32);
    }
}
void launch_mform(u64* dst, const u64* src, const Mod* mods, const int* map, int nslots, int N, hipStream_t st) {
    int bx = (N + PW_THREADS - 1) / PW_THREADS;
    hipLaunchKernelGGL(mform_kernel, dim3(bx, nslots), dim3(PW_THREADS), 0, st, dst, src, mods, map, N);
}

}  // namespace mkhe
