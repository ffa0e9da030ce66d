// ntt_kernels.h -- batched negacyclic NTT / INTT over RNS limbs for gfx950.
//
// One workgroup transforms one whole limb in ONE pass over HBM (read N words, write N
// words): N/32 threads hold 32 coefficients each in VGPRs (N = 2^15: 1024 threads,
// 16 wavefronts, the 256 KiB limb lives in the register file); the 15 butterfly stages
// run as three register-resident radix-32 phases and the data is re-distributed between
// phases through LDS as two 32-bit planes (N*4 B = 128 KiB <= 160 KiB LDS/CU).
//
//   forward (Cooley-Tukey, natural in -> bit-reversed out, lattigo ring.NTT semantics):
//     load  [layout A: regs = index bits n-1..n-5, lanes = low bits]   coalesced 8 B/lane
//     phase 1: stages on bits n-1..n-5      twiddles uniform -> SGPRs
//     X(A->B)  cross-wave exchange
//     phase 2: stages on bits 9..5          [layout B: regs = bits 9..5, lanes = bits 4..0]
//     X(B->C)  half-wave local
//     phase 3: stages on bits 4..0          [layout C: regs = bits 4..0]
//     X(C->B)  half-wave local, store in layout B (256-B contiguous per half wave)
//   inverse (Gentleman-Sande, lattigo ring.InvNTT / InvNTTLazy) is the mirror image.
//
// LDS image of a plane: forward kernels p + (p >> 5) (padded: every layout is base + r*stride with compile-time
// strides, the DS instructions carry immediate offsets), inverse kernel p ^ ((p >> 5) & 31) (swizzled: fewer live
// registers there); both conflict-free for every layout above under the 32-bank rule of ds_read_b32 / ds_write_b32.
//
// Arithmetic: signed-digit Montgomery products (modarith.h mont_mul_sd; twiddles and constants pre-split); moduli
// with 34q < 2^63 run the forward butterflies on signed lazy values without any reduction (MODE 1), larger ones as
// Harvey butterflies reduced every stage (MODE 0).  Persistent workgroups walk a slot-major job list.  Launches with
// fewer limbs than CUs and every N = 2^16 launch run split (NttBatch::split): a streaming cross-half radix-2 pass +
// two half-size register-resident sub-transforms per limb.
//
// Replaces: lattigo ring.NTTLvl / InvNTTLvl / InvNTTLazyLvl as called from
// mkrlwe/keyswitch.go:29-30,58,88,114-115,206 and keyswitch_hoisted.go:36-37,120-143.
#pragma once
#include "modarith.h"

// Timing experiments that produce WRONG RESULTS on purpose -- the MKHE_*_X_* switches of the NTT kernels (butterflies, LDS exchanges, result
// stores, source or twiddle loads taken out: tools/ntt16_ablation.sh, tools/build_variant.sh) -- exist only in builds that say so with
// -DMKHE_ABLATION; a product build that picks one of them up by accident does not compile.
#if !defined(MKHE_ABLATION) && (defined(MKHE_H16_X_NOBFLY) || defined(MKHE_H16_X_NOXCHG) || defined(MKHE_H16_X_NOSTORE) || defined(MKHE_H16_X_NOPARK) || \
                                defined(MKHE_H16_X_NOSRC) || defined(MKHE_H16_X_NOTWLOAD) || defined(MKHE_H32_X_NOBFLY) || defined(MKHE_H32_X_NOXCHG) || \
                                defined(MKHE_H32_X_NOSTORE) || defined(MKHE_H32_X_NOTWB) || defined(MKHE_X_NOXCHG) || defined(MKHE_X_NO_NTSTORE) || defined(MKHE_H16_X_SCHEDBYTE))
#error "MKHE_*_X_* switches give wrong results on purpose (timing experiments): build them with -DMKHE_ABLATION"
#endif

namespace mkhe {

constexpr int NTT_MAX_ITEMS = 64;
constexpr int NTT_MAX_SLOTS = 48;
constexpr int VI_MAX = 4;       // members of one merged group (NttBatch::vi)

// How job j of a batched launch finds its limb.  Jobs are slot-major: s = j / nouter selects the
// limb slot (modulus mod[s], position pos[s] inside a plain polynomial), outer = j % nouter the
// polynomial / gadget digit, optionally grouped in items with their own base pointers.
// The small lists live in the kernel-argument segment and are read with scalar loads.
struct NttBatch {
    const u64* src;
    u64* dst;
    const Mod* mods;        // [nmod]
    const u64* psi;         // [nmod][N] Montgomery, bit-reversed (forward or inverse table)
    const u64* aux;         // inverse: [nmod][2] = {N^-1 * R, psiinv[1] * N^-1 * R}
    long src_outer, src_inner;   // word strides; limb = base + outer*src_outer + (src_mapped ? mod[s] : pos[s])*src_inner
    long dst_outer, dst_inner;
    int nslots;             // limb slots in this launch
    int nouter;             // polynomials (or items * digits) in this launch
    int src_mapped, dst_mapped;
    int reduce_in;          // forward only: input is a digit spread under a foreign modulus (Decompose)
    int reduce_src_mod_is_outer;   // the digit's own modulus index: 1 = outer (alpha = 1), 2 = outer_mod[outer] (BFV digits of R)
    int src_lazy;           // digit values are lazy base-conversion outputs (< 4 * own modulus) rather than canonical
    int lazy_out;           // inverse only: leave [0,2q) (InvNTTLazy)
    int skip_norm;          // forward, moduli with 34q < 2^63 only: leave the un-normalised values (< 34q, same residues) --
                            // for engine-internal outputs whose only consumers are Montgomery products (hoisted digits)
    int split;              // N = 2^16: the register-resident kernels transform the two halves of a limb as 2^15-point
                            // sub-transforms (twiddle rows of 2^16 words, root index 2 + half); the cross-half radix-2 stage
                            // runs as a separate streaming pass (launch_ntt_* do both)
    unsigned long long small_slots;   // mixed forward launch (launch_ntt_fwd_mixed): bit s set = slot s is a small-modulus (MODE 1) limb
    int prestaged;          // forward, split launches only: 1 = the cross-half stage was already applied by the producer of src
                            // (decomp_spread_kernel with first_stage): only the sub-transforms run, in place on dst.  2 (N = 2^16, H16 only) = the
                            // first TWO stages were (first_stage = 2): what remains are four independent 2^14-point sub-transforms per limb,
                            // split = 2 (twiddle root 4 + quarter), one pass of the H16 kernel each
    int prestaged_oop;      // ... and that producer wrote into src (src_items), not into dst: the sub-transforms read src and write dst
    const u64* psi31;       // forward, ntt16_kernels.hip only: [nmod][N][2] the twiddle w as the constant pair (w 2^31 mod q, w 2^63 mod q),
                            // balanced, radix-2^31 digits -- operands of the one-round product (mm31)
    const u64* psi31c;      // ntt32_kernels.hip (N = 2^15): the pairs of stages 10..14 in the order in which the waves of the single-pass kernel load them,
                            // [nmod][16 waves][31 pairs][64 lanes][2]: every wave-instruction of its last phase reads 1 KiB of consecutive bytes
    const u64* psi31b;      // ntt32_kernels.hip: the pairs of stages 5..9, [nmod][16 waves][2 halves x 31 pairs (+ 2 unused)][2]: one 1 KiB row per wave
    const u64* psi31n;      // [nmod][4][2]: the pairs of -psi[1], -psi[2], -psi[3] (entries 1..3), for the second pass of the cross-half stage
    unsigned long long u_mods;   // bit m set = modulus m is of the U class (160 q < 2^62): its psi31 rows hold the UNSIGNED radix-2^30 format
                            // (u = w 2^30 mod q in [0, q) as digits u0, u1 >= 0; v = w 2^62 mod q balanced) of ntt16_kernels.hip mm30u
    alignas(4) unsigned char sched[NTT_MAX_SLOTS];   // ntt16_kernels.hip, indexed by MODULUS: where a limb of a 59/60-bit modulus gets its partial reductions when its
                            // inputs are below 2^60 (bit 0: at the load, bits 1..3: after phases A, B, C); Context::h16_sched_
    // ntt16_kernels.hip, filled by its launchers: reciprocals for the per-limb index arithmetic (scalar multiply-high instead of VALU
    // divisions) and the placement of the long jobs (the first B jobs of the list: big-modulus limbs) on the CUs that own fewer positions
    int half_jobs;          // the two passes of every limb are separate jobs of the walk (out-of-place N = 2^15 launches)
    unsigned magic_nouter, magic_opi, magic_vi_np;
    // inverse launches on ntt14_inv_kernel (ntt16_kernels.hip): psi31 / psi are then the INVERSE tables, and
    const u64* inv31c;      // [nmod][8][6]: entry `root` = the pair of N^-1, the pair of psiinv[root] N^-1 (N = the whole limb; format of the modulus's class), and both in signed-split Montgomery form
    unsigned long long small_mods;   // bit m set = modulus m has 31 q < 2^62 (no 59/60-bit reduction schedule)
    // the F class (ntt16_kernels.hip limb_f: double-precision butterflies for the quarters of N = 2^16 limbs): bit m set = 80 q < 2^52, and
    const u64* psif;        // [nmod][N] the forward twiddles as PLAIN residues in double format (bit patterns), bit-reversed order; NULL = class off
    unsigned long long f_mods;
    struct Lpt { int B, C, r, full, rem; unsigned magic_C; } lpt;
    int no_h16;             // this context has a modulus the H16 kernel's ranges do not cover (48q >= 2^62 > 31q): keep to the other kernels
    u64* trace;             // diagnostic: per job {start, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID}; normally NULL
    int nitems, outers_per_item;   // nitems > 0: outer = item * outers_per_item + digit, bases from the lists
    int mod[NTT_MAX_SLOTS];
    int pos[NTT_MAX_SLOTS];
    int outer_mod[NTT_MAX_SLOTS];  // reduce_src_mod_is_outer == 2
    const u64* src_items[NTT_MAX_ITEMS];
    u64* dst_items[NTT_MAX_ITEMS];
    // Merged inverse launch (vi != 0; Context::ext_batch): the external products that ModDown adds into ONE destination are
    // linear in their Q limbs -- sum_i (x_i - lift_i) * P^-1 = (sum_i x_i - sum_i lift_i) * P^-1 mod q -- so their Q limbs are
    // summed in the NTT domain and transformed ONCE; only the P limbs (the non-linear lift) are transformed per product.
    // outer = group g of vi_cnt[g] <= VI_MAX members, member k = item vi_mem[g][k] of the buffer src = dst
    // ([item][.. mapped limbs ..][N], item stride src_outer).  Slots [0, vi_q): Q limb mod[s], source = the canonical sum of
    // that limb over the members (formed at the load), written over the first member's limb; slots >= vi_q: P limb mod[s] of
    // member (s - vi_q) / vi_np, in place -- no job when the group has fewer members.  vi_jobs = jobs that exist.
    int vi, vi_q, vi_np, vi_jobs;
    int sum_in_cross;       // merged inverse launch whose sub-transforms ran per MEMBER (ext_fused_lds_kernel<2, true>): the cross pass (ntt_pass8_inv_kernel) sums
                            // the members of a group's Q limbs at its load; no vi_extra summand in such a launch
    // vi_extra[g] != NULL: one more Q-only summand of group g, a plain polynomial [.. limbs ..][N] in the NTT domain, canonical
    // (MulAndRelin's tensor term times P: it leaves ModDown as the tensor term itself, so that it needs no inverse NTT of its own)
    const u64* vi_extra[NTT_MAX_ITEMS];
    int vi_cnt[NTT_MAX_ITEMS];
    unsigned int vi_mem[NTT_MAX_ITEMS];          // item index of member k in bits 8k .. 8k+7 (32-bit lists: see ModDownMergedArgs)
    // Products that reach the launch in PARTS (ntt16_f2_kernel: one partial sum per run of digits): item i of the buffer holds part 0, the items
    // (vi_parts[i] >> 8) + k, k < (vi_parts[i] & 255), the others -- canonical residues of the same limbs, all of them summands of the item's jobs
    // (Q slots: of its group's sum; P slots: of the member's own limb, so that the lift sees the whole product).  0 = the item is complete.
    alignas(4) unsigned short vi_parts[NTT_MAX_ITEMS];
};
constexpr int VI_SUMS = 3 * VI_MAX + 1;   // summands an inverse job may have at its load: members x parts + the Q-only extra (Context::ext_front keeps to it)

// small_q[m] != 0 marks moduli with 34q < 2^63 (forward NTT without in-loop reductions).  The forward
// kernels are specialised per modulus class: split_ntt_fwd cuts the slots of `b` into one batch per
// class (returns how many, small-modulus class first), launch_ntt_fwd_class launches one of them.
int  split_ntt_fwd(const NttBatch& b, const unsigned char* small_q, NttBatch out[2]);
void launch_ntt_fwd_class(int logN, const NttBatch& b, hipStream_t st);
bool ntt_fwd_prestaged_oop_ok(int logN, const NttBatch& b, const unsigned char* small_q);
// both modulus classes of `b` in ONE persistent launch (N = 2^15 Decompose launches that fill the chip several times over);
// ntt_fwd_mixed_ok: false when the launch does not qualify (the caller then issues one launch per class)
bool ntt_fwd_mixed_ok(int logN, const NttBatch& b, const unsigned char* small_q);
void launch_ntt_fwd_mixed(int logN, const NttBatch& b, const unsigned char* small_q, hipStream_t st);
void launch_ntt_inv(int logN, const NttBatch& b, hipStream_t st);
// N = 2^15 launches that fill the chip (ntt16_kernels.hip): 16 coefficients per thread, two workgroups per CU, both modulus
// classes in one persistent launch.  ntt16_ok: false when the launch does not qualify (MKHE_NTT16=0 switches the path off).
bool ntt16_inv_ok(int logN, const NttBatch& b);
void launch_ntt16_inv(const NttBatch& b, hipStream_t st, int logN);
bool ntt16_ok(int logN, const NttBatch& b);
void launch_ntt16_fwd(const NttBatch& b, const unsigned char* small_q, hipStream_t st, int logN = 15);
// N = 2^15 launches of several limbs per CU, one pass per limb (ntt32_kernels.hip: one 1024-thread workgroup per CU, 32 coefficients per thread)
bool ntt32_ok(int logN, const NttBatch& b);
int ntt32_mode();      // MKHE_NTT32: 0 off, 1 on, 2 (default) = Context::ntt_pick times both kernels per launch shape inside the workload and keeps the faster
void launch_ntt32_fwd(const NttBatch& b, const unsigned char* small_q, hipStream_t st);
// the same kernel on the 2^15-point sub-transforms of a split N = 2^16 launch (one modulus class per launch)
bool ntt16_split_ok(const NttBatch& c);
void launch_ntt16_fwd_split(const NttBatch& c, bool small, hipStream_t st);


// Small rings (N = 2^14), small launches: the forward sub-transforms of an engine-internal Decompose and the inner products that consume them in ONE
// kernel (round 5).  A key switch of one ciphertext there is a chain of launches of a few dozen limbs that last as long as one workgroup each; the
// digits of a polynomial the engine decomposes for its own use (Rotate / Conjugate without a hoisted form, step F2 of MulAndRelin) are read exactly
// once, by the inner products with its one or two keys.  launch_ntt_cross8_dec leaves them after the three cross stages (ntt_pass8_fwd_kernel<true>:
// digit spread + radix-8 pass, eight independent 2^11-point blocks per limb); ext_fused_lds_kernel<NG> then gives a workgroup one (vector, limb, block):
// NG groups of 256 threads transform the block of NG digits at a time in LDS (the phases of ntt_fwd_lds_kernel<MODE, 11>, the last one back into
// LDS), every thread then multiplies 8 / NG coefficients of all NG digits with the key(s) and keeps the sums -- the same integers as
// ntt_fwd_lds_kernel + ext_inner[_group]_kernel (canonical sums of the same products), without the store and the reload of the digits and one launch less.
constexpr int EXTF_MAX_V = 32;
struct ExtFusedArgs {
    const u64* stage[EXTF_MAX_V];          // per vector: the cross-passed digits [digit][mtot][N] (a hoisted-form buffer)
    const u64* bg[EXTF_MAX_V][2];          // its one or two keys (gadget ciphertext halves), [digit][mtot][N], Montgomery form
    u64* out[EXTF_MAX_V][2];               // their products' limbs base ([mtot][N]: a c1 slot), canonical
    unsigned char nk[EXTF_MAX_V];
    const Mod* mods;
    const int* map;                        // limb slot -> modulus index
    const u64* psi;                        // forward twiddles [nmod][N]
    unsigned long long small_mask;         // bit m: modulus m runs the signed never-reduced butterflies (MODE 1), else Harvey (MODE 0)
    const u64* psiinv;                     // inv != 0: the inverse twiddles [nmod][N] and
    const u64* aux;                        // the inverse constants (NttBatch::aux)
    int inv;                               // the products leave the kernel after their inverse 2^11-point sub-transforms ([0, 2q), N^-1 folded in): the
                                           // caller follows with launch_ntt_inv_cross8_sum instead of launch_ntt_inv (always two groups of threads)
    long digit_stride;
    int nb, nslots, N, logN, nv;
};
// N = 2^15, alpha = 1 (round 6): step F2 of MulAndRelin inside the Decompose NTT of the t_i (ntt16_f2_kernels.hip).  The digits of t_i are read exactly
// once, by the products with v_i and u (mkrlwe/keyswitch_hoisted.go:170-177): a workgroup of ntt16_f2_kernel owns a RUN of digits of one (party, modulus,
// half limb) -- a segment -- transforms digit after digit as a pass of the H16 kernel does (same modulus for the whole run: the twiddles stay put), and
// multiplies the sixteen coefficients a thread holds with the two keys into two register accumulators.  The digits never reach HBM (470 MB written and
// read per 4-party MulRelin on PN15QP880) and the streaming launch that read them is gone; what leaves the kernel are the canonical partial sums of the
// runs, one c1 item per (product, part): part 0 in the product's own item, parts 1.. in the items extra[product] + part - 1, and the inverse NTT that
// follows adds the parts of a product at its load (NttBatch::vi_parts) -- sums of canonical residues of the same limb: the same integers as
// ext_inner_kernel on the stored digits.  The schedule (which workgroup walks which runs) is a host table balanced by the cost of the modulus classes.
constexpr int F2_MAX_P = 16;    // parties of op0
constexpr int F2_SEGS = 3;      // runs per workgroup (a run of fewer than twice the digits of a vector touches at most three groups)
struct F2Seg { unsigned char party, slot, half, d0, nd, part, pad0, pad1; };      // nd = 0: no run
struct F2FusedArgs {
    const F2Seg* segs;              // device, [nwg][F2_SEGS]
    const u64* src[F2_MAX_P];       // t_p in the coefficient domain, [>= nb][N]: digit d = limb d
    const u64* kv[F2_MAX_P];        // v_p [digit][mtot][N], Montgomery form
    const u64* ku;                  // the CRS u
    u64* c1;                        // products [item][mtot][N]
    int item_v[F2_MAX_P], item_u[F2_MAX_P];       // c1 item of part 0 of <h(t_p), v_p> / <h(t_p), u>
    int extra_v[F2_MAX_P], extra_u[F2_MAX_P];     // c1 item of part 1 (parts 2.. follow)
    long item_words, digit_stride;
    const Mod* mods;
    const u64* psi;                 // [nmod][N] forward twiddles (phase D: two-round product)
    const u64* psi31;               // [nmod][N][2] pairs of the one-round products
    const u64* psi31n;              // [nmod][4][2] pairs of -psi[1..3]
    unsigned long long u_mods;      // bit m: modulus m is of the U class
    unsigned long long small_mask;  // bit m: 48 q < 2^62 (no partial reductions)
    alignas(4) unsigned char sched[NTT_MAX_SLOTS];   // NttBatch::sched
    int mod[NTT_MAX_SLOTS];         // limb slot -> modulus index
    int nwg;
    u64* trace;                     // diagnostic (make trace): [workgroup][wave][pass][16] shader-clock stamps; normally NULL
};
bool ntt16_f2_ok(int logN, int nparties, int nb, int nslots);
int ntt16_f2_mode();              // MKHE_F2_FUSED (diagnostic library): 0 off, 1 planned grid (default), 2 one workgroup per CU or nothing
// parts a product may arrive in: the inverse job of a merged destination adds (members x parts + the tensor term [+ a step-E member]) <= VI_SUMS summands at
// its load; out_0 has one member per party of op0 (VI_MAX at most per job)
constexpr int F2_PARTS_MAX = 11;
constexpr int f2_max_parts(int np0) {
    const int members = np0 < 1 ? 1 : (np0 < VI_MAX ? np0 : VI_MAX);
    const int a = (VI_SUMS - 1) / members, b = VI_SUMS - 2;      // (a party's own slot: its u product's parts + the step-E member + the tensor term)
    const int r = a < b ? a : b;
    return r > F2_PARTS_MAX ? F2_PARTS_MAX : r;
}
constexpr long F2_RUN_COST = 80, F2_PART_COST = 8, F2_SLACK = 190;       // in units of a pass = 100 (f2_plan_schedule)
int f2_build_schedule(int np0, int nb, int nslots, const long* weights, int G, F2Seg* segs, int* parts_out, int max_parts = 0, long* cost_out = nullptr, long* worst_out = nullptr);      // engine_mulrelin.hip
int f2_plan_schedule(int np0, int nb, int nslots, const long* weights, int Gmax, F2Seg* segs, int* parts_out, int max_parts = 0);
void launch_ntt16_f2(const F2FusedArgs& a, hipStream_t st);
int ntt16_f2_grid();                // workgroups the schedule should be cut for (one per CU)
void launch_ntt_cross8_dec(const NttBatch& b, int logN, hipStream_t st);
void launch_ext_fused_lds(const ExtFusedArgs& a, hipStream_t st);
void launch_ntt_inv_cross8_sum(const NttBatch& b, int logN, hipStream_t st);

}  // namespace mkhe
