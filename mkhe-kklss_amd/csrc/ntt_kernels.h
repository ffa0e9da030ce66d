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
// LDS word address = p ^ ((p >> 5) & 31) (p = coefficient index): conflict-free for every
// layout above under the 32-bank rule of ds_read_b32 / ds_write_b32.
//
// Replaces: lattigo ring.NTTLvl / InvNTTLvl / InvNTTLazyLvl as called from
// mkrlwe/keyswitch.go:29-30,58,88,114-115,206 and keyswitch_hoisted.go:36-37,120-143.
#pragma once
#include "modarith.h"

namespace mkhe {

// How job j of a batched launch finds its limb.  job = outer * inner_count + s ;
// m = map[s] is the modulus index (and the limb slot inside a PolyQP-shaped buffer).
struct NttBatch {
    const u64* src;
    u64* dst;
    const Mod* mods;        // [nmod]
    const u64* psi;         // [nmod][N] Montgomery, bit-reversed (forward or inverse table)
    const u64* aux;         // inverse: [nmod][2] = {N^-1 * R, psiinv[1] * N^-1 * R}
    const int* map;         // [inner_count]
    long src_outer, src_inner;   // word strides; src limb = src + outer*src_outer + (src_mapped ? m : s)*src_inner
    long dst_outer, dst_inner;
    int inner_count;
    int njobs;
    int src_mapped, dst_mapped;
    int reduce_in;          // forward only: input is a digit spread under a foreign modulus (Decompose)
    int reduce_src_mod_is_outer;   // the digit's own modulus index = outer (alpha = 1)
    int lazy_out;           // inverse only: leave [0,2q) (InvNTTLazy)
};

void launch_ntt_fwd(int logN, const NttBatch& b, hipStream_t st);
void launch_ntt_inv(int logN, const NttBatch& b, hipStream_t st);

}  // namespace mkhe
