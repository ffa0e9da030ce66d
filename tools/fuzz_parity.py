#!/usr/bin/env python3
"""Random-shape parity run on the GPU box (not part of the test suite: minutes of oracle time):

    gpurun --timeout 900 -- 'python3 tools/fuzz_parity.py 420 [seed] [wide] > gpurun_out/fuzz.txt'      (wide: alpha = 2 chains and N = 2^15 among the rings; pn15: PN15QP880 only, full chain)

For the given number of seconds: draw a ring (N = 2^14 with the PN14QP439 chain -- the ring of the fused small-ring kernel -- or N = 2^12 / 2^13 with a
reduced chain), a level, one to four parties, an input level at or above the output's, an operation (Rotate, Conjugate, MulAndRelin with random -- equal,
overlapping, disjoint -- id sets, with or without caller-supplied hoisted forms, plain or with the fused Rescale; one case in six: mkbfv MulRelinNew or its
non-hoisted twin on N = 2^12 / 2^13), run it through the C ABI and through
the CPU oracle on the same seeded inputs, and compare every output word.  Prints one line per case and a summary; exit code 1 on the first mismatch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import harness as H                                       # noqa: E402
from gpu_common import Pair, oracle_mul_and_relin         # noqa: E402
from oracle import oracle as O                            # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20261004
rng = np.random.default_rng(seed)
O.set_threads(min(16, os.cpu_count() or 1))
SETS = {"N14": H.PN14QP439, "N13": H.small_ckks(13, 5), "N12": H.small_ckks(12, 4),
        # alpha = 2 (CRT-reconstructed digits), and the headline ring with a short chain
        "A12": H.small_alpha2(12, 5), "A11": H.small_alpha2(11, 4), "N15": H.small_ckks(15, 3)}
pairs = {k: Pair(v, seed=int(rng.integers(1 << 30))) for k, v in SETS.items()}
import harness_bfv as HB                                 # noqa: E402
from test_gpu_bfv import BfvPair                          # noqa: E402
bfv_pairs = {"B12": BfvPair(HB.small_bfv(12, 3), seed=int(rng.integers(1 << 30))), "B13": BfvPair(HB.small_bfv(13, 4), seed=int(rng.integers(1 << 30)))}
RINGS = ["N14", "N14", "N14", "N13", "N12"] if len(sys.argv) <= 3 or sys.argv[3] not in ("wide", "pn15") else ["N14", "N13", "N12", "A12", "A12", "A11", "N15"]
if len(sys.argv) > 3 and sys.argv[3] == "pn15":          # the headline ring with its full chain (seconds of oracle time per case)
    pairs["PN15"] = Pair(H.PN15QP880, seed=int(rng.integers(1 << 30))); RINGS = ["PN15"]
t0, n, counts = time.time(), 0, {}
while time.time() - t0 < budget:
    if rng.integers(6) == 0:
        # mkbfv MulRelinNew / its non-hoisted twin on random id sets (mkbfv/evaluator.go:84-150)
        ring = ["B12", "B13"][int(rng.integers(2))]
        bp = bfv_pairs[ring]
        pool = ["a", "b", "c", "d", "e"]
        ids0 = sorted(rng.choice(pool, size=int(rng.integers(0, 4)), replace=False).tolist())
        ids1 = ids0 if rng.integers(3) == 0 else sorted(rng.choice(pool, size=int(rng.integers(0, 4)), replace=False).tolist())
        alln = sorted(set(ids0) | set(ids1)); idx = {x: i for i, x in enumerate(alln)}
        h0, c0 = bp.ct(ids0); h1, c1 = (h0, c0) if ids1 is ids0 and rng.integers(2) else bp.ct(ids1)
        rlk_h, rlk_d = bp.rlk_set(alln)
        u_h, u_d = bp.swk(); bp.params.CRS[-1] = u_d
        twin = bool(rng.integers(2))
        out = bp.ev.mulRelin(c0, c1, rlk_d) if twin else bp.ev.MulRelinNew(c0, c1, rlk_d)
        ido, ref = bp.bfv.mul_relin_new([idx[i] for i in ids0], h0, [idx[i] for i in ids1], h1, {idx[x]: rlk_h[x] for x in alln}, u_h)
        ok = out.ids == [alln[i] for i in ido] and bool((out.download() == ref).all())
        n += 1; counts[(ring, "bfv")] = counts.get((ring, "bfv"), 0) + 1
        print("%4d %s  %s bfv %s ids %s x %s" % (n, "ok  " if ok else "MISMATCH", ring, "mulRelin" if twin else "MulRelinNew", "".join(ids0) or "-", "".join(ids1) or "-"), flush=True)
        if not ok:
            print("seed %d" % seed); sys.exit(1)
        continue
    ring = RINGS[int(rng.integers(len(RINGS)))]
    pr = pairs[ring]; mk = pr.mk
    level = int(rng.integers(0, pr.maxlevel + 1))
    in_level = min(pr.maxlevel, level + int(rng.integers(0, 2)))
    op = ["rotate", "conjugate", "mulrelin", "mulrelin"][int(rng.integers(4))]
    k = int(rng.integers(1, 5))
    names = ["p%d" % i for i in range(k)]
    desc = "%s %-9s level %d (in %d) parties %d" % (ring, op, level, in_level, k)
    if op in ("rotate", "conjugate"):
        h, d = pr.ct(names, in_level)
        crs_h = H.uniform_swk(pr.rng, pr.ks)
        if op == "rotate":
            rot = int(rng.integers(1, 9))
            pr.params.AddCRS(rot, crs_h)
            kset, k_h = mk.RotationKeySet(), []
            for i in names:
                kk = H.uniform_swk(pr.rng, pr.ks); k_h.append(kk)
                kset.AddRotationKey(mk.RotationKey(pr.params, rot, i, kk))
            out = mk.NewCiphertext(pr.params, names, level)
            hoisted = bool(rng.integers(2)) and in_level == level
            if hoisted:
                hh = mk.NewHoistedCiphertext()
                for i in names:
                    hh.Value[i] = mk.NewSwitchingKey(pr.params)
                    pr.ksw.Decompose(level, d, i, hh.Value[i])
                pr.ksw.RotateHoisted(d, rot, hh, kset, out)
            else:
                pr.ksw.Rotate(d, rot, kset, out)
            ref = pr.ks.rotate(level, pow(5, rot, 2 * pr.N), list(range(k)), h, k_h, crs_h)
            desc += " rot %d%s" % (rot, " hoisted" if hoisted else "")
        else:
            pr.params.AddCRS(-2, crs_h)
            kset, k_h = mk.ConjugationKeySet(), []
            for i in names:
                kk = H.uniform_swk(pr.rng, pr.ks); k_h.append(kk)
                kset.AddConjugationKey(mk.ConjugationKey(pr.params, i, kk))
            out = mk.NewCiphertext(pr.params, names, level)
            pr.ksw.Conjugate(d, kset, out)
            ref = pr.ks.conjugate(level, 2 * pr.N - 1, list(range(k)), h, k_h, crs_h)
        ok = bool((out.download() == ref).all())
    else:
        pool = ["p%d" % i for i in range(6)]
        ids0 = sorted(rng.choice(pool, size=int(rng.integers(1, 5)), replace=False).tolist())
        ids1 = ids0 if rng.integers(3) == 0 else sorted(rng.choice(pool, size=int(rng.integers(1, 5)), replace=False).tolist())
        alln = sorted(set(ids0) | set(ids1))
        h0, d0 = pr.ct(ids0, level, in_level + 1)
        h1, d1 = pr.ct(ids1, level, in_level + 1)
        rlk_h, rlk_d = pr.rlk_set(alln)
        u_h = H.uniform_swk(pr.rng, pr.ks)
        pr.params.AddCRS(-1, u_h)
        hoisted = bool(rng.integers(2)) and in_level == level
        rescaled = bool(rng.integers(2)) and level >= 1
        out = mk.NewCiphertext(pr.params, alln, level - 1 if rescaled else level)
        hh0 = hh1 = None
        if hoisted:
            hh0, hh1 = mk.NewHoistedCiphertext(), mk.NewHoistedCiphertext()
            for i in ids0:
                hh0.Value[i] = mk.NewSwitchingKey(pr.params); pr.ksw.Decompose(level, d0, i, hh0.Value[i])
            for i in ids1:
                hh1.Value[i] = mk.NewSwitchingKey(pr.params); pr.ksw.Decompose(level, d1, i, hh1.Value[i])
        if rescaled:
            pr.ksw.MulAndRelinHoisted(d0, d1, hh0, hh1, rlk_d, out, rescaled=True)
        elif hoisted:
            pr.ksw.MulAndRelinHoisted(d0, d1, hh0, hh1, rlk_d, out)
        else:
            pr.ksw.MulAndRelin(d0, d1, rlk_d, out)
        ido, ref = oracle_mul_and_relin(pr, level, ids0, h0, ids1, h1, rlk_h, u_h, alln)
        if rescaled:
            ref = np.stack([pr.ks.ringQ.div_round_last_many(ref[s], 1)[0] for s in range(1 + len(alln))])
        ok = ido == out.ids and bool((out.download() == ref).all())
        desc += " ids %s x %s%s%s" % ("".join(i[1] for i in ids0), "".join(i[1] for i in ids1), " hoisted" if hoisted else "", " +rescale" if rescaled else "")
    n += 1
    counts[(ring, op)] = counts.get((ring, op), 0) + 1
    print("%4d %s  %s" % (n, "ok  " if ok else "MISMATCH", desc), flush=True)
    if not ok:
        print("seed %d" % seed)
        sys.exit(1)
print("# %d cases in %.0f s, all bit-exact against the oracle (seed %d): %s" % (n, time.time() - t0, seed, ", ".join("%s %s %d" % (a, b, c) for (a, b), c in sorted(counts.items()))))
