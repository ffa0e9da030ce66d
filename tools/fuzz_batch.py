#!/usr/bin/env python3
"""The lock-step entry points (mkhe_*_batch, mkhe_rotate_multi: BatchEvaluator, Evaluator.Lanes) against the single-operation ones on random shapes, on the GPU box:

    gpurun --timeout 900 -- 'python3 tools/fuzz_batch.py 300 > gpurun_out/fuzz_batch.txt'

The single-operation entry points are held against the CPU oracle by the test suite and tools/fuzz_parity.py / fuzz_circuit.py; this run holds the batched
ones against THEM (device against device: thousands of cases per minute): B = 1..9 inputs of one random shape (cnn ring, one to four parties, level 0..6) through
MulRelinNew / MulRelinHoistedNew (operands distinct, equal, or one broadcast), RotateNew (an index with its own key or a power-of-two walk), RotateHoistedNew with one
index per input (lanes), RotateAndAddNew, AddNew, HoistedForm -- every output word compared with the single evaluator's.  Exit code 1 on the first mismatch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import harness_cnn as HC                                  # noqa: E402
from mkhe_kklss_amd import mkckks                          # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 9001
rng = np.random.default_rng(seed)
p = HC.PN14QP433
owners = dict(image="alice", kernels="bob", fc1="carol", fc2="dave")
sc = HC.CnnScenario(owners, seed=seed % 1000)
parties = sorted(set(owners.values()))
ev = sc.eval
N = 1 << p["logN"]
keyed = sorted(r for r in sc.params.CRS if r > 0)
bevs = {}


def bev(B):
    if B not in bevs:
        bevs[B] = mkckks.BatchEvaluator(sc.params, B, ev=ev)
    return bevs[B]


def random_ct(ids, level, scale):
    """uniform residues under the given shape (the engine does not care that they decrypt to nothing)"""
    h = np.empty((1 + len(ids), level + 1, N), dtype=np.uint64)
    for l in range(level + 1):
        h[:, l] = rng.integers(0, p["Q"][l], (1 + len(ids), N), dtype=np.uint64)
    return mkckks.NewCiphertext(sc.params, ids, level, scale).upload(h)


def same(batch, singles):
    return all(b.ids == s.ids and b.Level() == s.Level() and b.Scale == s.Scale and bool((b.download() == s.download()).all()) for b, s in zip(batch.cts, singles))


t0, n, counts = time.time(), 0, {}
while time.time() - t0 < budget:
    B = int(rng.integers(1, 10))
    k = int(rng.integers(1, 5))
    ids = sorted(rng.choice(parties, size=k, replace=False).tolist())
    level = int(rng.integers(0, len(p["Q"])))
    op = ["mul", "mulh", "rot", "lanes", "rotadd", "add", "hoist"][int(rng.integers(7))]
    if op in ("mul", "mulh") and level < 1:
        continue
    be = bev(B)
    xs = [random_ct(ids, level, p["scale"]) for _ in range(B)]
    bx = mkckks.BatchCiphertext(xs)
    desc = "%-6s B %d ids %s level %d" % (op, B, ",".join(ids), level)
    if op in ("mul", "mulh"):
        mode = int(rng.integers(3))                      # 0: distinct second operands, 1: the square, 2: one broadcast operand (the model of cnn)
        ids2 = ids if mode == 1 else sorted(rng.choice(parties, size=int(rng.integers(1, 5)), replace=False).tolist())
        if mode == 0:
            ys = [random_ct(ids2, level, p["scale"]) for _ in range(B)]; by = mkckks.BatchCiphertext(ys)
        elif mode == 1:
            ys, by = xs, bx
        else:
            y1 = random_ct(ids2, level, p["scale"]); ys, by = [y1] * B, y1
        if op == "mulh":
            hx, hy = be.HoistedForm(bx), (be.HoistedForm(by) if mode != 1 else None)
            if mode == 1:
                hy = hx
            out = be.MulRelinHoistedNew(bx, by, hx, hy, sc.rlkSet)
            ref = [ev.MulRelinHoistedNew(x, y, ev.HoistedForm(x), ev.HoistedForm(y), sc.rlkSet) for x, y in zip(xs, ys)]
        else:
            out = be.MulRelinNew(bx, by, sc.rlkSet)
            ref = [ev.MulRelinNew(x, y, sc.rlkSet) for x, y in zip(xs, ys)]
        desc += " x %s (%s)" % (",".join(ids2), ["distinct", "square", "broadcast"][mode])
    elif op == "rot":
        r = keyed[int(rng.integers(len(keyed)))] if rng.integers(2) else int(rng.integers(1, 64))
        out = be.RotateNew(bx, r, sc.rtkSet)
        ref = [ev.RotateNew(x, r, sc.rtkSet) for x in xs]
        desc += " by %d" % r
    elif op == "lanes":
        rs = [keyed[int(rng.integers(len(keyed)))] for _ in range(B)]
        hoisted = bool(rng.integers(2))
        hx = be.HoistedForm(bx) if hoisted else None
        out = be.RotateHoistedNew(bx, rs, hx, sc.rtkSet)
        ref = [ev.RotateHoistedNew(x, r, ev.HoistedForm(x), sc.rtkSet) for x, r in zip(xs, rs)]
        desc += " by %s%s" % (rs, " hoisted" if hoisted else "")
    elif op == "rotadd":
        r = keyed[int(rng.integers(len(keyed)))]
        out = be.RotateAndAddNew(bx, r, sc.rtkSet)
        ref = [ev.AddNew(x, ev.RotateNew(x, r, sc.rtkSet)) for x in xs]
        desc += " by %d" % r
    elif op == "add":
        ids2 = sorted(rng.choice(parties, size=int(rng.integers(1, 5)), replace=False).tolist())
        l2 = int(rng.integers(0, len(p["Q"])))
        ys = [random_ct(ids2, l2, p["scale"]) for _ in range(B)]
        out = be.AddNew(bx, mkckks.BatchCiphertext(ys))
        ref = [ev.AddNew(x, y) for x, y in zip(xs, ys)]
        desc += " + %s level %d" % (",".join(ids2), l2)
    else:
        hb = be.HoistedForm(bx)
        ok = True
        act = list(range(level + 1)) + [len(p["Q"]) + j for j in range(len(p["P"]))]        # (digits and limbs beyond the level are not written)
        for x, h in zip(xs, hb.hoisted):
            hs = ev.HoistedForm(x)
            ok = ok and all(bool((h.Value[i].download()[:level + 1][:, act] == hs.Value[i].download()[:level + 1][:, act]).all()) for i in ids)
        out = None
    if out is not None:
        ok = same(out, ref)
    n += 1
    counts[op] = counts.get(op, 0) + 1
    print("%5d %s  %s" % (n, "ok  " if ok else "MISMATCH", desc), flush=True)
    if not ok:
        print("seed %d" % seed); sys.exit(1)
print("# %d batched calls in %.0f s, every output identical to the single-operation entry point's (seed %d): %s" % (n, time.time() - t0, seed, ", ".join("%s %d" % kv for kv in sorted(counts.items()))))
