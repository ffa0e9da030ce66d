#!/usr/bin/env python3
"""Random circuits on the mkckks Evaluator surface, device against the CPU oracle evaluator (tests/oracle_evaluator.py), on the GPU box:

    gpurun --timeout 900 -- 'python3 tools/fuzz_circuit.py 420 > gpurun_out/fuzz_circuit.txt'

Four parties with keys made on the device (tests/harness_cnn.py CnnScenario: PN14QP433, the cnn ring), a pool of ciphertexts under different id sets and levels;
every step draws an operation -- MulRelinNew, RotateNew by an index with and without its own rotation key (power-of-two walk), rotate-and-add (one engine call on
the device: Evaluator.RotateAndAddNew; RotateNew + AddNew on the oracle), a layer's worth of lanes (RotateHoisted -> HoistedForm -> MulRelinHoisted chains through
Evaluator.Lanes + SumNew, as cnn.Convolution does) -- applies it on both sides, compares every word of the result, and puts it back into the pool.  Not part of the
test suite (minutes of oracle time); exit code 1 on the first mismatch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import harness_cnn as HC                                  # noqa: E402
import oracle_evaluator as OE                             # noqa: E402
from oracle import oracle as O                            # noqa: E402
from mkhe_kklss_amd import cnn                            # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4242
rng = np.random.default_rng(seed)
p = HC.PN14QP433
owners = dict(image="alice", kernels="bob", fc1="carol", fc2="dave")
sc = HC.CnnScenario(owners, seed=seed % 1000)
parties = sorted(set(owners.values()))
rots = sorted(set(HC.ROTS + [1 << i for i in range(p["logN"] - 1)]))
rlk_h = {id: tuple(sc.rlkSet.GetRelinearizationKey(id).Value[j].download() for j in range(3)) for id in parties}
rk_h = {(id, r): sc.rtkSet.GetRotationKey(id, r).Value.download() for id in parties for r in rots}
crs_h = {r: sc.params.CRS[r].download() for r in rots + [-1] if r in sc.params.CRS}
O.set_threads(min(16, os.cpu_count() or 1))
oev = OE.OracleEvaluator(O.KeySwitcher(p["logN"], p["Q"], p["P"], 2), p["Q"], p["scale"], rlk_h, rk_h, crs_h, p["logN"])
dev = sc.eval
mirror = lambda c: OE.OCt(c.ids, c.download(), c.Scale)


def fresh():
    id = parties[int(rng.integers(len(parties)))]
    c = sc.encrypt(rng.uniform(-1, 1, 1 << (p["logN"] - 1)), id)
    return c, mirror(c)


def plain_lane_products(ev, ct, rots_, others):
    """the chains of cnn.Convolution one operation at a time (cnn/cnn.go:16-30): what Evaluator.Lanes issues as one launch set"""
    h = ev.HoistedForm(ct)
    out = []
    for r, other in zip(rots_, others):
        temp, th = (ct, h) if r == 0 else (None, None)
        if r != 0:
            temp = ev.RotateHoistedNew(ct, r, h, None)
            th = ev.HoistedForm(temp)
        out.append(ev.MulRelinHoistedNew(temp, other, th, ev.HoistedForm(other), None))
    return out


pool = [fresh() for _ in range(6)]
t0, n, counts = time.time(), 0, {}
keyed = [r for r in rots if r in crs_h]
while time.time() - t0 < budget:
    op = ["mul", "rot", "rotadd", "rotadd", "lanes"][int(rng.integers(5))]
    i = int(rng.integers(len(pool)))
    d, o = pool[i]
    if d.Level() == 0 or (op in ("mul", "lanes") and d.Level() < 1):
        pool[i] = fresh(); continue
    if op == "mul":
        j = int(rng.integers(len(pool)))
        d2, o2 = pool[j]
        if d2.Level() < 1:
            pool[j] = fresh(); continue
        rd, ro = dev.MulRelinNew(d, d2, sc.rlkSet), oev.MulRelinNew(o, o2, None)
        desc = "MulRelinNew %s x %s  levels %d, %d" % (",".join(d.ids), ",".join(d2.ids), d.Level(), d2.Level())
    elif op == "rot":
        r = int(rng.integers(1, 1 << (p["logN"] - 1))) if rng.integers(2) else keyed[int(rng.integers(len(keyed)))]
        if bin(r).count("1") > 4 and r not in crs_h:
            r &= 0x30F                                         # (keep the power-of-two walk short)
            r = r or 3
        rd, ro = dev.RotateNew(d, r, sc.rtkSet), oev.RotateNew(o, r, None)
        desc = "RotateNew %d  %s level %d" % (r, ",".join(d.ids), d.Level())
    elif op == "rotadd":
        r = keyed[int(rng.integers(len(keyed)))]
        rd, ro = cnn._rot_add(dev, d, r, sc.rtkSet), cnn._rot_add(oev, o, r, None)
        desc = "rotate-and-add %d  %s level %d" % (r, ",".join(d.ids), d.Level())
    else:
        nl = int(rng.integers(2, 5))
        lr = [0] + [keyed[int(rng.integers(len(keyed)))] for _ in range(nl - 1)]
        others = [pool[int(rng.integers(len(pool)))]] * nl       # (the lanes of a launch set share the other operand's shape and scale: one pool entry)
        if others[0][0].Level() < 1:
            continue
        od, oo = [x[0] for x in others], [x[1] for x in others]
        pd = cnn._lane_products(dev, sc.rlkSet, sc.rtkSet, d, dev.HoistedForm(d), lr, od, [dev.HoistedForm(x) for x in od])
        po = plain_lane_products(oev, o, lr, oo)
        rd, ro = cnn._sum(dev, pd), cnn._sum(oev, po)
        desc = "lanes %s  %s level %d x %s level %d" % (lr, ",".join(d.ids), d.Level(), ",".join(od[0].ids), od[0].Level())
    ok = list(rd.ids) == list(ro.ids) and rd.Level() == ro.Level() and rd.Scale == ro.Scale and bool((rd.download() == ro.host).all())
    n += 1
    counts[op] = counts.get(op, 0) + 1
    print("%4d %s  %s -> %s level %d" % (n, "ok  " if ok else "MISMATCH", desc, ",".join(rd.ids), rd.Level()), flush=True)
    if not ok:
        print("seed %d" % seed); sys.exit(1)
    pool[int(rng.integers(len(pool)))] = (rd, ro) if rd.Level() >= 1 else fresh()
print("# %d operations in %.0f s, every result bit-exact against the oracle evaluator (seed %d): %s" % (n, time.time() - t0, seed, ", ".join("%s %d" % kv for kv in sorted(counts.items()))))
