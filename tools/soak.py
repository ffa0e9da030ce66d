#!/usr/bin/env python3
"""Determinism soak (GPU box): the same MulRelin evaluated `reps` times on resident operands, every result compared word for word with the first.
A race in a kernel (a missing barrier between LDS phases, a pool buffer reused too early) shows up as a run-to-run difference long before it shows
up against the oracle.   python tools/soak.py [PN15QP880|PN14QP439|PN16QP1761] [parties] [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H
from bench import synth_cts
from mkhe_kklss_amd import mkckks, mkrlwe
from mkhe_kklss_amd._abi import check, lib

pname = sys.argv[1] if len(sys.argv) > 1 else "PN15QP880"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
pset = {"PN15QP880": H.PN15QP880, "PN14QP439": H.PN14QP439, "PN16QP1761": H.PN16QP1761}[pname]
seed = 0x50414B
names = ["user%d" % i for i in range(k)]
params = mkckks.Parameters(pset["logN"], pset["Q"], pset["P"], pset["scale"], device=0)
level = len(pset["Q"]) - 1
params.AddCRS(-1, seed=seed)
rlk = mkrlwe.RelinearizationKeySet(params)
for i, n in enumerate(names):
    key = mkrlwe.RelinearizationKey(params, n)
    for j in range(3):
        check(lib().mkhe_crs_expand(params.ctx, seed, 1000 + 3 * i + j, key.Value[j].h))
    rlk.AddRelinearizationKey(key)
op0, op1 = synth_cts(pset, k, seed)
ct0 = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(op0)
ct1 = mkckks.NewCiphertext(params, names, level, pset["scale"]).upload(op1)
ev = mkckks.NewEvaluator(params)
first = ev.MulRelinNew(ct0, ct1, rlk).download()
rot = None
bad = 0
t0 = time.perf_counter()
for r in range(reps):
    got = ev.MulRelinNew(ct0, ct1, rlk).download()
    if not (got == first).all():
        bad += 1
        print("rep %d differs in %d words" % (r, int((got != first).sum())), flush=True)
    if r % 50 == 49:
        print("rep %d, %d differing so far, %.1f s" % (r + 1, bad, time.perf_counter() - t0), flush=True)
print("%s k=%d: %d repetitions, %d differ from the first" % (pname, k, reps, bad))
sys.exit(1 if bad else 0)
