"""Decompose-fused forward NTT at N = 2^15: time per launch for the launch sizes of a 4-party MulRelin (1792 and 896 limbs),
HIP events on the context stream (mkhe_prof_*), one line per launch size.  Usage: [MKHE_LIB=...] python tools/ntt16_bench.py [reps]
Also checks the hoisted digits of one party against a second evaluation with MKHE_NTT16 toggled in a child process (--ref file).
N = 2^15 has two forward kernels: MKHE_NTT32=0 / 1 time one of them; with the default (2) the engine measures both inside this loop and settles, so
both appear, each with its own launch count (tools/profile_round.sh runs this tool once per kernel)."""
import ctypes as C, os, sys, time, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import harness as H
from mkhe_kklss_amd import mkckks
from mkhe_kklss_amd._abi import check, lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
p = H.PN15QP880
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
check(lib().mkhe_set_overlap(params.ctx, 0))
rng = np.random.default_rng(5)
N, level = 1 << p["logN"], len(p["Q"]) - 1
ev = mkckks.NewEvaluator(params)
ncls = lib().mkhe_prof_nclass()
names = [lib().mkhe_prof_name(i).decode() for i in range(ncls)]
for k in (8, 4, 2, 1):
    ids = ["u%d" % i for i in range(k)]
    host = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"]]) for _ in range(1 + k)])
    ct = mkckks.NewCiphertext(params, ids, level, p["scale"]).upload(host)
    for _ in range(3):
        h = ev.HoistedForm(ct)
    params.sync()
    check(lib().mkhe_prof_enable(params.ctx, 1))
    for _ in range(reps):
        h = ev.HoistedForm(ct)
    ms = (C.c_double * ncls)(); cnt = (C.c_long * ncls)(); byt = (C.c_double * ncls)()
    check(lib().mkhe_prof_collect(params.ctx, ms, cnt, byt))
    check(lib().mkhe_prof_enable(params.ctx, 0))
    dig = hashlib.sha256(h.Value[ids[-1]].download().tobytes()).hexdigest()[:16]
    for i in range(ncls):
        if cnt[i]:
            us = 1e3 * ms[i] / cnt[i]
            print("limbs %5d  %-28s launches %3d  %8.1f us/launch  %6.1f GB/s (16N B/limb)  frac %.3f  digest %s" % (
                k * 224, names[i].split("  ")[0], cnt[i], us, byt[i] / (ms[i] * 1e-3) / 1e9, byt[i] / (ms[i] * 1e-3) / 8e12, dig), flush=True)
