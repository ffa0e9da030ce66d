"""The H16 kernels (forward and inverse) take launches from a size threshold up; the thresholds are read once per process.  This test re-runs a few
small-shape operations in a subprocess with every threshold at 1, so that the shapes only small launches have -- ONE polynomial per launch (nouter = 1),
one gadget digit per item (level 0: outers_per_item = 1), merged inverse launches of a dozen limbs -- go through those kernels' job walk too.
(Round 3: the reciprocal of the scalar job-index division had no representation for a divisor of 1; an inverse launch with nouter = 1 then walked
off its buffer.  Found by running the whole GPU suite with MKHE_NTT16_INV_MIN=1.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys
import numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import harness as H
from oracle import oracle as O
from mkhe_kklss_amd import mkckks, mkrlwe
p = H.PN14QP439
ks = O.KeySwitcher(p["logN"], p["Q"], p["P"], 2)
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
rng = np.random.default_rng(141)
N, mods = 1 << p["logN"], p["Q"] + p["P"]
# (a) one polynomial per launch, forward and inverse, in place too
a = np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in mods])[None]
src = mkrlwe.DeviceLimbs(params, 1, len(mods)).upload(a)
dst = mkrlwe.DeviceLimbs(params, 1, len(mods))
mkrlwe.ntt(params, src, dst)
f = dst.download()
mkrlwe.ntt(params, src, dst, inverse=True)
g = dst.download()
for j in range(len(mods)):
    r, i = (ks.ringQ, j) if j < len(p["Q"]) else (ks.ringP, j - len(p["Q"]))
    assert (f[0][j] == r.ntt(i, a[0][j])).all() and (g[0][j] == r.intt(i, a[0][j])).all(), j
mkrlwe.ntt(params, src, src); mkrlwe.ntt(params, src, src, inverse=True)
assert (src.download() == a).all()
# (b) hoisted forms at level 0 (one digit per component) and at a middle level
for level in (0, 3):
    names = ["p0", "p1", "p2"]
    h = np.empty((4, level + 1, N), dtype=np.uint64)
    for l in range(level + 1):
        h[:, l] = rng.integers(0, p["Q"][l], (4, N), dtype=np.uint64)
    ct = mkckks.NewCiphertext(params, names, level, p["scale"]).upload(h)
    hoisted = mkckks.NewEvaluator(params).HoistedForm(ct)
    beta = ks.beta(level)
    act = list(range(level + 1)) + [len(p["Q"]) + j for j in range(len(p["P"]))]
    for i, n in enumerate(names):
        ref = ks.decompose(level, h[1 + i])
        got = hoisted.Value[n].download()
        assert (got[:beta][:, act] == ref[:beta][:, act]).all(), (level, i)
# (c) a two-party MulRelin at level 2: merged inverse launches of a few limbs
level, k = 2, 2
names = ["u0", "u1"]
def ctx():
    h = np.empty((1 + k, level + 1, N), dtype=np.uint64)
    for l in range(level + 1):
        h[:, l] = rng.integers(0, p["Q"][l], (1 + k, N), dtype=np.uint64)
    return h
def swk():
    out = np.empty((len(p["Q"]), len(mods), N), dtype=np.uint64)
    for j, q in enumerate(mods):
        out[:, j] = rng.integers(0, q, (len(p["Q"]), N), dtype=np.uint64)
    return out
h0, h1 = ctx(), ctx()
u_h = swk()
params.AddCRS(-1, u_h)
rlk = mkrlwe.RelinearizationKeySet(params)
keys = {}
for i, n in enumerate(names):
    keys[i] = (swk(), swk(), swk())
    rlk.AddRelinearizationKey(mkrlwe.RelinearizationKey(params, n, *keys[i]))
ct0 = mkckks.NewCiphertext(params, names, level, p["scale"]).upload(h0)
ct1 = mkckks.NewCiphertext(params, names, level, p["scale"]).upload(h1)
out = mkckks.NewCiphertext(params, names, level, p["scale"] * p["scale"])
mkrlwe.NewKeySwitcher(params).MulAndRelin(ct0, ct1, rlk, out)
_, ref = ks.mul_and_relin(level, [0, 1], h0, [0, 1], h1, keys, u_h)
assert (out.download() == ref).all()
print("forced paths ok")
'''


def test_small_shapes_through_the_h16_kernels():
    env = dict(os.environ, MKHE_NTT16_MIN="1", MKHE_NTT14_MIN="1", MKHE_NTT16_INV_MIN="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(tests=os.path.join(ROOT, "tests"), root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "forced paths ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "Memory access fault" not in r.stderr
