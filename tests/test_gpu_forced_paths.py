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

# The thresholds and kernel selectors below are A/B instrumentation: they exist only in the -DMKHE_SWITCHES build of the same sources
# (csrc/switches.h, `make switches`); the product library does not read them.
SWITCHES_LIB = os.path.join(ROOT, "mkhe-kklss_amd", "lib", "libmkhe_hip_switches.so")

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
    env = dict(os.environ, MKHE_LIB=SWITCHES_LIB, MKHE_NTT16_MIN="1", MKHE_NTT14_MIN="1", MKHE_NTT16_INV_MIN="1")
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(tests=os.path.join(ROOT, "tests"), root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "forced paths ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "Memory access fault" not in r.stderr


# ---------------------------------------------------------------- round 4: 59/60-bit moduli behind producers whose outputs are not below 2^60
PRIMES = r'''
def is_prime(n):
    if n < 2: return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n %% p == 0: return n == p
    d, s = n - 1, 0
    while d %% 2 == 0: d //= 2; s += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1): continue
        for _ in range(s - 1):
            x = x * x %% n
            if x == n - 1: break
        else: return False
    return True
def primes_below(top, step, count, avoid=()):
    """`count` primes = 1 mod `step`, descending from `top`"""
    out, c = [], (top - 1) // step * step + 1
    while len(out) < count:
        if is_prime(c) and c not in avoid: out.append(c)
        c -= step
    return out
'''

SCRIPT_N16 = PRIMES + r'''
import sys
import numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import harness as H
from oracle import oracle as O
from mkhe_kklss_amd import mkckks, mkrlwe
N = 1 << 16
q60 = 0xfffffffff6a0001                                   # the head prime of mkckks.PN15QP880: = 1 mod 2^17
assert (q60 - 1) %% (2 * N) == 0
p59 = primes_below(1 << 59, 2 * N, 2)
rng = np.random.default_rng(160)
for name, Q, P in (("alpha = 1", [q60, 0x3fffffffd60001, 0x3fffffffca0001, 0x3fffffff360001], p59),
                   ("alpha = 2", [q60] + H.PN16_Q[1:5], H.PN16_P)):
    ks = O.KeySwitcher(16, Q, P, 2)
    params = mkckks.Parameters(16, Q, P, float(1 << 45), device=0)
    mods = Q + P
    # (a) plain forward transforms of a few polynomials over every modulus (split launches: the halves see values in [0, 4q))
    a = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in mods]) for _ in range(3)])
    a[0, :, :4] = np.array([[0, 1, q - 1, q // 2] for q in mods], dtype=np.uint64)
    src = mkrlwe.DeviceLimbs(params, 3, len(mods)).upload(a)
    dst = mkrlwe.DeviceLimbs(params, 3, len(mods))
    mkrlwe.ntt(params, src, dst)
    f = dst.download()
    for c in range(3):
        for j in range(len(mods)):
            r, i = (ks.ringQ, j) if j < len(Q) else (ks.ringP, j - len(Q))
            assert (f[c][j] == r.ntt(i, a[c][j])).all(), (name, c, j)
    # (b) hoisted forms at the top level and one below (Decompose-fused launches; alpha = 2: the digit spread in front of the halves)
    for level in (len(Q) - 1, len(Q) - 2):
        names = ["p0", "p1"]
        h = np.empty((3, level + 1, N), dtype=np.uint64)
        for l in range(level + 1):
            h[:, l] = rng.integers(0, Q[l], (3, N), dtype=np.uint64)
        ct = mkckks.NewCiphertext(params, names, level, float(1 << 45)).upload(h)
        hoisted = mkckks.NewEvaluator(params).HoistedForm(ct)
        beta = ks.beta(level)
        act = list(range(level + 1)) + [len(Q) + j for j in range(len(P))]
        for i, n in enumerate(names):
            ref = ks.decompose(level, h[1 + i])
            got = hoisted.Value[n].download()
            assert (got[:beta][:, act] == ref[:beta][:, act]).all(), (name, level, i)
    params.close()
print("n16 big-modulus paths ok")
'''

SCRIPT_BFV = PRIMES + r'''
import sys
import numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import harness as H
import harness_bfv as HB
from mkhe_kklss_amd import mkbfv, mkrlwe
from mkhe_kklss_amd._abi import check, lib
logN = 15
q60 = primes_below(1 << 60, 1 << 16, 3, avoid=(0xffffffffffc0001, 0xfffffffff840001))
pset = dict(logN=logN, Q=[q60[0], 0x3fffffffd60001, q60[2]], QMul=[q60[1], 0x3fffffffca0001, 0x3fffffff5d0001], P=HB.BFV_PN15QP880["P"], T=65537)
bfv = HB.make_bfv(pset)
params = mkbfv.Parameters(pset["logN"], pset["Q"], pset["QMul"], pset["P"], pset["T"])
ev = mkbfv.NewEvaluator(params)
rng = np.random.default_rng(61)
N, nq = 1 << logN, len(pset["Q"])
# ring-R forward transforms of lazy ModUpQtoR / Rescale outputs (< 3q under the 60-bit primes), then the whole MulRelinNew
x = np.stack([H.uniform_poly(rng, pset["Q"], N) for _ in range(6)])
src = mkrlwe.DeviceLimbs(params, 6, nq).upload(x)
for conv, ora in ((ev.conv.ModUpQtoR, bfv.modup_q_to_r), (ev.conv.Rescale, bfv.rescale)):
    dst = mkbfv.PolyR(params, 6)
    conv(src, dst)
    got = dst.download()
    ref = np.stack([ora(x[c]) for c in range(6)])
    assert (got == ref).all()
    check(lib().mkhe_bfv_ntt_r(params.ctx, dst.devptr(), dst.devptr(), 6, 0))
    got = dst.download()
    for c in range(6):
        assert (got[c] == bfv.ntt_r(ref[c])).all(), c
names = ["a", "b"]
h0, h1 = H.uniform_ct(rng, bfv.ks, 2, nq), H.uniform_ct(rng, bfv.ks, 2, nq)
c0, c1 = mkbfv.NewCiphertext(params, names).upload(h0), mkbfv.NewCiphertext(params, names).upload(h1)
rlk_h, rlk_d = {}, mkbfv.NewRelinearizationKeyKeySet(params)
for i, n in enumerate(names):
    ks5 = [H.uniform_swk(rng, bfv.ks) for _ in range(5)]
    rlk_h[i] = tuple(ks5)
    rlk_d.AddRelinearizationKey(mkbfv.RelinearizationKey(params, n, *ks5))
u_h = H.uniform_swk(rng, bfv.ks)
params.CRS[-1] = mkrlwe.SwitchingKey(params, u_h)
_, ref = bfv.mul_relin_new([0, 1], h0, [0, 1], h1, rlk_h, u_h)
assert (ev.MulRelinNew(c0, c1, rlk_d).download() == ref).all()
assert (ev.mulRelin(c0, c1, rlk_d).download() == ref).all()
print("bfv big-modulus paths ok")
'''


def _run(script, env_extra):
    env = dict(os.environ, MKHE_LIB=SWITCHES_LIB, **env_extra)
    r = subprocess.run([sys.executable, "-c", script % dict(tests=os.path.join(ROOT, "tests"), root=ROOT)], env=env, capture_output=True, text=True, timeout=1200)
    assert "Memory access fault" not in r.stderr
    return r


@pytest.mark.parametrize("env_extra", [dict(MKHE_NTT16_RADIX4="0", MKHE_SPREAD_RADIX4="0"), dict()], ids=["halves", "default"])
def test_n16_with_a_60_bit_modulus(env_extra):
    """N = 2^16 rings with a 60-bit head prime and 59-bit special primes: the H16 sub-transforms behind the streaming cross-half stage and behind
    the alpha = 2 digit spread start from values in [0, 4q) -- above the 2^60 that the per-modulus reduction schedule assumes"""
    r = _run(SCRIPT_N16, dict(MKHE_NTT16_MIN="1", MKHE_NTT14_MIN="1", MKHE_NTT16_INV_MIN="1", **env_extra))
    assert r.returncode == 0 and "n16 big-modulus paths ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("env_extra", [dict(MKHE_NTT32="0"), dict(MKHE_NTT32="1", MKHE_NTT32_MIN="1")], ids=["h16", "h32"])
def test_bfv_with_60_bit_moduli(env_extra):
    """mkbfv over 60-bit Q / QMul primes at N = 2^15: the ring-R forward transforms take lazy ModUpQtoR / Rescale outputs (< 3q = 2^61.6), which the
    H16 / H32 kernels must reduce at the load (NttBatch::src_lazy, set by Context::ntt_r since round 4)"""
    r = _run(SCRIPT_BFV, dict(MKHE_NTT16_MIN="1", MKHE_NTT16_INV_MIN="1", **env_extra))
    assert r.returncode == 0 and "bfv big-modulus paths ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


# ---------------------------------------------------------------- round 4: the single-pass forward kernel (not the default) on every N = 2^15 launch shape
SCRIPT_H32 = r'''
import sys
import numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import harness as H
from oracle import oracle as O
from mkhe_kklss_amd import mkckks, mkrlwe
from mkhe_kklss_amd._abi import lib
p = H.PN15QP880
ks = O.KeySwitcher(p["logN"], p["Q"], p["P"], 2)
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
rng = np.random.default_rng(321)
N, mods = 1 << p["logN"], p["Q"] + p["P"]
# (a) plain forward transforms, one polynomial and three, out of place and in place, edge values in front
for cnt in (1, 3):
    a = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in mods]) for _ in range(cnt)])
    a[0, :, :4] = np.array([[0, 1, q - 1, q // 2] for q in mods], dtype=np.uint64)
    src = mkrlwe.DeviceLimbs(params, cnt, len(mods)).upload(a)
    dst = mkrlwe.DeviceLimbs(params, cnt, len(mods))
    mkrlwe.ntt(params, src, dst)
    f = dst.download()
    for c in range(cnt):
        for j in range(len(mods)):
            r, i = (ks.ringQ, j) if j < len(p["Q"]) else (ks.ringP, j - len(p["Q"]))
            assert (f[c][j] == r.ntt(i, a[c][j])).all(), (cnt, c, j)
    mkrlwe.ntt(params, src, src)
    assert (src.download() == f).all()
# (b) hoisted forms (the Decompose-fused launches): top level, a middle level, level 0
ev = mkckks.NewEvaluator(params)
for level in (len(p["Q"]) - 1, 5, 0):
    names = ["p0", "p1"]
    h = np.empty((3, level + 1, N), dtype=np.uint64)
    for l in range(level + 1):
        h[:, l] = rng.integers(0, p["Q"][l], (3, N), dtype=np.uint64)
    ct = mkckks.NewCiphertext(params, names, level, p["scale"]).upload(h)
    beta = ks.beta(level)
    act = list(range(level + 1)) + [len(p["Q"]) + j for j in range(len(p["P"]))]
    refs = [ks.decompose(level, h[1 + i]) for i in range(len(names))]
    for rep in range(%(reps)d):                       # (MKHE_NTT32=2: 64 launches on H32, a block of 48 on H32, a block of 48 on H16, then the choice -- a sample of every phase is checked)
        hoisted = ev.HoistedForm(ct)
        if rep not in (0, 1, 63, 64, 80, 111, 112, 128, 159, 160, 169):
            continue
        for i, n in enumerate(names):
            got = hoisted.Value[n].download()
            assert (got[:beta][:, act] == refs[i][:beta][:, act]).all(), (level, rep, i)
top = len(p["Q"]) - 1
per = ks.beta(top) * len(mods)
print("choice", lib().mkhe_ntt_choice(params.ctx, 2 * per, 1))
print("h32 paths ok")
'''


@pytest.mark.parametrize("mode,reps", [("1", 1), ("2", 170), ("0", 1)], ids=["forced", "measured", "two_pass"])
def test_single_pass_forward_kernel_on_every_launch_shape(mode, reps):
    """ntt32_fwd_kernel (MKHE_NTT32=1) with its size threshold at 1: plain transforms of one
    and three polynomials and Decompose launches at three levels against the oracle; MKHE_NTT32=2 (the default): per shape the
    engine times a block of launches of each kernel and settles -- a sample of the launches of every phase gives the oracle's digits"""
    r = _run(SCRIPT_H32 % dict(tests="%(tests)s", root="%(root)s", reps=reps), dict(MKHE_NTT32=mode, MKHE_NTT32_MIN="1", MKHE_NTT16="1", MKHE_NTT16_MIN="1"))        # (MKHE_NTT16=1: the measured choice needs both kernels, whatever the caller's switch set says)
    assert r.returncode == 0 and "h32 paths ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    choice = int(r.stdout.split("choice")[1].split()[0])
    assert (choice in (0, 1)) if mode == "2" else choice == -1, r.stdout[-300:]


# Round 6: step F2 inside the Decompose NTT of the t_i (csrc/ntt16_f2_kernels.hip).  The shipped schedule cuts every group of digits evenly (two parts at
# four parties); the weighted schedule (MKHE_F2_BALANCE=5: percent of a pass per reduction point) cuts by the cost of the modulus classes -- runs of unequal length, three parts, groups whose
# last run zeroes the part they do not have -- MKHE_F2_FUSED=0 is the unfused launch set of round 5, 2 the first
# version's rule (one workgroup per CU or no fused launch; 1, the default, plans the grid: f2_plan_schedule): the same bits from all of them.
SCRIPT_F2 = r'''
import sys
import numpy as np
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
import harness as H
from oracle import oracle as O
from mkhe_kklss_amd import mkckks, mkrlwe
p = H.PN15QP880
ks = O.KeySwitcher(p["logN"], p["Q"], p["P"], 2)
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
rng = np.random.default_rng(1506)
N, mods = 1 << p["logN"], p["Q"] + p["P"]
def swk():
    out = np.empty((len(p["Q"]), len(mods), N), dtype=np.uint64)
    for j, q in enumerate(mods):
        out[:, j] = rng.integers(0, q, (len(p["Q"]), N), dtype=np.uint64)
    return out
K = 5
keys = {i: (swk(), swk(), swk()) for i in range(K)}
u_h = swk()
params.AddCRS(-1, u_h)
rlk = mkrlwe.RelinearizationKeySet(params)
for i in range(K):
    rlk.AddRelinearizationKey(mkrlwe.RelinearizationKey(params, "u%%d" %% i, *keys[i]))
ev = mkckks.NewEvaluator(params)
for k, level in ((4, 13), (5, 13), (4, 6), (2, 11), (1, 8)):
    names = ["u%%d" %% i for i in range(k)]
    def ct():
        h = np.empty((1 + k, level + 1, N), dtype=np.uint64)
        for l in range(level + 1):
            h[:, l] = rng.integers(0, p["Q"][l], (1 + k, N), dtype=np.uint64)
        return h
    h0, h1 = ct(), ct()
    ct0 = mkckks.NewCiphertext(params, names, level, p["scale"]).upload(h0)
    ct1 = mkckks.NewCiphertext(params, names, level, p["scale"]).upload(h1)
    got = ev.MulRelinNew(ct0, ct1, rlk).download()
    ids = list(range(k))
    _, ref = ks.mul_and_relin(level, ids, h0, ids, h1, {i: keys[i] for i in ids}, u_h)
    ref = np.stack([ks.ringQ.div_round_last_many(ref[s], 1)[0] for s in range(1 + k)])
    assert got.shape == ref.shape and (got == ref).all(), (k, level)
    assert (ev.MulRelinNew(ct0, ct1, rlk).download() == ref).all(), (k, level)
print("ok")
'''


@pytest.mark.parametrize("env_extra", [dict(MKHE_F2_BALANCE="5"), dict(MKHE_F2_BALANCE="2"), dict(MKHE_F2_FUSED="0"), dict(MKHE_F2_FUSED="2"),
                                       dict(MKHE_FUSE_E="0"), dict(MKHE_FUSE_Y="0", MKHE_FUSE_RESCALE="0")],
                         ids=["weighted_cuts", "weighted_cuts_light", "unfused", "whole_chip_or_nothing", "step_E_as_plain_items", "x_y_E_as_plain_launches"])
def test_fused_f2_schedules(env_extra):
    r = _run(SCRIPT_F2, env_extra)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
