"""Diagnostic: per-wave phase timeline of ntt16_f2_kernel inside a 4-party PN15QP880 MulRelin (trace build: make -C mkhe-kklss_amd/csrc trace XFLAGS=-DMKHE_F2_SPIPE=0;
MKHE_LIB=.../libmkhe_hip_trace.so python tools/f2_trace.py [parties]).  Where do a pass's cycles go when one workgroup owns the CU?
(The stamps cost registers: with the pipelined source loads the trace build spills inside the digit loop and its U-class passes come out slower than the product
library's -- build the trace library with the two-group loads, -DMKHE_F2_SPIPE=0, whose timeline the stamps do not disturb.)"""
import sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import harness as H
from mkhe_kklss_amd import mkckks, mkrlwe
from mkhe_kklss_amd._abi import lib, check
p = H.PN15QP880
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"], device=0)
N, level = 1 << 15, len(p["Q"]) - 1
rng = np.random.default_rng(0)
names = ["u%d" % i for i in range(k)]
rlk = mkrlwe.RelinearizationKeySet(params)
for i, n in enumerate(names):
    key = mkrlwe.RelinearizationKey(params, n)
    for j in range(3):
        check(lib().mkhe_crs_expand(params.ctx, 7, 1000 + 3 * i + j, key.Value[j].h))
    rlk.AddRelinearizationKey(key)
params.AddCRS(-1, seed=7)
def ct():
    h = np.stack([np.stack([rng.integers(0, q, N, dtype=np.uint64) for q in p["Q"]]) for _ in range(1 + k)])
    return mkckks.NewCiphertext(params, names, level, p["scale"]).upload(h)
c0, c1 = ct(), ct()
ev = mkckks.NewEvaluator(params)
for _ in range(30): ev.MulRelinNew(c0, c1, rlk)
params.sync()
NWG, NW, NP = 256, 16, 16
words = NWG * NW * NP * 16
words = max(words, 8 * k * 224 * 16 * 32)         # (the hoisting launches of the trace build stamp the same buffer: room for them)
tr = mkrlwe.DeviceLimbs(params, (words + N - 1) // N, 1)
check(lib().mkhe_ntt_trace(params.ctx, tr.devptr()))
ev.MulRelinNew(c0, c1, rlk)
params.sync()
check(lib().mkhe_ntt_trace(params.ctx, None))
t = tr.download().reshape(-1)[:NWG * NW * NP * 16].reshape(NWG, NW, NP, 16).astype(np.int64)
# passes that exist: stamp 10 > stamp 0 > 0 and modulus index plausible
names_ = ["loads + stage 0", "phase A", "xchg A->B (barriers)", "phase B (+ C twiddle requests)", "xchg B->C", "phase C (+ D requests)", "xchg C->D",
          "phase D (+ first key requests)", "xchg D->E", "products (16 x 2, key ring)"]
ok = (t[..., 10] > t[..., 0]) & (t[..., 0] > 0) & (t[..., 12] < 64) & ((t[..., 10] - t[..., 0]) < 10_000_000)
print("# tools/f2_trace.py: lib/libmkhe_hip_trace.so (make trace XFLAGS=-DMKHE_F2_SPIPE=0: two-group source loads, see the script's header), %d-party MulRelin on PN15QP880" % k)
print("passes traced:", int(ok.sum()), "of", NWG * NW * 7 * (k // 4 if k >= 4 else 1))
sel = t[ok]
d = np.diff(sel[:, :11], axis=1)
tot = sel[:, 10] - sel[:, 0]
print("per-wave cycles per pass: mean %.0f  min %.0f  max %.0f" % (tot.mean(), tot.min(), tot.max()))
for i, n in enumerate(names_):
    print("  %-34s mean %8.0f (%5.1f%%)  min %7.0f  max %8.0f" % (n, d[:, i].mean(), 100 * d[:, i].mean() / tot.mean(), d[:, i].min(), d[:, i].max()))
for cls, label in ((lambda m: (m >= 1) & (m <= 13), "U class"), (lambda m: m == 0, "60-bit head"), (lambda m: m >= 14, "59-bit special")):
    mk = cls(sel[:, 12])
    if mk.any():
        print("  %-14s per-wave cycles per pass mean %.0f" % (label, (sel[mk, 10] - sel[mk, 0]).mean()))
# per workgroup: span from its first stamp to its last, in real time (100 MHz)
rt = t[..., 11]
rt_ok = np.where(ok, rt, 0)
first = np.where(ok, t[..., 0], np.iinfo(np.int64).max)
span_cyc = (np.where(ok, t[..., 10], 0).max(axis=(1, 2)) - first.min(axis=(1, 2)))
print("workgroup span (shader cycles): mean %.0f  min %.0f  max %.0f" % (span_cyc.mean(), span_cyc.min(), span_cyc.max()))
print("real-time end spread over workgroups: %.1f us" % ((rt_ok.max(axis=(1, 2)).max() - rt_ok.max(axis=(1, 2)).min()) / 100.0))
# gap between consecutive passes of one wave (flush / loop overhead / waiting at the next pass's loads is inside stamp 0->1)
# shader clock during the kernel: cycles between the product-phase ends of consecutive passes of a wave over the real time between them (100 MHz)
c10, r11 = t[..., 10], t[..., 11]
okp = ok[..., 1:] & ok[..., :-1]
dc = (c10[..., 1:] - c10[..., :-1])[okp]; dr = (r11[..., 1:] - r11[..., :-1])[okp]
good = dr > 0
print("shader clock inside the kernel: %.3f GHz (median %.3f)" % ((dc[good].sum() / (dr[good].sum() * 10.0)), np.median(dc[good] / (dr[good] * 10.0))))
print("real time per pass: mean %.2f us" % (dr[good].mean() / 100.0))
