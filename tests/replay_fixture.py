#!/usr/bin/env python3
"""Replay a recorded call of the Go reference (tools/fixture.py format, written by shim/go/dump/dump_test.go)
on the CPU oracle and, when a GPU and libmkhe_hip.so are present, on the engine through the C ABI, and
assert bit-equality with the recorded output.  TEST INFRASTRUCTURE (uses the oracle as the checker).

    python tests/replay_fixture.py <fixture> [--no-gpu]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import fixture as FX  # noqa: E402


def _ct(arrays, prefix, ids):
    return np.stack([arrays["%s/0" % prefix]] + [arrays["%s/%s" % (prefix, i)] for i in ids])


def replay_oracle(meta, arrays):
    from oracle import oracle as O
    ks = O.KeySwitcher(meta["logN"], meta["Q"], meta["P"], meta["gamma"], psiQ=meta.get("psiQ"), psiP=meta.get("psiP"))
    names = sorted(set(meta["ids0"]) | set(meta["ids1"]))
    idx = {n: k for k, n in enumerate(names)}
    rl = {idx[n]: (arrays["rlk/%s/b" % n], arrays["rlk/%s/d" % n], arrays["rlk/%s/v" % n]) for n in names}
    ids0, ids1 = sorted(meta["ids0"]), sorted(meta["ids1"])
    ido, out = ks.mul_and_relin(meta["level"], [idx[i] for i in ids0], _ct(arrays, "op0", ids0),
                                [idx[i] for i in ids1], _ct(arrays, "op1", ids1), rl, arrays["crs_u"])
    return [names[i] for i in ido], out


def replay_device(meta, arrays):
    from mkhe_kklss_amd import mkrlwe
    params = mkrlwe.Parameters(meta["logN"], meta["Q"], meta["P"], meta["gamma"], psiQ=meta.get("psiQ"), psiP=meta.get("psiP"))
    names = sorted(set(meta["ids0"]) | set(meta["ids1"]))
    rlk = mkrlwe.RelinearizationKeySet(params)
    for n in names:
        rlk.AddRelinearizationKey(mkrlwe.RelinearizationKey(params, n, arrays["rlk/%s/b" % n], arrays["rlk/%s/d" % n], arrays["rlk/%s/v" % n]))
    params.AddCRS(-1, arrays["crs_u"])
    ids0, ids1 = sorted(meta["ids0"]), sorted(meta["ids1"])
    level = meta["level"]
    c0 = mkrlwe.NewCiphertext(params, ids0, level).upload(_ct(arrays, "op0", ids0))
    c1 = mkrlwe.NewCiphertext(params, ids1, level).upload(_ct(arrays, "op1", ids1))
    out = mkrlwe.NewCiphertext(params, names, level)
    mkrlwe.NewKeySwitcher(params).MulAndRelin(c0, c1, rlk, out)
    return out.ids, out.download()


def expected(meta, arrays):
    names = sorted(set(meta["ids0"]) | set(meta["ids1"]))
    return names, _ct(arrays, "out", names)


def replay(path, gpu=True):
    meta, arrays = FX.read(path)
    if meta.get("op") != "mkrlwe.MulAndRelin":
        raise ValueError("unsupported fixture op %r" % meta.get("op"))
    names, want = expected(meta, arrays)
    res = {}
    ido, got = replay_oracle(meta, arrays)
    res["oracle"] = ido == names and bool((got == want).all())
    if gpu:
        ido, got = replay_device(meta, arrays)
        res["device"] = ido == names and bool((got == want).all())
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("fixture")
    ap.add_argument("--no-gpu", action="store_true")
    a = ap.parse_args()
    r = replay(a.fixture, gpu=not a.no_gpu)
    print(r)
    sys.exit(0 if all(r.values()) else 1)
