"""The Go-dump fixture format (tools/fixture.py) and the replay tool (tests/replay_fixture.py): round trip, and a
replay of a fixture whose "reference output" was written by the oracle itself (the Go writer cannot run here)."""
import numpy as np
import pytest

import harness as H
from oracle import oracle as O

import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import fixture as FX  # noqa: E402
import replay_fixture as RP  # noqa: E402


def _make(tmp_path, pset, corrupt=False):
    ks = O.KeySwitcher(pset["logN"], pset["Q"], pset["P"], 2)
    rng = np.random.default_rng(5)
    names = ["alice", "bob"]
    level = len(pset["Q"]) - 1
    arrays = {"crs_u": H.uniform_swk(rng, ks)}
    rl = {}
    for k, n in enumerate(names):
        b, d, v = (H.uniform_swk(rng, ks) for _ in range(3))
        arrays["rlk/%s/b" % n], arrays["rlk/%s/d" % n], arrays["rlk/%s/v" % n] = b, d, v
        rl[k] = (b, d, v)
    op0, op1 = H.uniform_ct(rng, ks, 2, level + 1), H.uniform_ct(rng, ks, 2, level + 1)
    for pre, ct in (("op0", op0), ("op1", op1)):
        arrays[pre + "/0"] = ct[0]
        for k, n in enumerate(names):
            arrays["%s/%s" % (pre, n)] = ct[1 + k]
    _, out = ks.mul_and_relin(level, [0, 1], op0, [0, 1], op1, rl, arrays["crs_u"])
    if corrupt:
        out = out.copy()
        out[1, 0, 3] ^= np.uint64(1)
    arrays["out/0"] = out[0]
    for k, n in enumerate(names):
        arrays["out/%s" % n] = out[1 + k]
    meta = dict(op="mkrlwe.MulAndRelin", logN=pset["logN"], Q=pset["Q"], P=pset["P"], gamma=2,
                psiQ=[ks.ringQ.psi(i) for i in range(len(pset["Q"]))], psiP=[ks.ringP.psi(i) for i in range(len(pset["P"]))],
                level=level, ids0=names, ids1=names)
    path = str(tmp_path / "mr.fix")
    FX.write(path, meta, arrays)
    return path, meta, arrays


def test_roundtrip(tmp_path):
    path, meta, arrays = _make(tmp_path, H.small_ckks(10, 2))
    m2, a2 = FX.read(path)
    assert m2 == meta and set(a2) == set(arrays)
    for k in arrays:
        assert a2[k].dtype == np.uint64 and (a2[k] == arrays[k]).all()
    with open(path, "r+b") as f:
        f.truncate(os.path.getsize(path) - 8)
    with pytest.raises(ValueError, match="truncated"):
        FX.read(path)


def test_replay_on_oracle(tmp_path):
    path, _, _ = _make(tmp_path, H.small_ckks(10, 3))
    assert RP.replay(path, gpu=False) == {"oracle": True}
    bad, _, _ = _make(tmp_path, H.small_ckks(10, 3), corrupt=True)
    assert RP.replay(bad, gpu=False) == {"oracle": False}


@pytest.mark.gpu
def test_replay_on_device(tmp_path):
    path, _, _ = _make(tmp_path, H.small_ckks(12, 3))
    assert RP.replay(path, gpu=True) == {"oracle": True, "device": True}
