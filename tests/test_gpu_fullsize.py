"""Full-size parity (-m gpu): BASELINE.json configs[3] shape on one GPU -- PN16QP1761 (N = 2^16, 34 + 4 primes,
alpha = 2, beta = 17; mkrlwe_test.go:22-35), 2-party MulAndRelin and hoisted Rotate, device vs oracle bit for bit.
(configs[1] / configs[2] at full size are checked by bench.py's cpu_baseline leg on every run.)
"""
import numpy as np
import pytest

import harness as H
from gpu_common import Pair, oracle_mul_and_relin

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pair():
    return Pair(H.PN16QP1761, seed=16)


def test_pn16_mul_and_relin_two_parties(pair):
    mk = pair.mk
    assert pair.params.Alpha() == 2 and pair.params.Beta(pair.maxlevel) == 17
    level = pair.maxlevel
    names = ["a", "b"]
    h0, d0 = pair.ct(names, level)
    h1, d1 = pair.ct(names, level)
    rlk_h, rlk_d = pair.rlk_set(names)
    u_h = H.uniform_swk(pair.rng, pair.ks)
    pair.params.AddCRS(-1, u_h)
    out = mk.NewCiphertext(pair.params, names, level)
    pair.ksw.MulAndRelin(d0, d1, rlk_d, out)
    _, ref = oracle_mul_and_relin(pair, level, names, h0, names, h1, rlk_h, u_h, names)
    got = out.download()
    assert (got == ref).all()
    for l, q in enumerate(pair.Q):
        assert (got[:, l] < q).all()


def test_pn16_rotate_hoisted(pair):
    mk = pair.mk
    level = pair.maxlevel
    names = ["a"]
    h, d = pair.ct(names, level)
    crs_h = H.uniform_swk(pair.rng, pair.ks)
    pair.params.AddCRS(5, crs_h)
    k = H.uniform_swk(pair.rng, pair.ks)
    rkset = mk.RotationKeySet()
    rkset.AddRotationKey(mk.RotationKey(pair.params, 5, "a", k))
    hh = mk.NewHoistedCiphertext()
    hh.Value["a"] = mk.NewSwitchingKey(pair.params)
    pair.ksw.Decompose(level, d, "a", hh.Value["a"])
    out = mk.NewCiphertext(pair.params, names, level)
    pair.ksw.RotateHoisted(d, 5, hh, rkset, out)
    ref = pair.ks.rotate(level, pow(5, 5, 2 * pair.N), [0], h, [k], crs_h)
    assert (out.download() == ref).all()


@pytest.mark.parametrize("drop", [1, 31])
def test_pn16_rotate_hoisted_lower_levels(pair, drop):
    """levels whose last gadget digit has ONE limb (33 and 3 limbs, alpha = 2: the copy path of DecomposeAndSplit, basis_extension.go:443-451);
    at 33 limbs the digits go through the radix-4 spread + quarter sub-transforms, at 3 limbs through the small-launch path"""
    mk = pair.mk
    level = pair.maxlevel - drop
    assert (level + 1) % 2 == 1
    names = ["a"]
    h, d = pair.ct(names, level)
    crs_h = H.uniform_swk(pair.rng, pair.ks)
    pair.params.AddCRS(7, crs_h)
    k = H.uniform_swk(pair.rng, pair.ks)
    rkset = mk.RotationKeySet()
    rkset.AddRotationKey(mk.RotationKey(pair.params, 7, "a", k))
    hh = mk.NewHoistedCiphertext()
    hh.Value["a"] = mk.NewSwitchingKey(pair.params)
    pair.ksw.Decompose(level, d, "a", hh.Value["a"])
    out = mk.NewCiphertext(pair.params, names, level)
    pair.ksw.RotateHoisted(d, 7, hh, rkset, out)
    ref = pair.ks.rotate(level, pow(5, 7, 2 * pair.N), [0], h, [k], crs_h)
    assert (out.download() == ref).all()
