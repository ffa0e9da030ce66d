"""CPU check of the schedule of ntt16_f2_kernel (csrc/engine_mulrelin.hip f2_build_schedule through mkhe_f2_schedule_probe): for every shape the engine
may ask for, every pass (party, limb slot, half limb, digit) is dealt to exactly one workgroup, the runs of a group are numbered 0, 1, .. in digit order,
the group's last run zeroes exactly the parts the group does not have, and no workgroup carries more runs than the kernel walks.  No GPU involved.
Replaces nothing in the reference: the loop order of mkrlwe/keyswitch_hoisted.go:161-178 is free (canonical sums)."""
import ctypes as C
import itertools

import numpy as np
import pytest

from mkhe_kklss_amd._abi import lib

F2_SEGS = 3


def _probe(parties, nb, nslots, weights, grid):
    w = (C.c_long * nslots)(*weights)
    segs = (C.c_ubyte * (grid * F2_SEGS * 8))()
    parts = C.c_int(0)
    nwg = lib().mkhe_f2_schedule_probe(parties, nb, nslots, w, grid, segs, C.byref(parts))
    return nwg, parts.value, np.frombuffer(segs, dtype=np.uint8).reshape(grid, F2_SEGS, 8)


def _check(parties, nb, nslots, weights, grid):
    nwg, parts, segs = _probe(parties, nb, nslots, weights, grid)
    if nwg == 0:
        return 0
    assert 1 <= nwg <= grid and 1 <= parts and 4 * parts + 1 <= 13
    seen = np.zeros((parties, nslots, 2, nb), dtype=np.int32)
    runs = {}
    for wg in range(grid):
        for si in range(F2_SEGS):
            party, slot, half, d0, nd, part, pad0, pad1 = (int(x) for x in segs[wg, si])
            if nd == 0:
                continue
            assert wg < nwg and party < parties and slot < nslots and half < 2 and d0 + nd <= nb and pad1 == 0
            seen[party, slot, half, d0:d0 + nd] += 1
            runs.setdefault((party, slot, half), []).append((d0, nd, part, pad0))
    assert (seen == 1).all(), "a pass dealt %s times" % sorted(set(seen.flatten().tolist()))
    for g, rs in runs.items():
        rs.sort()
        assert [r[2] for r in rs] == list(range(len(rs))), (g, rs)            # parts in digit order
        assert all(r[3] == 0 for r in rs[:-1]) and rs[-1][3] == parts - len(rs), (g, rs, parts)
        assert len(rs) <= parts
    # the weighted load of the heaviest workgroup stays within one pass of the mean (the cut is by the midpoint of a pass's weight interval)
    load = np.zeros(grid)
    for wg in range(grid):
        for si in range(F2_SEGS):
            party, slot, half, d0, nd = (int(x) for x in segs[wg, si][:5])
            load[wg] += nd * weights[slot]
    total = sum(weights[s] for s in range(nslots)) * 2 * parties * nb
    assert load.max() <= total / grid + max(weights) + 1e-9
    return parts


def test_headline_shape_is_two_even_parts():
    # PN15QP880, 4 parties, 14 digits, 16 limb slots, 256 workgroups, every pass the same cost: two runs of seven digits per group
    nwg, parts, segs = _probe(4, 14, 16, [100] * 16, 256)
    assert nwg == 256 and parts == 2
    assert (segs[:, 0, 4] == 7).all() and (segs[:, 1:, 4] == 0).all()
    _check(4, 14, 16, [100] * 16, 256)


@pytest.mark.parametrize("parties", [1, 2, 3, 4, 5, 6, 7, 8, 12, 16])
@pytest.mark.parametrize("level", [0, 1, 5, 9, 13])
def test_every_pass_exactly_once(parties, level):
    nb, nslots = level + 1, level + 1 + 2
    for grid in (256, 304, 64):
        _check(parties, nb, nslots, [100] * nslots, grid)
        w = [100] * nslots
        w[0] = 115
        w[-1] = w[-2] = 105
        _check(parties, nb, nslots, w, grid)


def test_random_shapes():
    rng = np.random.default_rng(62)
    found = 0
    for _ in range(400):
        parties, nb = int(rng.integers(1, 17)), int(rng.integers(1, 40))
        nslots = int(rng.integers(1, 41))
        grid = int(rng.choice([32, 104, 256, 304]))
        weights = [int(x) for x in rng.integers(80, 160, nslots)]
        found += _check(parties, nb, nslots, weights, grid) > 0
    assert found > 50


def test_refusals():
    assert _probe(0, 14, 16, [100] * 16, 256)[0] == 0
    assert _probe(17, 14, 16, [100] * 16, 256)[0] == 0
    assert _probe(4, 14, 16, [0] * 16, 256)[0] == 0
    # one party on 256 workgroups: 448 passes, a group of 14 digits would be cut into eight or more parts -- more than an inverse job adds at its load
    assert _probe(1, 14, 16, [100] * 16, 256)[0] == 0
