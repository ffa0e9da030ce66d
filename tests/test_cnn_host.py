"""Host-side checks of the encrypted-CNN test material (no GPU): the slot packing + circuit of cnn.go on plaintext slot
vectors equals the network itself; the FFT encoder equals the matrix encoder of the harness."""
import numpy as np

import harness as H
import harness_cnn as HC


def test_slot_circuit_equals_network():
    for seed in (1, 2):
        m = HC.synthetic_model(seed)
        a, b = HC.plain_forward(m), HC.slot_forward(m)
        assert np.abs(a - b).max() < 1e-9 * max(1.0, np.abs(a).max())


def test_fast_encoder_matches_matrix_encoder():
    logN = 8
    Q = H.PN15QP880["Q"][:2]
    rng = np.random.default_rng(0)
    z = rng.normal(size=1 << (logN - 1)) + 1j * rng.normal(size=1 << (logN - 1))
    slow, fast = H.CKKSEncoder(logN), HC.FastEncoder(logN)
    scale = float(1 << 40)
    p0, p1 = slow.encode(z, scale, Q), fast.encode(z, scale, Q)
    diff = (p0.astype(np.int64) - p1.astype(np.int64))[0]          # rounding of a half may differ by one unit
    assert np.abs(np.where(np.abs(diff) > (1 << 40), 0, diff)).max() <= 1
    assert np.abs(fast.decode(p1, scale, Q) - z).max() < 1e-9
    assert np.abs(slow.decode(p1, scale, Q) - z).max() < 1e-9
