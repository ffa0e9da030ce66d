"""The H16 kernels divide job indices by launch constants with ONE scalar multiply-high (csrc/ntt16_kernels.hip udiv_magic / magic_of):
n // d == (n * magic) >> 32 with magic = floor(2^32 / d) + 1, claimed exact for 0 <= n < 2^16, 2 <= d < 2^16; d = 1 has no 32-bit magic and is
encoded as 0 (the quotient is n).  Checked here on the host for every n and a dense sample of d (all small ones, all near powers of two, 2000 random)."""
import numpy as np


def magic_of(d):
    return 0 if d <= 1 else ((1 << 32) // d + 1) & 0xFFFFFFFF


def udiv_magic(n, magic):
    return n if magic == 0 else (n.astype(np.uint64) * np.uint64(magic)) >> np.uint64(32)


def test_udiv_magic_is_exact_below_2_16():
    n = np.arange(1 << 16, dtype=np.uint64)
    rng = np.random.default_rng(7)
    ds = set(range(1, 1025)) | {int(x) for x in rng.integers(2, 1 << 16, 2000)} | {65535, 65534, 32768, 32767, 32769, 4097, 4095}
    for k in range(1, 16):
        ds |= {(1 << k) - 1, 1 << k, (1 << k) + 1}
    for d in sorted(x for x in ds if 1 <= x < (1 << 16)):
        m = magic_of(d)
        assert m < (1 << 32)
        assert (udiv_magic(n, m) == n // np.uint64(d)).all(), d
