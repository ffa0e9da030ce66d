"""Test-side pieces of the reference's encrypted-CNN caller (cnn/cnn_test.go): parameter set, slot packing of the
image / kernels / FC matrices / biases, a plaintext model of the network and of the slot-level circuit, and an
O(N log N) CKKS encoder.  The reference's MNIST file is absent and nothing under /root/reference may be
read at run time, so the image is synthetic (seeded); the model is either synthetic or the reference's trained weights (reference_model, a
committed fixture); what is checked is encrypted == plaintext.

Network (cnn_test.go:21-36): 28x28 image -> 5 kernels 4x4, stride 2 -> 13x13x5 -> square -> FC 845 -> 64 (+B1)
-> square -> FC 64 -> 10 (+B2).
"""
import numpy as np

import harness as H

PN14QP433 = dict(  # cnn/cnn_test.go:80-97: 57 + 47 x 6 bit Q, 47 x 2 bit P, scale 2^47
    logN=14,
    Q=[0x2000000002b0001, 0x800000020001, 0x800000280001, 0x800000520001, 0x800000770001, 0x800000aa0001, 0x800000ad0001],
    P=[0x800000df0001, 0x800000f80001], scale=float(1 << 47))
ROTS = [14, 15, 384, 512, 640, 768, 896, 8191, 8190, 8188, 8184]          # genTestParams, cnn_test.go:187

IMG, NK, KS, BLK, CO, NFC, GAP, NCLS, SLOTS = 28, 5, 4, 14, 13, 64, 128, 10, 8192


def synthetic_model(seed):
    rng = np.random.default_rng(seed)
    return dict(image=rng.random((IMG, IMG)), kernels=rng.normal(0, 0.3, (NK, KS, KS)), FC1=rng.normal(0, 0.05, (CO * CO * NK, NFC)),
                FC2=rng.normal(0, 0.1, (NFC, NCLS)), B1=rng.normal(0, 0.1, NFC), B2=rng.normal(0, 0.1, NCLS))


def reference_model(seed):
    """the trained model of the reference's own test (cnn/data/{k1,FC1,FC2,B1,B2}.txt, loaded as cnn_test.go:234-349 does; stored
    as tests/golden/cnn_weights.npz by tests/golden/make_cnn_weights.py) with a seeded synthetic image: the reference's MNIST
    csv (cnn_test.go:65) is not part of its repository."""
    import os
    w = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cnn_weights.npz"))
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:IMG, 0:IMG]
    img = np.clip(np.exp(-((xx - 14 + 3 * np.sin(yy / 4.0)) ** 2) / 8.0) + 0.05 * rng.random((IMG, IMG)), 0, 1)      # a stroke, values in [0, 1]
    return dict(image=img, kernels=w["kernels"], FC1=w["FC1"], FC2=w["FC2"], B1=w["B1"], B2=w["B2"])


def plain_forward(m):
    """the network itself"""
    img, ker = m["image"], m["kernels"]
    conv = np.zeros((NK, CO, CO))
    for a in range(KS):
        for b in range(KS):
            conv += ker[:, a, b][:, None, None] * img[a:a + 2 * CO:2, b:b + 2 * CO:2][None]
    sq = conv ** 2
    vec = np.zeros(CO * CO * NK)
    i, j, k = np.meshgrid(np.arange(NK), np.arange(CO), np.arange(CO), indexing="ij")
    vec[(i + NK * (j * CO + k)).ravel()] = sq.ravel()              # FC1 row index of (kernel i, row j, column k), cnn_test.go:436
    fc1 = (vec @ m["FC1"] + m["B1"]) ** 2
    return fc1 @ m["FC2"] + m["B2"]


# ---- slot packing (restating cnn_test.go:333-543)
def pack_image(m):
    img, v = m["image"], np.zeros(SLOTS)
    for q, (da, db) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):          # the four 14x14 polyphase components, 1024 slots apart
        comp = img[da::2, db::2]                                              # [14][14]
        for k in range(NK):
            v[1024 * q + BLK * BLK * k: 1024 * q + BLK * BLK * (k + 1)] = comp.ravel()
    v[4096:] = v[:4096]
    return v


def pack_kernels(m):
    ker, out = m["kernels"], np.zeros((KS, SLOTS))
    # ciphertext t multiplies the image rotated by (0, 1, 14, 15)[t] = block offset (t // 2, t % 2); component q = (da, db)
    for t in range(4):
        oa, ob = t // 2, t % 2
        for q, (da, db) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
            for k in range(NK):
                blk = np.zeros((BLK, BLK))
                blk[:CO, :CO] = ker[k, 2 * oa + da, 2 * ob + db]
                out[t, 1024 * q + BLK * BLK * k: 1024 * q + BLK * BLK * (k + 1)] = blk.ravel()
    out[:, 4096:] = out[:, :4096]
    return out


def pack_fc1(m):
    tmp = np.zeros((NFC, 1024))
    i, j, k = np.meshgrid(np.arange(NK), np.arange(CO), np.arange(CO), indexing="ij")
    tmp[:, (BLK * BLK * i + BLK * j + k).ravel()] = m["FC1"][(i + NK * (j * CO + k)).ravel(), :].T
    out = np.zeros((8, SLOTS))
    for i in range(8):
        for j in range(NFC):
            s = 128 * ((i + j) % 8)
            out[i, 128 * j:128 * (j + 1)] = tmp[j, s:s + 128]
    return out


def pack_fc2(m):
    v = np.zeros(SLOTS)
    for x in range(NFC):
        v[x * GAP: x * GAP + NCLS] = m["FC2"][x]
    return v


def pack_b1(m):
    v = np.zeros(SLOTS)
    v[:NFC * GAP:GAP] = m["B1"]
    return v


def pack_b2(m):
    v = np.zeros(SLOTS)
    v[:NCLS] = m["B2"]
    return v


def mask():
    v = np.zeros(SLOTS)
    v[::GAP] = 1
    return v


def rot(v, k):
    """mkckks RotateNew: slot i of the result = slot i + k of the input"""
    return np.roll(v, -k)


def slot_forward(m):
    """the circuit of cnn.go on plaintext slot vectors (no encryption): checks the packing"""
    img, ker = pack_image(m), pack_kernels(m)
    conv = img * ker[0] + rot(img, 1) * ker[1] + rot(img, 14) * ker[2] + rot(img, 15) * ker[3]      # cnn.go:16-31
    conv = conv + rot(conv, 2048)
    conv = conv + rot(conv, 1024)                                                                    # :33-37
    sq1 = conv ** 2
    fc1m = pack_fc1(m)
    out = sum(rot(sq1, 128 * i) * fc1m[i] for i in range(8))                                         # :51-60
    for i in range(7):
        out = out + rot(out, 1 << i)                                                                 # :63-67
    sq2 = (out + pack_b1(m)) ** 2
    f = sq2 * mask()                                                                                 # cnn.go:77
    for i in range(4):
        f = f + rot(f, -(1 << i))                                                                    # :80-84
    f = f * pack_fc2(m)
    for i in range(6):
        f = f + rot(f, 128 * (1 << i))                                                               # :88-92
    return (f + pack_b2(m))[:NCLS]


class FastEncoder:
    """CKKS canonical embedding by FFT: slot j <-> evaluation at zeta^(5^j), zeta = exp(i pi / N) (same map as
    harness.CKKSEncoder, O(N log N) instead of an N/2 x N matrix)."""

    def __init__(self, logN):
        self.N = N = 1 << logN
        g = np.array([pow(5, j, 2 * N) for j in range(N // 2)], dtype=np.int64)
        self.t = (g - 1) // 2                      # zeta^(2t+1) = zeta^(5^j)
        self.tc = (2 * N - g - 1) // 2             # the conjugate root
        self.tw = np.exp(1j * np.pi * np.arange(N) / N)

    def encode(self, z, scale, moduli):
        v = np.zeros(self.N, dtype=np.complex128)
        z = np.asarray(z, dtype=np.complex128)
        v[self.t], v[self.tc] = z, np.conj(z)
        mk = np.fft.fft(v) / self.N / self.tw       # v_t = sum_k m_k zeta^k e^(2 pi i k t / N)
        coeffs = [int(x) for x in np.rint(mk.real * scale)]
        return H.int_poly_to_rns(coeffs, moduli)

    def decode(self, poly, scale, moduli):
        c, _ = H.crt_center(poly, moduli)
        mk = np.array([float(x) for x in c]) / scale
        return (np.fft.ifft(mk * self.tw) * self.N)[self.t]


class CnnScenario:
    """keys and CRS made on the device (own random samples), encoder / encryptor / decryptor on the host (test harness);
    owners: dict role -> party id for the roles image, kernels, fc1, fc2 (the reference uses dataOwner for the image
    and modelOwner for everything else, cnn_test.go:34-35,124-129)."""

    def __init__(self, owners, seed=1):
        from mkhe_kklss_amd import mkckks, mkrlwe
        from oracle import oracle as O
        self.mkckks, self.mkrlwe = mkckks, mkrlwe
        p = PN14QP433
        self.owners = owners
        self.params = params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"])
        self.level, self.scale = len(p["Q"]) - 1, p["scale"]
        params.GenDefaultCRS(seed=seed)
        for r in ROTS:
            params.AddCRS(r, seed=seed)
        self.hkg = H.KeyGen(O.KeySwitcher(p["logN"], p["Q"], p["P"], 2), seed)          # host encryptor / decryptor
        self.enc = FastEncoder(p["logN"])
        kgen = mkrlwe.NewKeyGenerator(params, mkrlwe.HostSampler(np.random.default_rng(seed), insecure_test_only=True))
        self.rlkSet, self.rtkSet = mkrlwe.RelinearizationKeySet(params), mkrlwe.RotationKeySet()
        self.sk, self.pk = {}, {}
        rots = ROTS + [1 << i for i in range(p["logN"] - 1)]
        for id in sorted(set(owners.values())):
            sk, pk = kgen.GenKeyPair(id)
            self.rlkSet.AddRelinearizationKey(kgen.GenRelinearizationKey(sk, kgen.GenSecretKey(id)))
            for r in rots:
                self.rtkSet.AddRotationKey(kgen.GenRotationKey(r, sk))
            self.sk[id] = sk.Value.download()[0]
            pkh = pk.Value.download()
            self.pk[id] = (pkh[0], pkh[1])
        self.eval = mkckks.NewEvaluator(params)

    def encrypt(self, slots, id):
        pt = self.enc.encode(slots, self.scale, PN14QP433["Q"])
        c0, c1 = self.hkg.encrypt(pt, self.pk[id], self.level)
        return self.mkckks.NewCiphertext(self.params, [id], self.level, self.scale).upload(np.stack([c0, c1]))

    def decrypt(self, ct):
        host = ct.download()
        vals = {"0": host[0]}
        for i, id in enumerate(ct.ids):
            vals[id] = host[1 + i]
        poly = self.hkg.decrypt(vals, self.sk)
        return self.enc.decode(poly, ct.Scale, PN14QP433["Q"][: ct.Level() + 1])

    def encrypt_model(self, m):
        o = self.owners
        return dict(ctImage=self.encrypt(pack_image(m), o["image"]),
                    ctKernels=[self.encrypt(v, o["kernels"]) for v in pack_kernels(m)],
                    ctFC1=[self.encrypt(v, o["fc1"]) for v in pack_fc1(m)],
                    ctFC2=self.encrypt(pack_fc2(m), o["fc2"]), ctB1=self.encrypt(pack_b1(m), o["fc1"]),
                    ctB2=self.encrypt(pack_b2(m), o["fc2"]))

    def mask_plaintext(self, level):
        """EncodeMsgNew of the 0/1 mask (cnn_test.go:143-149) at `level`"""
        return self.enc.encode(mask(), self.scale, PN14QP433["Q"][: level + 1]), self.scale
