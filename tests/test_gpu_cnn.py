"""The reference's encrypted CNN (cnn/cnn.go, TestCNN cnn_test.go:100-182) on the device (-m gpu): keys and CRS generated
on the GPU, fresh encryptions from the host harness, the whole circuit (12 MulRelin, 29 rotations, additions, MulPtxt)
on resident ciphertexts, decrypted logits against the plaintext network.  The reference asserts argmax only (:181)."""
import numpy as np
import pytest

import harness_cnn as HC

pytestmark = pytest.mark.gpu

TWO = dict(image="dataOwner", kernels="modelOwner", fc1="modelOwner", fc2="modelOwner")          # cnn_test.go:34-35
FOUR = dict(image="dataOwner", kernels="convOwner", fc1="fc1Owner", fc2="fc2Owner")              # BASELINE.json configs[4]: 4 parties


@pytest.mark.parametrize("weights", ["synthetic", "reference"])
@pytest.mark.parametrize("owners", [TWO, FOUR], ids=["2party", "4party"])
def test_encrypted_cnn_matches_plaintext(owners, weights):
    """weights = reference: the trained model the reference's TestCNN loads (cnn/data/*.txt as a committed fixture)"""
    from mkhe_kklss_amd import cnn
    sc = HC.CnnScenario(owners, seed=3)
    model = HC.synthetic_model(7) if weights == "synthetic" else HC.reference_model(7)
    cts = sc.encrypt_model(model)
    # the mask is multiplied into square2Out, which sits 4 levels below the fresh ciphertexts
    pt, pt_scale = sc.mask_plaintext(sc.level - 4)
    out = cnn.Inference(sc.eval, sc.rlkSet, sc.rtkSet, cts["ctImage"], cts["ctKernels"], cts["ctFC1"], cts["ctFC2"],
                        cts["ctB1"], cts["ctB2"], pt, pt_scale)
    assert sorted(out.ids) == sorted(set(owners.values())) and out.Level() == 0
    got = sc.decrypt(out)[:HC.NCLS]
    ref = HC.plain_forward(model)
    err = np.abs(got.real - ref).max()
    print("logits", np.round(ref, 4), "max abs error", err, "imag", np.abs(got.imag).max())
    assert int(np.argmax(got.real)) == int(np.argmax(ref))
    assert err < 1e-3 * max(1.0, np.abs(ref).max())


def test_layers_one_by_one():
    """each layer function against the slot-level circuit on its own inputs (2 parties)"""
    from mkhe_kklss_amd import cnn
    sc = HC.CnnScenario(TWO, seed=4)
    m = HC.synthetic_model(8)
    ev = sc.eval
    ctImage = sc.encrypt(HC.pack_image(m), "dataOwner")
    ctK = [sc.encrypt(v, "modelOwner") for v in HC.pack_kernels(m)]
    conv = cnn.Convolution(ev, sc.rlkSet, sc.rtkSet, ctImage, ev.HoistedForm(ctImage), ctK, [ev.HoistedForm(c) for c in ctK])
    img, ker = HC.pack_image(m), HC.pack_kernels(m)
    exp = img * ker[0] + HC.rot(img, 1) * ker[1] + HC.rot(img, 14) * ker[2] + HC.rot(img, 15) * ker[3]
    exp = exp + HC.rot(exp, 2048)
    exp = exp + HC.rot(exp, 1024)
    assert conv.Level() == sc.level - 1
    assert np.abs(sc.decrypt(conv) - exp).max() < 1e-6
    # FC2 on a fresh vector: mask, 4 negative rotations, MulRelin, 6 rotations, bias
    vec = np.random.default_rng(1).normal(size=HC.SLOTS)
    ctVec = sc.encrypt(vec, "dataOwner")
    pt, pts = sc.mask_plaintext(sc.level)
    out = cnn.FC2Layer(ev, sc.rlkSet, sc.rtkSet, ctVec, sc.encrypt(HC.pack_fc2(m), "modelOwner"), sc.encrypt(HC.pack_b2(m), "modelOwner"), pt, pts)
    f = vec * HC.mask()
    for i in range(4):
        f = f + HC.rot(f, -(1 << i))
    f = f * HC.pack_fc2(m)
    for i in range(6):
        f = f + HC.rot(f, 128 * (1 << i))
    f = f + HC.pack_b2(m)
    assert np.abs(sc.decrypt(out)[:HC.NCLS] - f[:HC.NCLS]).max() < 1e-5


def test_forked_contexts_give_identical_results():
    """the independent chains of Convolution / FC1Layer issued through forked engine contexts (own streams, shared keys and
    ciphertexts): bit-identical to the single-stream evaluation, repeated to exercise buffer reuse across contexts"""
    from mkhe_kklss_amd import cnn
    sc = HC.CnnScenario(TWO, seed=5)
    model = HC.synthetic_model(9)
    cts = sc.encrypt_model(model)
    pt, pt_scale = sc.mask_plaintext(sc.level - 4)
    args = (sc.rlkSet, sc.rtkSet, cts["ctImage"], cts["ctKernels"], cts["ctFC1"], cts["ctFC2"], cts["ctB1"], cts["ctB2"], pt, pt_scale)
    ref = cnn.Inference(sc.eval, *args).download()
    for nf in (3, 7):
        forks = [sc.eval.Fork() for _ in range(nf)]
        for _ in range(3):
            out = cnn.Inference(sc.eval, *args, forks=forks)
            assert (out.download() == ref).all()
        sc.params.sync()
        for f in forks:
            f.params.sync()


def test_captured_graph_replays_identically():
    """the whole inference captured into a HIP graph (mkhe_capture_*; forks become parallel branches); replays give the
    eager result bit for bit, also after a new image has been uploaded into the same input handle.  Runs in a fresh
    interpreter: a process that has imported torch is bound to torch's bundled ROCm 7.0 HIP runtime, whose
    hipStreamEndCapture cannot end the engine's multi-stream captures (mkhe_capture_begin refuses there, checked below)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "cnn_graph_check.py")], capture_output=True, text=True, timeout=600)
    print(out.stdout[-2000:], out.stderr[-2000:])
    assert out.returncode == 0 and "graph replay ok" in out.stdout


def test_capture_refused_on_old_runtime_or_works():
    """in THIS process: either the runtime supports the capture (no torch imported before) or mkhe_capture_begin raises"""
    from mkhe_kklss_amd import mkckks
    from mkhe_kklss_amd._abi import MkheError
    p = HC.PN14QP433
    params = mkckks.Parameters(p["logN"], p["Q"], p["P"], p["scale"])
    try:
        with params.Capture() as g:
            pass
    except MkheError as e:
        assert "cannot end a multi-stream capture" in str(e)


def test_encrypted_cnn_matches_the_oracle_evaluator():
    """the whole inference, device against the CPU oracle running the same circuit on the same ciphertexts and keys (tests/oracle_evaluator.py:
    the cpu_baseline leg of bench.py --scheme cnn): every output limb bit for bit, same level, same scale"""
    import oracle_evaluator as OE
    from oracle import oracle as O
    from mkhe_kklss_amd import cnn
    p = HC.PN14QP433
    sc = HC.CnnScenario(TWO, seed=6)
    cts = sc.encrypt_model(HC.synthetic_model(10))
    pt, pt_scale = sc.mask_plaintext(sc.level - 4)
    out = cnn.Inference(sc.eval, sc.rlkSet, sc.rtkSet, cts["ctImage"], cts["ctKernels"], cts["ctFC1"], cts["ctFC2"], cts["ctB1"], cts["ctB2"], pt, pt_scale)
    parties = sorted(set(TWO.values()))
    rots = sorted(set(HC.ROTS + [1 << i for i in range(p["logN"] - 1)]))
    rlk_h = {id: tuple(sc.rlkSet.GetRelinearizationKey(id).Value[j].download() for j in range(3)) for id in parties}
    rk_h = {(id, r): sc.rtkSet.GetRotationKey(id, r).Value.download() for id in parties for r in rots}
    crs_h = {r: sc.params.CRS[r].download() for r in rots + [-1] if r in sc.params.CRS}
    O.set_threads(8)
    try:
        oev = OE.OracleEvaluator(O.KeySwitcher(p["logN"], p["Q"], p["P"], 2), p["Q"], p["scale"], rlk_h, rk_h, crs_h, p["logN"])
        H_ = lambda c: OE.OCt(c.ids, c.download(), c.Scale)
        ref = cnn.Inference(oev, None, None, H_(cts["ctImage"]), [H_(c) for c in cts["ctKernels"]], [H_(c) for c in cts["ctFC1"]], H_(cts["ctFC2"]),
                            H_(cts["ctB1"]), H_(cts["ctB2"]), pt, pt_scale)
    finally:
        O.set_threads(1)
    assert out.ids == ref.ids and out.Level() == ref.Level() and out.Scale == ref.Scale
    assert (out.download() == ref.host).all()
