"""Graph-capture check of the encrypted CNN, run as a script in a fresh interpreter by tests/test_gpu_cnn.py (no torch in
the process: the system ROCm HIP runtime is the one loaded)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import harness_cnn as HC          # noqa: E402

TWO = dict(image="dataOwner", kernels="modelOwner", fc1="modelOwner", fc2="modelOwner")


def main():
    from mkhe_kklss_amd import cnn, mkrlwe
    sc = HC.CnnScenario(TWO, seed=6)
    model = HC.synthetic_model(10)
    cts = sc.encrypt_model(model)
    pt, pt_scale = sc.mask_plaintext(sc.level - 4)
    pt = mkrlwe.DeviceLimbs(sc.params, 1, sc.level - 3).upload(pt[None])
    args = (sc.rlkSet, sc.rtkSet, cts["ctImage"], cts["ctKernels"], cts["ctFC1"], cts["ctFC2"], cts["ctB1"], cts["ctB2"], pt, pt_scale)
    forks = [sc.eval.Fork() for _ in range(7)]
    hoisted = (sc.eval.HoistedForm(cts["ctImage"]), [sc.eval.HoistedForm(c) for c in cts["ctKernels"]], [sc.eval.HoistedForm(c) for c in cts["ctFC1"]])
    ref = cnn.Inference(sc.eval, *args, hoisted=hoisted, forks=forks).download()          # eager (also warms every pool)
    with sc.params.Capture() as graph:
        out = cnn.Inference(sc.eval, *args, hoisted=hoisted, forks=forks)
    for _ in range(3):
        graph.launch()
        assert (out.download() == ref).all()
    # a second image through the same handles: the image's hoisted form is an input too (Decompose outside the graph)
    model2 = dict(model, image=HC.synthetic_model(11)["image"])
    fresh = sc.encrypt(HC.pack_image(model2), "dataOwner")
    cts["ctImage"].upload(fresh.download())
    for id in cts["ctImage"].ids:
        sc.eval.ksw.Decompose(cts["ctImage"].Level(), cts["ctImage"], id, hoisted[0].Value[id])
    graph.launch()
    got = sc.decrypt(out)[:HC.NCLS]
    assert np.abs(got.real - HC.plain_forward(model2)).max() < 1e-3
    print("graph replay ok")


if __name__ == "__main__":
    main()
