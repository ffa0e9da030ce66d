"""Generator of tests/golden/cnn_weights.npz: the trained MNIST model the reference's own CNN test loads
(/root/reference/cnn/data/{k1,FC1,FC2,B1,B2}.txt, parsed as cnn/cnn_test.go:276-349 does: kernels[i][j / 4][j % 4], FC rows = lines,
biases = one line).  The data files are the reference test's fixtures; only their numeric content is stored (float64).
Run in the build container (the reference tree is not available on the GPU box):  python tests/golden/make_cnn_weights.py"""
import os
import numpy as np

D = "/root/reference/cnn/data"
rows = lambda f: [[float(v) for v in l.split(" ")] for l in open(os.path.join(D, f)).read().split("\n")]
k = np.array(rows("k1.txt")).reshape(5, 4, 4)
FC1 = np.array(rows("FC1.txt"))
FC2 = np.array(rows("FC2.txt"))
B1 = np.array([float(v) for v in open(os.path.join(D, "B1.txt")).read().split(" ")])
B2 = np.array([float(v) for v in open(os.path.join(D, "B2.txt")).read().split(" ")])
assert k.shape == (5, 4, 4) and FC1.shape == (845, 64) and FC2.shape == (64, 10) and B1.shape == (64,) and B2.shape == (10,)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cnn_weights.npz"), kernels=k, FC1=FC1, FC2=FC2, B1=B1, B2=B2)
print("kernels", k.shape, "FC1", FC1.shape, "FC2", FC2.shape, "B1", B1.shape, "B2", B2.shape)
