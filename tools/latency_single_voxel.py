import sys, time, io, contextlib
import numpy as np, torch
sys.path.insert(0, ".")
import pybold_amd
g = np.load("tests/golden/case1.npz")
y, hrf = g["y"], g["hrf"]
for _ in range(3):
    pybold_amd.deconv(y, 1.0, hrf, lbda=1.0, nb_iter=500, early_stopping=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    pybold_amd.deconv(y, 1.0, hrf, lbda=1.0, nb_iter=500, early_stopping=False)
print("single-voxel deconv(500 it): %.2f ms per call (reference: ~90 ms on one core)" % ((time.perf_counter() - t0) / 20 * 1e3))
t0 = time.perf_counter()
for _ in range(20):
    pybold_amd.deconv(y, 1.0, hrf, lbda=1.0)      # defaults: early stopping, 1000 iterations
print("single-voxel deconv(defaults, 1000 it + window rule): %.2f ms per call" % ((time.perf_counter() - t0) / 20 * 1e3))
