"""Fixed cost of a matrix-pipe launch: one round (16 384 problems) with 0, 1, 10, 100, 500 iterations,
plain (taps as kernel arguments) and shared-HRF form (taps from device memory), cold and warm start.
Usage (GPU): python tools/r3_mfma_overhead.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pybold_amd import solver  # noqa: E402
from pybold_amd.hrf_model import spm_hrf  # noqa: E402
from pybold_amd.utils import gram_frobenius  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    n, V = 300, 16384
    hrf = spm_hrf(1.0, 1.0, 30.0, False)[0]
    step = 1.0 / gram_frobenius(hrf, n)
    Y = torch.randn((V, n), device=dev, dtype=torch.float32)
    W = torch.zeros((V, n), device=dev, dtype=torch.float64)
    taps = torch.from_numpy(np.ascontiguousarray(hrf)).to(dev)
    stepd = torch.tensor([step], dtype=torch.float64, device=dev)
    print("one round = %d problems x %d scans; ms per launch (incl. the 6 us re-solve launch)" % (V, n))
    for it in (0, 1, 10, 100, 500):
        cold = solver.FistaPlan(Y, hrf, 1.0, step, it, force="mfma")
        t_cold = timeit(lambda: cold.launch(cold=True))
        t_warm = timeit(lambda: cold.launch(cold=False))
        t_pp = timeit(lambda: solver.fista_solve_pp(Y, taps, stepd, 1.0, it, W0=W, inplace=True))
        print("n_iter %3d: plain cold %.4f  plain warm %.4f  shared-HRF (device taps, warm, in place) %.4f" % (it, t_cold, t_warm, t_pp), flush=True)


if __name__ == "__main__":
    main()
