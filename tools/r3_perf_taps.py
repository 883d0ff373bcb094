"""HRFs of 34..48 taps (short TR) at N = 300: the matrix-pipe form with three near tiles against the
vector dispatch (single-row form: no pair form above 32 taps), and K = 30 for the per-MAC comparison.
Usage (GPU): python tools/r3_perf_taps.py [voxels]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pybold_amd import solver  # noqa: E402
from pybold_amd.hrf_model import spm_hrf  # noqa: E402
from pybold_amd.utils import gram_frobenius  # noqa: E402


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    n, n_iter = 300, 500
    dev = torch.device("cuda:0")
    Y = torch.randn((V, n), device=dev, dtype=torch.float32)
    for k in (30, 33, 40, 48):
        hrf = spm_hrf(1.0, 30.0 / k, 30.0, False)[0][:k]
        step = 1.0 / gram_frobenius(hrf, n)
        for force in (None, "valu"):
            plan = solver.FistaPlan(Y, hrf, 1.0, step, n_iter, force=force)
            for _ in range(3):
                plan.launch(cold=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                plan.launch(cold=True)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            rate = V * n_iter / ms / 1e6
            print("N=%d K=%2d %-5s %8.3f ms  %.3f G voxel-it/s  %.1f T MAC/s (2NK per voxel-iteration)  [%s]"
                  % (n, k, force or "lib", ms, rate, rate * 2 * n * k / 1e3, solver.launch_plan(n, k, V, force=force)[1]))


if __name__ == "__main__":
    main()
