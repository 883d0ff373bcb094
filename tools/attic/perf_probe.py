"""Quick timing of pb_fista_solve on synthetic data (development aid)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf

V = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 500
N = int(sys.argv[3]) if len(sys.argv) > 3 else 300
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
Y = torch.randn(V, N, device="cuda", dtype=torch.float32)
step = 1.0 / 723876.27
for force in ("fast",):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        W, _, _ = solver.fista_solve(Y, hrf, 1.0, step, nit, force=force)
        torch.cuda.synchronize(); dt = time.time() - t0
        vi = V * nit / dt
        print("%s V=%d nit=%d N=%d: %.3f ms  %.3e voxel-iter/s  -> %.2f TB/s algorithmic (%.1f%% of 8 TB/s)"
              % (force, V, nit, N, dt * 1e3, vi, vi * 12 * N / 1e12, vi * 12 * N / 8e12 * 100))
