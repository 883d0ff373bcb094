import sys
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
V = 100000
Y = torch.randn(V, 300, device="cuda", dtype=torch.float32)
import time
for rnd in range(2):
    for force in ("fast1", "fast2"):
        solver.fista_solve(Y, hrf, 1.0, step, 500, want_J=True, force=force); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            W, J, _ = solver.fista_solve(Y, hrf, 1.0, step, 500, want_J=True, force=force)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("with J, %s: min %.3f ms -> %.3e voxel-iter/s" % (force, min(ts) * 1e3, V * 500 / min(ts)), flush=True)
