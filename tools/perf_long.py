import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=0.72, dur=20.)[0]
step = 1.0 / 5.0e6
for V, N in ((20000, 1200), (20000, 900), (10000, 2400)):
    Y = torch.randn(V, N, device="cuda", dtype=torch.float32)
    for force, nit in (("fast", 500), ("generic", 50)):
        solver.fista_solve(Y, hrf, 1.0, step, nit, force=force); torch.cuda.synchronize()
        t0 = time.perf_counter(); solver.fista_solve(Y, hrf, 1.0, step, nit, force=force); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("N=%d K=%d V=%d %-8s %s: %.2f ms for %d it -> %.3e voxel-iter/s (%.3e sample-iter/s)" % (
            N, len(hrf), V, force, solver.which_kernel(N, len(hrf), V) if force == "fast" else "generic LDS",
            dt * 1e3, nit, V * nit / dt, V * nit * N / dt), flush=True)
