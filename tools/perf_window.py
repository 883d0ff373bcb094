import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
for V in (10000, 100000):
    Y = torch.randn(V, 300, device="cuda", dtype=torch.float32)
    for name, kw in (("fast none", dict(force="fast")), ("fast loops", dict(force="fast", stop="loops", tol=0.0)),
                     ("fast window", dict(force="fast", stop="window", tol=0.0, wind=6)),
                     ("fast window+J", dict(force="fast", stop="window", tol=0.0, wind=6, want_J=True)),
                     ("generic window", dict(force="generic", stop="window", tol=0.0, wind=6))):
        nit = 100 if "generic" in name else 500
        solver.fista_solve(Y, hrf, 1.0, step, nit, **kw); torch.cuda.synchronize()
        t0 = time.perf_counter(); solver.fista_solve(Y, hrf, 1.0, step, nit, **kw); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-16s V=%6d it=%d: %8.2f ms  %.3e voxel-iter/s" % (name, V, nit, dt * 1e3, V * nit / dt), flush=True)
