"""Series of 305..608 scans (the reference's shipped demo is 600 scans): split pair form vs the
single-row form, plain / cost trace / window rule (voxel-iterations/s; N = 600 counts double per voxel)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf30 = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
hrf32 = spm_hrf(1.0, t_r=1.0, dur=32.)[0]
step = 1.0 / 2.9e6
def t(Y, hrf, **kw):
    solver.fista_solve(Y, hrf, 1.0, step, 500, **kw); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); solver.fista_solve(Y, hrf, 1.0, step, 500, **kw); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best
for V, N, hrf in ((50000, 600, hrf30), (8192, 600, hrf30), (4096, 600, hrf30), (1024, 600, hrf30), (50000, 400, hrf30),
                  (50000, 600, hrf32), (50000, 320, hrf30)):
    Y = torch.randn(V, N, device="cuda", dtype=torch.float32)
    w = dict(stop="window", tol=1e-6, wind=6, want_J=True)
    for name, kw in (("lib", {}), ("single-row", dict(force="fast1")), ("lib +J", dict(want_J=True)),
                     ("lib window+J", w), ("full window+J", dict(force="nocert", **w))):
        try:
            dt = t(Y, hrf, **kw)
            print("V=%6d N=%3d K=%2d %-14s %9.3f ms  %.3e voxel-iter/s  [%s]" %
                  (V, N, len(hrf), name, dt * 1e3, V * 500 / dt, solver.which_kernel(N, len(hrf), V, stop=kw.get("stop"))), flush=True)
        except Exception as e:
            print("V=%6d N=%3d %-14s failed: %s" % (V, N, name, str(e)[:100]), flush=True)
