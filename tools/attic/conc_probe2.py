import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
for P in (10000, 12288, 12500):
    Y = torch.randn(P, 300, device="cuda", dtype=torch.float32)
    for force in ("seq", None, "seq", None):
        plan = solver.FistaPlan(Y, hrf, 1.0, step, 500, force=force)
        for _ in range(3): plan.run()
        torch.cuda.synchronize()
        # (a) synchronised per step
        ts = []
        for _ in range(9):
            torch.cuda.synchronize(); t0 = time.perf_counter(); plan.run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        a = np.median(ts) * 1e3
        # (b) back to back
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): plan.run()
        torch.cuda.synchronize(); b = (time.perf_counter() - t0) / 20 * 1e3
        print("P=%6d %-5s per-step sync %.3f ms   back-to-back %.3f ms" % (P, force or "auto", a, b), flush=True)
