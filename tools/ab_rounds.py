"""One stream vs whole rounds + concurrent closing group, by problem count (development aid)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27

def ms(V, force, reps=9, nit=500):
    Y = torch.randn(V, 300, device="cuda", dtype=torch.float32)
    plan = solver.FistaPlan(Y, hrf, 1.0, step, nit, force=force)
    for _ in range(3): plan.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); plan.run(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3

for P in (4500, 5000, 6144, 9000, 10000, 12288, 12500, 13312, 14000, 14336, 20000, 21000, 24576, 25000, 29000, 41000, 50000, 58000, 75000, 100000):
    a, b = ms(P, "seq"), ms(P, None)
    print("P=%7d  one stream %7.3f ms  auto %7.3f ms  (%+.1f %%)   plan %s" % (P, a, b, (b / a - 1) * 100, solver.launch_plan(300, 30, P)), flush=True)
