import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
for V in (100000, 10000, 99999):
    Y = torch.randn(V, 300, device="cuda", dtype=torch.float32)
    res = {}
    for rnd in range(2):
        for force in ("fast1", "fast"):
            plan = solver.FistaPlan(Y, hrf, 1.0, step, 500, force=force)
            plan.run(); torch.cuda.synchronize()
            ts = []
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                plan.W.zero_(); e0.record(); plan.launch(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            res[force] = plan.W.clone()
            print("V=%d %-6s round %d: min %.3f ms median %.3f ms -> %.3e voxel-iter/s" % (V, force, rnd, min(ts), np.median(ts), V * 500 / (min(ts) * 1e-3)), flush=True)
    d = (res["fast"] - res["fast1"]).norm(dim=1) / res["fast1"].norm(dim=1)
    print("   pair vs single-row kernels: max rel diff %.2e" % float(d.max()))
