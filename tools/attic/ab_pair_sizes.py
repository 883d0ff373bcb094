import sys
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
for V in (16384, 20000, 32768, 40000, 50000, 65536, 80000, 98304, 100000, 131072, 200000, 400000, 1000000):
    Y = torch.randn(V, 300, device="cuda", dtype=torch.float32)
    out = []
    for force in ("fast1", "fast"):
        plan = solver.FistaPlan(Y, hrf, 1.0, step, 500, force=force)
        plan.run(); torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            plan.W.zero_(); e0.record(); plan.launch(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        out.append(min(ts))
        del plan
    print("V=%7d  single-row %.3f ms (%.3e/s)   pair %.3f ms (%.3e/s)   pair/single %.3f" % (
        V, out[0], V * 500 / out[0] * 1e3, out[1], V * 500 / out[1] * 1e3, out[1] / out[0]), flush=True)
