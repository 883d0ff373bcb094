"""Calibration of the dispatch model of pb_fista_solve (capi.hip: COST_FAST1, COST_WIDE,
COST_PARTIAL): time of a 500-iteration plain solve per kernel form and problem count, in ms
and in units of one full pair round (16 384 problems)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf

hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27


def ms(V, force, reps=7, nit=500):
    Y = torch.randn(V, 300, device="cuda", dtype=torch.float32)
    plan = solver.FistaPlan(Y, hrf, 1.0, step, nit, force=force)
    for _ in range(2):
        plan.run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); plan.run(); torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


T = ms(16384, "fast2")
print("pair round (16384 problems): %.3f ms" % T)
print("%8s %18s %18s %18s %18s %18s %18s" % ("P", "fast1", "pair(ffa)", "pair(direct)", "wide", "auto 1 stream", "auto"))
for P in (1, 64, 256, 512, 1024, 1696, 2048, 3072, 4096, 5000, 6000, 8192, 9000, 10000, 11000, 12288, 12500, 12800, 14000, 16384,
          18000, 20480, 24576, 25000, 32768, 50000, 100000):
    row = "%8d" % P
    for force in ("fast1", "fast2", "fast2d", "wide", "seq", None):
        if force == "wide" and P > 20480:
            row += " %18s" % "-"
            continue
        t = ms(P, force)
        row += "  %7.3f ms (%5.3f)" % (t, t / T)
    print(row + "   " + solver.which_kernel(300, 30, P), flush=True)
