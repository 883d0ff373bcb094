"""ms per 500-iteration solve of the window-rule kernel forms by batch size (dispatch calibration)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import data, solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
Yall, _, _ = data.gen_rnd_bloc_bold_batch(100000, dur=5.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0,
                                          snr=1.0, seed=1, device=torch.device("cuda"))
def t(Y, **kw):
    best = 1e9
    solver.fista_solve(Y, hrf, 1.0, step, 500, **kw); torch.cuda.synchronize()
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        W = torch.empty((Y.shape[0], 300), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter(); solver.fista_solve(Y, hrf, 1.0, step, 500, **kw); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
print("%8s %10s %10s %10s %10s %10s %10s" % ("P", "cert2+J", "fast1 w+J", "wide w+J", "lib w+J", "plain+J", "plain"))
for P in (212, 424, 848, 1696, 1808, 2048, 4096, 4308, 8192, 10000, 12500, 16384, 20000, 25000, 50000, 100000):
    Y = Yall[:P].contiguous()
    w = dict(stop="window", tol=1e-6, wind=6, want_J=True)
    row = [t(Y, force="cert2", **w), t(Y, force="fast1", **w), t(Y, force="wide", **w) if P <= 20000 else float("nan"),
           t(Y, **w), t(Y, want_J=True), t(Y)]
    print("%8d " % P + " ".join("%10.3f" % v for v in row), flush=True)
