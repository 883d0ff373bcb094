import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
V, N, nit = 100000, 300, 500
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
Y = torch.randn(V, N, device="cuda", dtype=torch.float32)
step = 1.0 / 723876.27
taps = torch.from_numpy(np.tile(hrf, (V, 1))).cuda()
steps = torch.full((V,), step, dtype=torch.float64, device="cuda")
def t(fn):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
a = t(lambda: solver.fista_solve(Y, hrf, 1.0, step, nit, force="fast"))
b = t(lambda: solver.fista_solve_pp(Y, taps, steps, 1.0, nit, force="fast"))
print("shared taps: %.2f ms (%.3e/s); per-problem taps: %.2f ms (%.3e/s)" % (a * 1e3, V * nit / a, b * 1e3, V * nit / b))
W1, _, _ = solver.fista_solve(Y[:1000], hrf, 1.0, step, 50, force="fast")
W2, _ = solver.fista_solve_pp(Y[:1000], taps[:1000], steps[:1000], 1.0, 50, force="fast")
print("bitwise equal:", bool(torch.equal(W1, W2)))
