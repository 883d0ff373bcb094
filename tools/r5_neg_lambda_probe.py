"""Probe: negative lambda through the float64 kernels (plain, window rule) against the NumPy oracle."""
import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pybold_oracle as orc
from pybold_amd import solver
g = np.load("tests/golden/case1.npz")
y, hrf, lip = g["y"], g["hrf"], float(g["lipschitz"])
Yd = torch.from_numpy(y[None, :]).cuda()
def rel(a, b): return np.linalg.norm(a - b) / np.linalg.norm(b)
for lb in (0.7, -0.7, -100.0):
    for n in (1, 2, 5, 60, 300):
        ref = orc.fista_batch(y[None, :], hrf, lb, 1.0 / lip, n)
        for force in (None, "generic"):
            W, _, nd = solver.fista_solve(Yd, hrf, lb, 1.0 / lip, n, force=force)
            Ww, _, ndw = solver.fista_solve(Yd, hrf, lb, 1.0 / lip, n, force=force, stop="window", tol=1e-30, wind=6)
            print("lbda %7.1f n %3d %-8s plain %.2e  window(tol=0) %.2e" % (lb, n, force, rel(W.cpu().numpy(), ref), rel(Ww.cpu().numpy(), ref)))
# warm start with negative lambda
W0 = orc.fista_batch(y[None, :], hrf, 0.5, 1.0 / lip, 50)
H = orc._MatrixFreeH(hrf)
for lb in (-0.7, -100.0):
    w = orc._inner_fista(W0[0].copy(), H, H.adj(y), 1.0 / lip, lb / lip, 300, True, 6, 1e-3)
    for force in (None, "generic"):
        Ww, _, ndw = solver.fista_solve(Yd, hrf, lb, 1.0 / lip, 300, W0=torch.from_numpy(W0).cuda(), force=force, stop="window", tol=1e-3, wind=6)
        print("warm lbda %7.1f %-8s window tol 1e-3: n_done %d rel %.2e" % (lb, force, int(ndw[0]), rel(Ww.cpu().numpy()[0], w)))
