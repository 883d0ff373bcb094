"""Round 5: which series make the 22-bit operators (and the float32 ones) lose digits?  Families whose energy the operator
H = K_h . cumsum barely sees (alternating signs, high-pass noise, fast sinusoids) against coherence
gamma = lambda_max / (max|y| * sum|c|),  c = cumsum(h)  (= ||H^T y||_inf over its upper bound)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import c_oracle, pybold_oracle as orc
from pybold_amd import solver

g = np.load("tests/golden/case1.npz")
hrf, lip = g["hrf"], float(g["lipschitz"])
N, step = 300, 1.0 / lip
t = np.arange(N)
rng = np.random.RandomState(1)
fams = {"golden": g["y"], "white noise": rng.randn(N), "alternating": np.where(t % 2 == 0, 1.0, -1.0)}
for per in (3, 4, 6, 10, 20, 50):
    fams["sinusoid period %d" % per] = np.sin(2 * np.pi * t / per)
hp = rng.randn(N)
fams["high-pass noise (diff)"] = np.diff(np.r_[0.0, hp])
fams["high-pass noise (diff^2)"] = np.diff(np.r_[0.0, 0.0, hp], n=2)
fams["alternating + 1e-3 golden"] = np.where(t % 2 == 0, 1.0, -1.0) + 1e-3 * g["y"]
fams["alternating + 1e-2 golden"] = np.where(t % 2 == 0, 1.0, -1.0) + 1e-2 * g["y"]
fams["alternating + 0.1 golden"] = np.where(t % 2 == 0, 1.0, -1.0) + 0.1 * g["y"]
fams["alternating x envelope"] = np.where(t % 2 == 0, 1.0, -1.0) * (1 + np.sin(2 * np.pi * t / 100))
fams["constant"] = np.ones(N)
names = list(fams)
Y = np.stack([fams[k] for k in names])
Yd = torch.from_numpy(Y.astype(np.float32)).cuda()
Yo = Yd.cpu().numpy().astype(np.float64)
lmax = solver.lambda_max(Yd, hrf).cpu().numpy()
c = np.cumsum(np.r_[hrf, np.zeros(N - len(hrf))])
gamma = lmax / (np.abs(Yo).max(axis=1) * np.abs(c).sum())


def rel(a, b):
    return np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-300)


print("%-28s %9s | %s" % ("family", "gamma", "max rel. error over diff_z, z, x: matrix pipe / float32 vector form, at lambda = 0, 0.05 lambda_max, 1"))
rows = {k: [] for k in names}
for lam in (0.0, 0.05 * lmax, 1.0):
    ref, _, _ = c_oracle.fista_batch(Yo, hrf, lam, step, 500, threads=8)
    xr, zr = orc.fista_outputs(ref, hrf)
    for force in ("mfmaonly", "fast1"):
        W, _, nd = solver.fista_solve(Yd, hrf, lam, step, 500, force=force)
        X, Z = solver.fista_outputs(W, hrf)
        e = np.maximum(rel(W.cpu().numpy(), ref), np.maximum(rel(Z.cpu().numpy(), zr), rel(X.cpu().numpy(), xr)))
        for i, k in enumerate(names):
            rows[k].append("%s%.1e" % ("*" if int(nd[i]) < 0 else " ", e[i]) if np.linalg.norm(ref[i]) > 0 else "   zero ")
for i, k in enumerate(names):
    r = rows[k]
    print("%-28s %9.2e | %s / %s   %s / %s   %s / %s" % (k, gamma[i], r[0], r[1], r[2], r[3], r[4], r[5]))
print("(* = handed back by the matrix-pipe form's own guards: the library re-solves it on the vector form)")
