import sys
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
import pybold_amd
from oracle import c_oracle, pybold_oracle as orc
g = np.load("tests/golden/grid.npz")
hrf, lip = g["hrf"], float(g["lip_s0"])
Y = np.stack([g["y_s%d" % (s % 4)] * (1.0 + 0.05 * (s // 4)) for s in range(16)])
Yd = torch.from_numpy(Y.astype(np.float32)).cuda()
Y32 = Y.astype(np.float32).astype(np.float64)
step = 1.0 / lip
lam = np.linspace(0.05, 0.5, 16)
# cold start, 1000 its, window rule tol=1e-6, per-problem lambda
for W0 in (None, "warm"):
    if W0 == "warm":
        W0t, _, _ = solver.fista_solve(Yd, hrf, lam * 2, step, 300, force="generic")
    else:
        W0t = None
    outs = {}
    for force in ("generic", "fast1", "wide", None):
        W, J, nd = solver.fista_solve(Yd, hrf, lam, step, 1000, W0=W0t, stop="window", tol=1e-6, wind=6, force=force)
        outs[force] = (W.cpu().numpy(), nd.cpu().numpy())
    ref = outs["generic"]
    for force in ("fast1", "wide", None):
        e = np.linalg.norm(outs[force][0] - ref[0], axis=1) / np.linalg.norm(ref[0], axis=1)
        print(W0, force, "n_done", outs[force][1][:8], "ref", ref[1][:8], "max rel err %.2e" % e.max())
for nb_iter in (3, 20, 100):
    sigma32 = np.array([orc.mad_daub_noise_est(y) for y in Y32])
    Wo, Jo, _, _, n_outer = c_oracle.deconv_auto_lbda_batch(Y32, hrf, sigma32, lip, nb_iter=nb_iter, threads=16)
    np.random.seed(0)
    X, Z, W, Jb, Rb, Gb = pybold_amd.deconv(Y, 1.0, hrf, lbda=None, nb_iter=nb_iter)
    e = np.linalg.norm(W - Wo, axis=1) / np.linalg.norm(Wo, axis=1)
    print("auto-lambda nb_iter=%d: max rel err %.2e" % (nb_iter, e.max()), "J err", np.nanmax(np.abs(Jb.T / Jo[:, :Jb.shape[0]] - 1)), "n_outer", n_outer[:8])
    print("   per voxel:", np.array2string(e, precision=1))
