"""Relative error of the matrix-pipe kernel along a regularisation path (per lambda index), for the
scale of the series given in PB_MFMA_YBITS (development aid)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import data, solver
from pybold_amd.hrf_model import spm_hrf
from oracle import c_oracle
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.2744579345
V, L = 64, 20
Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=5.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0,
                                       snr=1.0, seed=11, device=torch.device("cuda"))
lmax = solver.lambda_max(Y, hrf)
grid = torch.logspace(-2, 0, L, dtype=torch.float64, device="cuda")
lam = (lmax[:, None] * grid[None, :]).reshape(-1)
Yh = np.repeat(Y.cpu().numpy().astype(np.float64), L, axis=0)
Wo, _, _ = c_oracle.fista_batch(Yh, hrf, lam.cpu().numpy(), step, 500, threads=8)
for force in ("mfma", "valu"):
    W, _, nd = solver.fista_solve(Y, hrf, lam, step, 500, y_rep=L, force=force)
    W = W.cpu().numpy()
    nrm = np.linalg.norm(Wo, axis=1)
    err = np.where(nrm > 0, np.linalg.norm(W - Wo, axis=1) / np.maximum(nrm, 1e-300), 0.0).reshape(V, L)
    print(force, "ybits", os.environ.get("PB_MFMA_YBITS", "default"), "flagged", int((nd < 0).sum()))
    print("  max err per lambda index:", " ".join("%.1e" % e for e in err.max(axis=0)))
    print("  ||w|| / max|y| median per lambda index:", " ".join("%.0e" % e for e in np.median(nrm.reshape(V, L) / np.abs(Yh).max(axis=1).reshape(V, L), axis=0)))
# risk metric: threshold / largest entry of the solution
W, _, nd = solver.fista_solve(Y, hrf, lam, step, 500, y_rep=L, force="mfma")
W = W.cpu().numpy()
nrm = np.linalg.norm(Wo, axis=1)
err = np.where(nrm > 0, np.linalg.norm(W - Wo, axis=1) / np.maximum(nrm, 1e-300), 0.0)
rho = lam.cpu().numpy() * step / np.maximum(np.abs(Wo).max(axis=1), 1e-300)
rho2 = lam.cpu().numpy() * step * np.sqrt(300) / np.maximum(nrm, 1e-300)
order = np.argsort(rho)
print("err vs rho = th / max|w| (sorted by rho, every 64th):")
for i in order[::64]:
    print("  rho %.3e  rho2 %.3e  err %.2e" % (rho[i], rho2[i], err[i]))
for lo, hi in ((0, .03), (.03, .1), (.1, .3), (.3, 1), (1, 3), (3, 10), (10, 1e9)):
    m = (rho >= lo) & (rho < hi) & (nrm > 0)
    if m.any():
        print("  rho in [%g, %g): n=%4d  max err %.2e  max err/rho %.2e" % (lo, hi, m.sum(), err[m].max(), (err[m] / rho[m]).max()))
