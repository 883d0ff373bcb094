import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch
from pybold_amd import data, solver
from oracle import c_oracle
g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "case1.npz"))
hrf, lip = g["hrf"], float(g["lipschitz"])
Y, _, _ = data.gen_rnd_bloc_bold_batch(40, dur=5.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=5)
Y[7] = 0.0
Yh = Y.cpu().numpy().astype(np.float64)
rng = np.random.RandomState(1)
W0 = 1e-3 * rng.randn(40, 300)
W0[3] *= 1e9
W0[21] = np.nan
lmax = solver.lambda_max(Y, hrf).cpu().numpy()
lam = np.where(np.arange(40) % 2 == 0, 1.0, 0.7 * lmax)
lam[7] = 1.0
for n in (1, 2, 5, 20, 120):
    Wo, _, _ = c_oracle.fista_batch(Yh, hrf, lam, 1.0 / lip, n, W0=W0, threads=8)
    for force in ("mfma", "mfmaonly"):
        W, _, nd = solver.fista_solve(Y, hrf, lam, 1.0 / lip, n, W0=torch.from_numpy(W0).cuda(), force=force)
        Wn = W.cpu().numpy(); nd = nd.cpu().numpy()
        err = np.linalg.norm(Wn - Wo, axis=1) / (np.linalg.norm(Wo, axis=1) + 1e-300)
        print(n, force, " ".join("%d:%.1e%s" % (i, err[i], "*" if nd[i] < 0 else "") for i in range(40) if i not in (21,)))
