"""500-iteration parity of every realistic shape against the C oracle (development aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver, data
from pybold_amd.hrf_model import spm_hrf
from pybold_amd.linear import ConvAndLinear, DiscretInteg
from pybold_amd.utils import spectral_radius_est, gram_frobenius
from oracle import c_oracle
for (N, tr, hdur, snr, mode) in ((600, 1.0, 30., 1.0, "rho"), (240, 0.75, 20., 10.0, "fro"), (284, 0.72, 20., 10.0, "fro"),
                                 (300, 0.75, 20., 10.0, "fro"), (300, 2.0, 30., 1.0, "rho"), (150, 2.0, 30., 5.0, "rho")):
    hrf = spm_hrf(1.0, t_r=tr, dur=hdur, normalized_hrf=(mode == "rho"))[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(512, dur=N * tr / 60.0, tr=tr, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=snr, seed=2)
    assert Y.shape[1] == N, (Y.shape, N)
    if mode == "rho":
        np.random.seed(0)
        L = 0.9 * spectral_radius_est(ConvAndLinear(DiscretInteg(), hrf, N, N), (N,))
    else:
        L = gram_frobenius(hrf, N)
    for lbda in (0.1, 1.0):
        W, _, _ = solver.fista_solve(Y, hrf, lbda, 1.0 / L, 500)
        Wo, _, _ = c_oracle.fista_batch(Y.cpu().numpy().astype(np.float64), hrf, lbda, 1.0 / L, 500, threads=16)
        Wg = W.cpu().numpy()
        err = (np.linalg.norm(Wg - Wo, axis=1) / (np.linalg.norm(Wo, axis=1) + 1e-300))
        Zg, Zo = np.cumsum(Wg, 1), np.cumsum(Wo, 1)
        errz = (np.linalg.norm(Zg - Zo, axis=1) / (np.linalg.norm(Zo, axis=1) + 1e-300))
        print("N=%d K=%d tr=%.2f %s lbda=%.1f fast=%s: diff_z max %.2e median %.2e | z max %.2e" % (
            N, len(hrf), tr, mode, lbda, solver.has_fast_path(N, len(hrf)), err.max(), np.median(err), errz.max()), flush=True)
