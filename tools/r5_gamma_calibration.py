"""Round 5: error of the matrix-pipe form and of the float32 vector form against coherence gamma = lambda_max / (max|y| sum|c|)
over populations of series (white noise, high-pass noise of several orders, alternating / fast-sinusoid carriers plus a
random fraction of an ordinary series), lambda in {0, 0.05 lambda_max, 1}: where must the float64 guard start?"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from oracle import c_oracle, pybold_oracle as orc
from pybold_amd import data, solver

n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_taps = int(sys.argv[2]) if len(sys.argv) > 2 else 30       # (34+: the three-near-tile variants of the matrix-pipe forms)
hrf = orc.spm_hrf(1.0, 30.0 / n_taps, 30.0)[0][:n_taps]
N = n_scans
lip = 0.9 * orc.spectral_radius_est(orc._MatrixFreeH(hrf), np.random.RandomState(0).randn(N))
step = 1.0 / lip
rng = np.random.RandomState(5)
t = np.arange(N)
Yg, _, _ = data.gen_rnd_bloc_bold_batch(512, dur=N / 60.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=3)
Yg = Yg.cpu().numpy().astype(np.float64)
rows = [rng.randn(1024, N)]
for order in (1, 2):
    rows.append(np.diff(np.concatenate([np.zeros((512, order)), rng.randn(512, N)], axis=1), n=order, axis=1))
for per in (2, 3, 4, 5, 6, 8):
    carrier = np.cos(2 * np.pi * t / per + rng.uniform(0, 6.28, (512, 1)))
    frac = 10.0 ** rng.uniform(-4, 0, (512, 1))
    rows.append(carrier + frac * Yg / np.abs(Yg).max(axis=1, keepdims=True))
Y = np.concatenate(rows)
Yd = torch.from_numpy(Y.astype(np.float32)).cuda()
Yo = Yd.cpu().numpy().astype(np.float64)
lmax = solver.lambda_max(Yd, hrf)
c = np.cumsum(np.r_[hrf, np.zeros(N - len(hrf))])
gamma = lmax.cpu().numpy() / (np.abs(Yo).max(axis=1) * np.abs(c).sum())


def rel(a, b):
    return np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-300)


fam = np.concatenate([np.full(1024, 0), np.full(512, 1), np.full(512, 2)] + [np.full(512, 3 + i) for i in range(6)])
fam_names = ["white noise", "diff noise", "diff^2 noise"] + ["carrier period %d + fraction" % p for p in (2, 3, 4, 5, 6, 8)]
MP, VEC = ("mfma2only", "valu") if N > 640 else ("mfmaonly", "fast1")      # (641+ scans: the four-wave form, the one-problem-per-wave form)
err = {MP: np.zeros(len(Y)), VEC: np.zeros(len(Y)), None: np.zeros(len(Y))}      # (None: the DEFAULT dispatch, guard on)
err_l = {k: np.zeros((3, len(Y))) for k in err}
li = -1
for lam in (0.0, 0.05 * lmax, 1.0):
    li += 1
    lam_o = lam.cpu().numpy() if torch.is_tensor(lam) else lam
    ref, _, _ = c_oracle.fista_batch(Yo, hrf, lam_o, step, 500, threads=16)
    xr, zr = orc.fista_outputs(ref, hrf)
    nz = np.linalg.norm(ref, axis=1) > 0
    for force in err:
        W, _, nd = solver.fista_solve(Yd, hrf, lam, step, 500, force=force)
        X, Z = solver.fista_outputs(W, hrf)
        e = np.maximum(rel(W.cpu().numpy(), ref), np.maximum(rel(Z.cpu().numpy(), zr), rel(X.cpu().numpy(), xr)))
        e[~nz] = 0.0
        if force is not None:
            e[(nd < 0).cpu().numpy()] = 0.0                # handed back by the form's own guards: re-solved elsewhere
        err[force] = np.maximum(err[force], e)
        err_l[force][li] = e
print("# %d series x %d scans; worst error (diff_z, z, x; lambda = 0, 0.05 lambda_max, 1) per bin of gamma" % (len(Y), N))
print("%-22s %6s %12s %12s %12s" % ("gamma", "series", "matrix pipe", "float32 vector", "DEFAULT"))
edges = [0, 1e-3, 2e-3, 3e-3, 4e-3, 5e-3, 6e-3, 7e-3, 8e-3, 1e-2, 1.5e-2, 2e-2, 3e-2, 4e-2, 5e-2, 7e-2, 1e-1, 1.0]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (gamma >= lo) & (gamma < hi)
    if m.any():
        print("[%.0e, %.0e) %16d %12.1e %12.1e %12.1e" % (lo, hi, m.sum(), err[MP][m].max(), err[VEC][m].max(), err[None][m].max()))
wn = gamma[:1024]
print("white noise: gamma min %.2e, 1 %% %.2e, 5 %% %.2e, median %.2e" % (wn.min(), np.quantile(wn, 0.01), np.quantile(wn, 0.05), np.median(wn)))

gamma2 = lmax.cpu().numpy() / (np.linalg.norm(Yo, axis=1) * np.abs(c).sum() / np.sqrt(N))
print("# the same per bin of gamma_2 = lambda_max / (||y||_2 sum|c| / sqrt(N))    (RMS instead of max|y|)")
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (gamma2 >= lo) & (gamma2 < hi)
    if m.any():
        print("[%.0e, %.0e) %16d %12.1e %12.1e %12.1e   %s" % (lo, hi, m.sum(), err[MP][m].max(), err[VEC][m].max(), err[None][m].max(),
              " ".join("%s:%d" % (fam_names[f].split()[0][:4] + fam_names[f].split()[-3][-1:] if f >= 3 else fam_names[f][:5], (m & (fam == f)).sum()) for f in range(len(fam_names)) if (m & (fam == f)).any())))
print("white noise: gamma_2 min %.2e, 1 %% %.2e, median %.2e; block signals at SNR 1 dB: gamma_2 min %.2e median %.2e"
      % (gamma2[:1024].min(), np.quantile(gamma2[:1024], 0.01), np.median(gamma2[:1024]),
         (solver.lambda_max(torch.from_numpy(Yg.astype(np.float32)).cuda(), hrf).cpu().numpy() / (np.linalg.norm(Yg, axis=1) * np.abs(c).sum() / np.sqrt(N))).min(),
         np.median(solver.lambda_max(torch.from_numpy(Yg.astype(np.float32)).cuda(), hrf).cpu().numpy() / (np.linalg.norm(Yg, axis=1) * np.abs(c).sum() / np.sqrt(N)))))
print("# DEFAULT dispatch (partitioned call, conditioning guard on): worst error over ALL %d series: %.1e" % (len(Y), err[None].max()))
print("worst kept problems of the matrix-pipe form with gamma >= 1e-2:")
m = gamma >= 1e-2
order = np.argsort(-err[MP] * m)[:12]
for i in order:
    print("  %-28s gamma %.2e  errors at lambda 0 / 0.05 lmax / 1: %.1e %.1e %.1e   (vector form %.1e %.1e %.1e)  ||w||/||y|| ~ lmax %.2e"
          % (fam_names[fam[i]], gamma[i], err_l[MP][0][i], err_l[MP][1][i], err_l[MP][2][i],
             err_l[VEC][0][i], err_l[VEC][1][i], err_l[VEC][2][i], float(lmax[i])))
print("per family, gamma >= 1e-2: worst matrix-pipe error")
for f, nm in enumerate(fam_names):
    mm = m & (fam == f)
    if mm.any():
        print("  %-28s %5d series  %.1e  (lambda 0: %.1e, 0.05 lmax: %.1e, 1: %.1e)" % (nm, mm.sum(), err[MP][mm].max(),
              err_l[MP][0][mm].max(), err_l[MP][1][mm].max(), err_l[MP][2][mm].max()))
