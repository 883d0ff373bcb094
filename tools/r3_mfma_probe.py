"""Matrix-pipe kernel (fista_mfma.h) vs the pair kernel: parity against the C oracle and time."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import data, solver
from pybold_amd.hrf_model import spm_hrf
from oracle import c_oracle
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.2744579345
V = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=5.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0,
                                       snr=1.0, seed=1, device=torch.device("cuda"))
for n_it in (1, 2, 10, 500):
    Wm, _, nd = solver.fista_solve(Y[:64], hrf, 1.0, step, n_it, force="mfma")
    Wo, _, _ = c_oracle.fista_batch(Y[:64].cpu().numpy().astype(np.float64), hrf, 1.0, step, n_it, threads=8)
    Wm = Wm.cpu().numpy()
    err = (np.linalg.norm(Wm - Wo, axis=1) / (np.linalg.norm(Wo, axis=1) + 1e-300)).max()
    print("n_iter %4d: max rel L2 vs C oracle %.3e   (n_done min %d)" % (n_it, err, int(nd.min())), flush=True)
def t(**kw):
    solver.fista_solve(Y, hrf, 1.0, step, 500, **kw); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); solver.fista_solve(Y, hrf, 1.0, step, 500, **kw); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
for P in (16384, V):
    Yp = Y[:P]
    Ysave = Y; Y = Yp
    a, b = t(force="mfma"), t()
    print("P=%6d: mfma %8.3f ms (%.3e voxel-it/s)   library (pair) %8.3f ms (%.3e)" % (P, a, P * 500 / a * 1e3, b, P * 500 / b * 1e3), flush=True)
    Y = Ysave
