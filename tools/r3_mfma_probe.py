"""Matrix-pipe kernel (fista_mfma.h) vs the vector forms: parity against the C oracle (plain, cost trace,
other series lengths, per-problem lambda) and time."""
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
def rel(a, b):
    return (np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-300)).max()
for n_it in (1, 3, 500):
    Wm, Jm, nd = solver.fista_solve(Y[:80], hrf, 1.0, step, n_it, want_J=True, force="mfma")
    Wo, Jo, _ = c_oracle.fista_batch(Y[:80].cpu().numpy().astype(np.float64), hrf, 1.0, step, n_it, want_J=True, threads=8)
    print("n_iter %4d: W %.3e  J %.3e (n_done min %d)" % (n_it, rel(Wm.cpu().numpy(), Wo),
          np.abs(Jm.cpu().numpy() / Jo - 1).max(), int(nd.min())), flush=True)
for N in (129, 160, 200, 240, 256, 257, 284, 288, 289, 300, 320):
    K = 27 if N < 290 else 33
    h = spm_hrf(1.0, t_r=1.0, dur=float(K))[0]
    Yn = torch.randn(40, N, device="cuda")
    lam = np.linspace(0.2, 3.0, 40)
    Wm, _, _ = solver.fista_solve(Yn, h, lam, step, 200, force="mfma")
    name = solver.which_kernel(N, len(h), 100000)
    Wo, _, _ = c_oracle.fista_batch(Yn.cpu().numpy().astype(np.float64), h, lam, step, 200, threads=8)
    print("N=%3d K=%2d: W %.3e   [%s]" % (N, len(h), rel(Wm.cpu().numpy(), Wo), name.split("(")[0]), flush=True)
def t(Yp, **kw):
    solver.fista_solve(Yp, hrf, 1.0, step, 500, **kw); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); solver.fista_solve(Yp, hrf, 1.0, step, 500, **kw); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
print("%8s %10s %10s %10s %10s" % ("P", "lib", "lib +J", "valu", "valu +J"))
for P in (8192, 10000, 12500, 16384, 20000, 25000, 50000, V):
    Yp = Y[:P]
    print("%8d %10.3f %10.3f %10.3f %10.3f" % (P, t(Yp), t(Yp, want_J=True), t(Yp, force="valu"), t(Yp, want_J=True, force="valu")), flush=True)
