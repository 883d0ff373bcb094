"""Throughput of every solver path / BASELINE config on one GPU (development aid)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver, data, distributed
from pybold_amd.hrf_model import spm_hrf
from pybold_amd.utils import gram_frobenius

def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best

hrf30 = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
hrf27 = spm_hrf(1.0, t_r=0.75, dur=20., normalized_hrf=False)[0]
step = 1.0 / 723876.27
rows = []
def case(name, V, N, hrf, nit, **kw):
    Y = torch.randn(V, N, device="cuda", dtype=torch.float32)
    y_rep = kw.get("y_rep", 1)
    # a step that is valid for THIS shape (1 / ||A^T A||_F <= 1 / rho): with the N = 300 constant a 600-scan solve
    # diverges, and since round 4 the matrix-pipe forms notice (range guard -> every problem re-solved on the vector forms)
    step = 1.0 / gram_frobenius(hrf, N)
    dt = timeit(lambda: solver.fista_solve(Y, hrf, kw.pop("lbda", 1.0) if False else kw.get("lbda", 1.0), step, nit,
                                           **{k: v for k, v in kw.items() if k != "lbda"}))
    vi = V * y_rep * nit / dt
    print("%-42s V=%7d N=%3d K=%2d it=%4d  %9.3f ms  %.3e voxel-iter/s" % (name, V * y_rep, N, len(hrf), nit, dt * 1e3, vi), flush=True)

case("fast (19,30) cfg3", 100000, 300, hrf30, 500)
case("fast (19,30) cfg2", 10000, 300, hrf30, 500)
case("fast (19,30) +J", 100000, 300, hrf30, 500, want_J=True)
case("fast (19,30) stop=loops tol=0", 100000, 300, hrf30, 500, stop="loops", tol=0.0)
case("fast (19,27) cfg4 shape", 50000, 300, hrf27, 500)
case("fast (15,27) N=240", 50000, 240, hrf27, 500)
case("fast (18,28) N=284", 50000, 284, hrf27, 500)
case("fast (38,30) N=600", 50000, 600, hrf30, 500)
case("fast (8,16)  N=128 K=16", 100000, 128, hrf30[:16], 500)
case("cfg5: 50k voxels x 20 lambdas", 50000, 300, hrf30, 500, y_rep=20, lbda=np.tile(np.logspace(-2, 0, 20), 50000))
case("generic N=300", 10000, 300, hrf30, 100, force="generic")
case("generic +window stop tol=0", 10000, 300, hrf30, 100, force="generic", stop="window", tol=0.0)
case("generic N=1000 K=40", 5000, 1000, np.r_[hrf30, np.zeros(10) + .01], 100)
# outputs / stats / cost kernels
Y = torch.randn(100000, 300, device="cuda", dtype=torch.float32)
W = torch.randn(100000, 300, device="cuda", dtype=torch.float64)
for name, fn in (("fista_outputs", lambda: solver.fista_outputs(W, hrf30)),
                 ("fista_stats", lambda: solver.fista_stats(W, Y, hrf30)),
                 ("hrf_cost x2", lambda: solver.hrf_cost(W, Y, np.stack([hrf30, hrf30]))),
                 ("integ_op", lambda: solver.integ_op(W))):
    dt = timeit(fn)
    print("%-42s 100k x 300: %.3f ms  (%.1f GB/s of touched data)" % (name, dt * 1e3, (W.numel() * 8 * 2) / dt / 1e9), flush=True)
# config 4
t_r, dur = 0.75, 20.0
h_true = spm_hrf(0.7, t_r, dur, False)[0]
Yb, _, _ = data.gen_rnd_bloc_bold_batch(50000, dur=3.75, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=10.0, seed=0)
distributed.bd_shared(Yb, t_r, lbda=1.7, hrf_dur=dur, nb_iter=20, nb_inner=100)       # first call: one-time set-up
torch.cuda.synchronize(); t0 = time.perf_counter()
Wb, h, d = distributed.bd_shared(Yb, t_r, lbda=1.7, hrf_dur=dur, nb_iter=20, nb_inner=100)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("cfg4 bd_shared 50k voxels, 20 outer x 100 inner: %.3f s, theta=%.4f (true 0.7), evals/outer=%s" % (dt, d["theta"][-1], d["evals"][:5]))
print("  -> %.3e voxel-iter/s incl. theta-steps" % (50000 * 21 * 100 / dt))
