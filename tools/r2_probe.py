"""Round-2 probe: small-batch rates per kernel form, config 4 with the device theta-step."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver, data, distributed
from pybold_amd.hrf_model import spm_hrf

hrf30 = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27

def rate(V, force, reps=5, nit=500):
    Y = torch.randn(V, 300, device="cuda", dtype=torch.float32)
    plan = solver.FistaPlan(Y, hrf30, 1.0, step, nit, force=force)
    plan.run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); plan.run(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best

for V in (10000, 12500, 16384, 25000, 50000, 100000):
    line = "V=%6d" % V
    for force in (None, "fast1", "fast2"):
        dt = rate(V, force)
        line += "  %-5s %7.3f ms %5.3f G" % (force or "auto", dt * 1e3, V * 500 / dt / 1e9)
    print(line, flush=True)

t_r, dur = 0.75, 20.0
h_true = spm_hrf(0.7, t_r, dur, False)[0]
Yb, _, _ = data.gen_rnd_bloc_bold_batch(50000, dur=3.75, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0,
                                        std_dur=1.0, snr=10.0, seed=0)
for solver_name in ("device", "device", "lbfgsb"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Wb, h, d = distributed.bd_shared(Yb, t_r, lbda=1.7, hrf_dur=dur, nb_iter=20, nb_inner=100,
                                     theta_solver=solver_name)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("cfg4 bd_shared[%s] 50k voxels, 20 outer x 100 inner: %.1f ms, theta=%.6f (true 0.7) J[-1]=%.6f"
          % (solver_name, dt * 1e3, d["theta"][-1], d["J"][-1]), flush=True)
# pieces of the theta-step
W = torch.randn(50000, 300, device="cuda", dtype=torch.float64)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("integ_op 50k: %.3f ms" % t(lambda: solver.integ_op(W)))
Z = solver.integ_op(W)
print("normal_eq 50k (K=27): %.3f ms" % t(lambda: solver.hrf_normal_eq(Z, Yb, 27)))
ne = solver.hrf_normal_eq(Z, Yb, 27)
print("theta_fit M=1: %.3f ms" % t(lambda: solver.theta_fit(ne, t_r, dur, (0.6, 1.9))))
pv = solver.hrf_normal_eq(Z, Yb, 27, per_voxel=True)
print("normal_eq per-voxel 50k: %.3f ms; theta_fit M=50k: %.3f ms" % (
    t(lambda: solver.hrf_normal_eq(Z, Yb, 27, per_voxel=True)), t(lambda: solver.theta_fit(pv, t_r, dur, (0.6, 1.9)), reps=2)))
