"""Regularisation path (BASELINE config 5) on the matrix pipe by partition (round 4): which lambdas of a path the
accuracy guard of the matrix-pipe form hands back, and what the partition ratio of pb_fista_solve_path costs.

usage: python tools/r4_path_probe.py [voxels] > profiles/r4_path_partition.txt
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybold_amd import data, solver              # noqa: E402
from pybold_amd.hrf_model import spm_hrf         # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
L, N, n_iter = 20, 300, 500
dev = torch.device("cuda")
hrf = spm_hrf(1.0, t_r=1.0, dur=30.0)[0]
step = 1.0 / 723876.2744579345
Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=5.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=1000, device=dev)
lmax = solver.lambda_max(Y, hrf)
grid = torch.logspace(-2.0, 0.0, L, dtype=torch.float64, device=dev)
lam = (lmax[:, None] * grid[None, :]).reshape(-1)
P = V * L


def ms(plan, reps=3):
    for _ in range(2):
        plan.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("# config-5 workload: %d voxels x %d lambdas (logspace(-2, 0, %d) x lambda_max) = %d problems, %d iterations" % (V, L, L, P, n_iter))
_, _, nd = solver.fista_solve(Y, hrf, lam, step, n_iter, y_rep=L, force="mfmaonly")
back = (nd.reshape(V, L) < 0).double().mean(dim=0).cpu().numpy()
print("# hand-back rate of the matrix-pipe form's guards by lambda index (everything forced onto it):")
print("  " + " ".join("%5.2f" % g for g in grid.cpu().numpy()))
print("  " + " ".join("%5.3f" % b for b in back))
ref = solver.FistaPlan(Y, hrf, lam, step, n_iter, y_rep=L, force=None)
t_ref = ms(ref)
Wref = ref.W.clone()
print("vector forms (pb_fista_solve with per-problem lambdas): %8.2f ms  %.2f G voxel-iterations/s" % (t_ref, P * n_iter / t_ref / 1e6))
for ratio in (0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.4, 0.6, 1.01):
    plan = solver.FistaPlan(Y, hrf, lam, step, n_iter, y_rep=L, force=None, lmax=lmax, dense_ratio=ratio)
    t = ms(plan)
    dense = int((lam < ratio * lmax.repeat_interleave(L)).sum())
    chk = solver.FistaPlan(Y, hrf, lam, step, n_iter, y_rep=L, force="noresolve", lmax=lmax, dense_ratio=ratio)
    chk.run()
    nb = int((chk.n_done < 0).sum())
    err = float(((plan.W - Wref).norm(dim=1) / (Wref.norm(dim=1) + 1e-300)).max())
    print("ratio %.2f: dense class %7d (%.0f %%), handed back %6d (%.2f %% of it)  %8.2f ms  %.2f G voxel-iterations/s   max rel diff vs vector forms %.1e"
          % (ratio, dense, 100.0 * dense / P, nb, 100.0 * nb / max(dense, 1), t, P * n_iter / t / 1e6, err), flush=True)
