"""Every kernel of the library a few times each, for rocprofv3 (development aid).

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/var_trace -- python3 tools/profile_variants.py
    rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv \
        -d gpurun_out/var_sq -- python3 tools/profile_variants.py
    python tools/summarise_variants.py gpurun_out/var_trace gpurun_out/var_sq profiles/r2_variants.json

Stop rules run with tol = 0 so that every problem executes all its iterations (the rate of
the rule's arithmetic, not of a lucky early exit)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybold_amd import solver, data, distributed
from pybold_amd.hrf_model import spm_hrf

REPS = 3
hrf30 = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
hrf27 = spm_hrf(1.0, t_r=0.75, dur=20., normalized_hrf=False)[0]
step = 1.0 / 723876.27
gen = torch.Generator(device="cuda").manual_seed(0)
Y = torch.randn(100000, 300, device="cuda", dtype=torch.float32, generator=gen)


def rep(fn):
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()


# the register-resident float32-FIR kernels, config 3 shape
rep(lambda: solver.fista_solve(Y, hrf30, 1.0, step, 500))
rep(lambda: solver.fista_solve(Y, hrf30, 1.0, step, 500, want_J=True))
rep(lambda: solver.fista_solve(Y, hrf30, 1.0, step, 500, stop="loops", tol=0.0))
rep(lambda: solver.fista_solve(Y, hrf30, 1.0, step, 500, stop="window", tol=0.0))
rep(lambda: solver.fista_solve(Y, hrf30, 1.0, step, 500, stop="window", tol=0.0, want_J=True))
rep(lambda: solver.fista_solve(Y, hrf30, 1.0, step, 500, force="fast1"))
rep(lambda: solver.fista_solve(Y, hrf30, 1.0, step, 500, force="fast2d"))
# config 5: 50 k voxels x 20 lambdas, y shared by the 20 problems of a voxel
lam = np.tile(np.logspace(-2, 0, 20), 50000)
rep(lambda: solver.fista_solve(Y[:50000], hrf30, lam, step, 500, y_rep=20))
# sub-round batches (side stream)
rep(lambda: solver.fista_solve(Y[:12500], hrf30, 1.0, step, 500))
rep(lambda: solver.fista_solve(Y[:10000], hrf30, 1.0, step, 500))
# float64 end to end
Y64 = Y[:50000].double()
rep(lambda: solver.fista_solve(Y64, hrf30, 1.0, step, 500))
rep(lambda: solver.fista_solve(Y64, hrf30, 1.0, step, 500, stop="window", tol=0.0, want_J=True))
# LDS kernel
rep(lambda: solver.fista_solve(Y[:10000], hrf30, 1.0, step, 100, force="generic"))
# one-launch helpers
W = torch.randn(100000, 300, device="cuda", dtype=torch.float64, generator=gen)
rep(lambda: solver.fista_outputs(W, hrf30))
rep(lambda: solver.fista_stats(W, Y, hrf30))
rep(lambda: solver.integ_op(W))
rep(lambda: solver.lambda_max(Y, hrf30))
# blind step (config 4 shape: N = 300, K = 27)
t_r, dur = 0.75, 20.0
h_true = spm_hrf(0.7, t_r, dur, False)[0]
Yb, _, _ = data.gen_rnd_bloc_bold_batch(50000, dur=3.75, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0,
                                        std_dur=1.0, snr=10.0, seed=0)
Z = solver.integ_op(W[:50000])
rep(lambda: solver.hrf_normal_eq(Z, Yb, 27))
rep(lambda: solver.hrf_normal_eq(Z, Yb, 27, per_voxel=True))
ne = solver.hrf_normal_eq(Z, Yb, 27)
pv = solver.hrf_normal_eq(Z, Yb, 27, per_voxel=True)
rep(lambda: solver.theta_fit(ne, t_r, dur, (0.6, 1.9)))
rep(lambda: solver.theta_fit(pv, t_r, dur, (0.6, 1.9)))
for _ in range(2):
    distributed.bd_shared(Yb, t_r, lbda=1.7, hrf_dur=dur, nb_iter=20, nb_inner=100)
torch.cuda.synchronize()
from pybold_amd import blind
for _ in range(2):
    blind.bd_batch(Yb[:20000], t_r, lbda=1.7, hrf_dur=dur, nb_iter=20)
torch.cuda.synchronize()
print("done")
