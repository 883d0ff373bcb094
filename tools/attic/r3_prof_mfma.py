"""A few launches of the matrix-pipe kernel and of the pair kernel, one full round each -- for rocprofv3 passes."""
import sys
import torch
sys.path.insert(0, ".")
from pybold_amd import data, solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
Y, _, _ = data.gen_rnd_bloc_bold_batch(16384, dur=5.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0,
                                       snr=1.0, seed=1, device=torch.device("cuda"))
for _ in range(4):
    solver.fista_solve(Y, hrf, 1.0, step, 500, force="mfma")
    solver.fista_solve(Y, hrf, 1.0, step, 500, force="fast2")
torch.cuda.synchronize()
