"""Batched blind deconvolution (one theta per voxel), 50 k voxels: end-to-end time (development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybold_amd import blind, data
from pybold_amd.hrf_model import spm_hrf
t_r, dur = 0.75, 20.0
h_true = spm_hrf(0.7, t_r, dur, False)[0]
Yb, _, _ = data.gen_rnd_bloc_bold_batch(50000, dur=3.75, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=10.0, seed=0)
for nb_iter in (20, 50):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        X, Z, W, H, d = blind.bd_batch(Yb, t_r, lbda=1.7, hrf_dur=dur, nb_iter=nb_iter)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("bd_batch 50k voxels, %d outer x %d inner iterations: %.1f ms   median theta %.4f"
          % (nb_iter, nb_iter, dt * 1e3, float(__import__("numpy").median(d["theta"]))), flush=True)
