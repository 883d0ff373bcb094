#!/usr/bin/env python3
"""Batched counterpart of the reference's examples/synth_data/deconv.py (no
plotting): generate block BOLD signals, deconvolve them all in one call, report
error against the generating block signals and the wall-clock.

    python examples/synth_deconv.py [n_voxels] [nb_iter]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pybold_amd import data, deconv, spm_hrf  # noqa: E402

n_voxels = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nb_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 500
t_r = 1.0
hrf = spm_hrf(1.0, t_r=t_r, dur=30.0)[0]
noisy, clean, blocks = data.gen_rnd_bloc_bold_batch(n_voxels, dur=5, tr=t_r, hrf=hrf, nb_events=5,
                                                     avg_dur=12.0, std_dur=1.0, snr=1.0, seed=0)
np.random.seed(0)
torch.cuda.synchronize()
t0 = time.time()
x, z, diff_z, J, _, _ = deconv(noisy, t_r, hrf, lbda=1.0, nb_iter=nb_iter, early_stopping=False)
dt = time.time() - t0
clean = clean.cpu().numpy()
err_x = np.linalg.norm(x - clean, axis=1) / np.linalg.norm(clean, axis=1)
print("deconvolved %d voxels x %d scans, %d iterations in %.3f s "
      "(%.3e voxel-iterations/s incl. host transfers)" % (n_voxels, x.shape[1], nb_iter, dt,
                                                          n_voxels * nb_iter / dt))
print("median relative error of the denoised BOLD signal vs the noise-free one: %.3f"
      % np.median(err_x))
print("final normalised cost (median over voxels): %.4f" % np.median(J[:, -1]))
