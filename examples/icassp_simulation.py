"""The reference's ICASSP-2019 simulation (examples/icassp_2019/simulation.py:100-155) with no
host loop over voxels and no host round trip inside the solve: 6 SNR levels x 100 voxels are
generated on the GPU, deconvolved blindly in ONE batch of 600 voxels (one HRF dilation and one
lambda per voxel; the reference fans `bd` out over voxels with joblib, :62-72), inf-norm
normalised on the GPU (:59) and scored against the generating HRF / block signals.

    python examples/icassp_simulation.py [--voxels 100] [--iters 500]

Prints the mean +- std relative L2 error of the HRF and of the block signal per SNR level --
the two curves of the reference's figure."""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pybold_amd import bd, data                                   # noqa: E402
from pybold_amd.hrf_model import MAX_DELTA, MIN_DELTA, spm_hrf     # noqa: E402
from pybold_amd.utils import inf_norm                             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=100, help="voxels per SNR level")
    ap.add_argument("--iters", type=int, default=500, help="nb_iter of bd (simulation.py:115)")
    args = ap.parse_args()

    t_r, hrf_dur, theta_true, dur_run = 0.75, 20.0, 0.7, 3.0      # simulation.py:102-106
    l_snr = [1.0, 3.0, 5.0, 10.0, 15.0, 20.0]                      # :107
    l_lbda = [4.0, 3.0, 2.5, 1.7, 1.6, 1.5]                        # :113-114 ("already grid-search")
    n_vox = args.voxels
    orig_hrf, _ = spm_hrf(theta_true, t_r=t_r, dur=hrf_dur)        # normalised, as :118

    # one batch: voxel v belongs to SNR level v // n_vox
    snr = torch.tensor(np.repeat(l_snr, n_vox), dtype=torch.float64, device="cuda")
    lbda = torch.tensor(np.repeat(l_lbda, n_vox), dtype=torch.float64, device="cuda")
    Y, _, blocks = data.gen_rnd_bloc_bold_batch(len(l_snr) * n_vox, dur=dur_run, tr=t_r, hrf=orig_hrf,
                                                nb_events=5, avg_dur=12.0, std_dur=1.0, snr=snr, seed=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, z, diff_z, h, d = bd(Y, t_r, lbda=lbda, theta_0=MAX_DELTA, hrf_dur=hrf_dur,
                            bounds=[(MIN_DELTA + 0.1, MAX_DELTA - 0.1)], nb_iter=args.iters)
    est_blocks, est_hrfs = inf_norm([z, h])                        # CUDA in -> CUDA out
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    ref_hrf = torch.from_numpy(orig_hrf).cuda()
    err_h = ((est_hrfs - ref_hrf[None]).norm(dim=1) / ref_hrf.norm()).reshape(len(l_snr), n_vox)
    err_z = ((est_blocks - blocks).norm(dim=1) / blocks.norm(dim=1)).reshape(len(l_snr), n_vox)
    print("blind deconvolution of %d voxels x %d scans, %d outer x %d inner iterations: %.2f s"
          % (Y.shape[0], Y.shape[1], args.iters, args.iters, dt))
    print(" SNR [dB]  lambda   HRF error (mean +- std)   block-signal error (mean +- std)   median theta")
    theta = torch.from_numpy(d["theta"]).reshape(len(l_snr), n_vox)
    for i, (s, lb) in enumerate(zip(l_snr, l_lbda)):
        print("  %5.1f    %4.1f     %.4f +- %.4f           %.4f +- %.4f                  %.4f"
              % (s, lb, float(err_h[i].mean()), float(err_h[i].std()), float(err_z[i].mean()),
                 float(err_z[i].std()), float(theta[i].median())))


if __name__ == "__main__":
    main()
