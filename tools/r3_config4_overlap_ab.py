"""Config 4 end to end with and without the remainder of the z-step running beside the normal equations
(HipOps.overlap).  Usage (GPU): python tools/r3_config4_overlap_ab.py [voxels]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pybold_amd import data, distributed  # noqa: E402
from pybold_amd.hrf_model import spm_hrf  # noqa: E402


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    dev = torch.device("cuda:0")
    t_r, hrf_dur, N = 0.75, 20.0, 300
    h_true = spm_hrf(0.7, t_r, hrf_dur, False)[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=N * t_r / 60.0, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0,
                                           std_dur=1.0, snr=10.0, seed=4000, device=dev)
    for rep in range(3):
        for overlap in (False, True):
            ops = distributed.HipOps(t_r, hrf_dur, N)
            ops.overlap = overlap
            for _ in range(2):
                distributed.bd_shared(Y, t_r, lbda=1.7, theta_0=2.0, hrf_dur=hrf_dur, nb_iter=20, nb_inner=100, ops=ops)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                _, _, d = distributed.bd_shared(Y, t_r, lbda=1.7, theta_0=2.0, hrf_dur=hrf_dur, nb_iter=20, nb_inner=100, ops=ops)
            torch.cuda.synchronize()
            print("V=%d overlap=%-5s %.3f ms per bd_shared   theta_final %.9f" % (V, overlap, (time.perf_counter() - t0) / n * 1e3, d["theta"][-1]), flush=True)


if __name__ == "__main__":
    main()
