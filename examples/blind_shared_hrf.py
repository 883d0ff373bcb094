#!/usr/bin/env python3
"""Semi-blind deconvolution with one HRF dilation shared by all voxels (and all
ranks): BASELINE config 4.  Single GPU:

    python examples/blind_shared_hrf.py [n_voxels]

Several GPUs (one process per GPU, RCCL all-reduce of the 16-byte cost vector):

    python -m torch.distributed.run --nproc-per-node 8 examples/blind_shared_hrf.py 50000
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pybold_amd import data, distributed, spm_hrf  # noqa: E402

n_voxels = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
if "RANK" in os.environ:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.distributed.init_process_group("nccl", rank=rank, world_size=world,
                                         device_id=torch.device("cuda", torch.cuda.current_device()))
t_r, hrf_dur, theta_true = 0.75, 20.0, 0.7
lo, hi = distributed.shard_bounds(n_voxels, world, rank)
h_true = spm_hrf(theta_true, t_r, hrf_dur, False)[0]
Y, _, _ = data.gen_rnd_bloc_bold_batch(hi - lo, dur=3.75, tr=t_r, hrf=h_true, nb_events=5,
                                       avg_dur=12.0, std_dur=1.0, snr=10.0, seed=rank)
torch.cuda.synchronize()
t0 = time.time()
W, h, d = distributed.bd_shared(Y, t_r, lbda=1.7, hrf_dur=hrf_dur, nb_iter=20, nb_inner=100)
torch.cuda.synchronize()
if rank == 0:
    print("%d voxels on %d GPU(s): theta %.4f (generated with %.2f), normalised cost %.4f, %.2f s"
          % (n_voxels, world, d["theta"][-1], theta_true, d["J"][-1], time.time() - t0))
if "RANK" in os.environ:
    torch.distributed.destroy_process_group()
