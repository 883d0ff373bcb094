"""Config 4's exchange step on real kernels with more than one rank: the ranks share the one
GPU of the box and synchronise over gloo (RCCL refuses two ranks on one device), plus one rank
over RCCL (the collective runs on the device, no host copy).  An N-rank run must reproduce the
one-process run: same theta after every outer iteration, same cost, same iterates."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

T_R, DUR = 0.75, 20.0
V, NB_ITER, NB_INNER = 700, 4, 30


def _batch():
    from pybold_amd import data
    from pybold_amd.hrf_model import spm_hrf
    h_true = spm_hrf(0.8, T_R, DUR, False)[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=3.75, tr=T_R, hrf=h_true, nb_events=5, avg_dur=12.0,
                                           std_dur=1.0, snr=10.0, seed=3)
    return Y


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, backend, ret):
    import torch.distributed as dist
    from pybold_amd import distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if backend.startswith("nccl"):
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Y = _batch()
        # 3 ranks: ceil(700 / 3) = 234 rows each but the LAST rank is given none, to cover a rank
        # that owns no voxel and still has to join every all-reduce
        if world == 3:
            lo, hi = [(0, 400), (400, 700), (700, 700)][rank]
        else:
            lo, hi = distributed.shard_bounds(V, world, rank)
        if backend == "nccl-graph":              # the whole loop, all-reduces included, as ONE captured HIP graph
            runner = distributed.BdSharedGraph(Y[lo:hi].contiguous(), T_R, lbda=1.7, hrf_dur=DUR, nb_iter=NB_ITER,
                                               nb_inner=NB_INNER)
            assert runner.graph is not None, runner.fallback
            runner.launch()
            runner.launch()                      # a replay gives the same answer
            W, h, d = runner.result()
        else:
            W, h, d = distributed.bd_shared(Y[lo:hi].contiguous(), T_R, lbda=1.7, hrf_dur=DUR, nb_iter=NB_ITER,
                                            nb_inner=NB_INNER)
        ret[rank] = (lo, hi, W.cpu().numpy(), h, d["theta"], d["J"])
    finally:
        dist.destroy_process_group()


def _single():
    from pybold_amd import distributed
    W, h, d = distributed.bd_shared(_batch(), T_R, lbda=1.7, hrf_dur=DUR, nb_iter=NB_ITER, nb_inner=NB_INNER)
    return W.cpu().numpy(), h, d["theta"], d["J"]


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (3, "gloo"), (1, "nccl"), (1, "nccl-graph")])
def test_bd_shared_ranks_on_one_gpu_equal_one_process(world, backend):
    W1, h1, th1, J1 = _single()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), backend, ret), nprocs=world, join=True)
    assert len(ret) == world
    Wn = np.zeros_like(W1)
    for rank in range(world):
        lo, hi, W, h, th, J = ret[rank]
        Wn[lo:hi] = W
        np.testing.assert_allclose(th, th1, rtol=0, atol=2e-7)       # every outer iteration (the shards' sums
            # come in another order: 1e-16 on the normal equations, carried from one outer iteration to the next)
        np.testing.assert_allclose(J, J1, rtol=1e-7)
        np.testing.assert_allclose(h, h1, rtol=1e-6, atol=1e-9)
    scale = np.abs(W1).max()
    assert np.abs(Wn - W1).max() / scale < 1e-5
    assert th1[-1] < th1[0]                                          # moved from 2.0 towards the true 0.8


def test_bd_shared_as_one_hip_graph():
    """`BdSharedGraph`: the whole outer loop captured once and replayed -- the same bits as the eager loop, replay
    after replay, also after the series were refilled in place; gloo communicators fall back to the eager loop."""
    from pybold_amd import distributed
    Y = _batch()
    W1, h1, th1, J1 = _single()
    runner = distributed.BdSharedGraph(Y, T_R, lbda=1.7, hrf_dur=DUR, nb_iter=NB_ITER, nb_inner=NB_INNER)
    assert runner.graph is not None, runner.fallback
    for _ in range(3):
        runner.launch()
    W, h, d = runner.result()
    assert np.array_equal(W.cpu().numpy(), W1) and np.array_equal(d["theta"], th1) and np.array_equal(d["J"], J1)
    np.testing.assert_array_equal(h, h1)
    Y2 = Y.clone()
    Y.mul_(0.5)                                  # new data in the captured buffer
    runner.launch()
    Wh, hh, dh = runner.result()
    We, he, de = distributed.bd_shared(Y, T_R, lbda=1.7, hrf_dur=DUR, nb_iter=NB_ITER, nb_inner=NB_INNER)
    assert torch.equal(Wh, We) and np.array_equal(dh["theta"], de["theta"])
    assert not np.array_equal(dh["theta"], th1)
    Y.copy_(Y2)
