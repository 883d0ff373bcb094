"""CPU-only, world_size 2, gloo backend: the multi-rank logic of
pybold_amd.distributed (contiguous voxel shards, ONE 16-byte all-reduce per
shared-HRF cost evaluation, identical scalar L-BFGS-B step on every rank).
The per-rank cost here comes from the CPU oracle standing in for the HIP
reduction kernel (pb_hrf_cost) so that the logic is exercised without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import pybold_oracle as orc
from pybold_amd import distributed

T_R, HRF_DUR, THETA_TRUE = 0.75, 20.0, 0.8


def make_problem(n_vox=6, n=160, seed=0):
    rng = np.random.RandomState(seed)
    h = orc.spm_hrf(THETA_TRUE, T_R, HRF_DUR, False)[0]
    Z = np.zeros((n_vox, n))
    for v in range(n_vox):
        for _ in range(3):
            o = rng.randint(0, n - 20)
            Z[v, o:o + rng.randint(5, 15)] = 1.0
    Y = orc.causal_conv(h, Z) + 0.01 * rng.randn(n_vox, n)
    return Z, Y


def oracle_local_cost(Z, Y):
    def local_cost(thetas):
        out = []
        for th in thetas:
            h = orc.spm_hrf(th, T_R, HRF_DUR, False)[0]
            out.append(0.5 * np.sum(np.square(Y - orc.causal_conv(h, Z))))
        return np.array(out)
    return local_cost


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Z, Y = make_problem()
        lo, hi = distributed.shard_bounds(len(Y), world, rank)
        comm = distributed.Comm()
        assert comm.world_size == world and comm.rank == rank
        total = comm.allreduce_sum([float(hi - lo), 1.0])
        assert total[0] == len(Y) and total[1] == world
        theta, f, evals = distributed.shared_theta_fit(
            oracle_local_cost(Z[lo:hi], Y[lo:hi]), 2.0, [(0.6, 1.9)], comm)
        ret[rank] = (theta, f, evals)
    finally:
        dist.destroy_process_group()


def test_shared_theta_fit_two_ranks_equals_one_process():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, free_port(), ret), nprocs=world, join=True)
    (t0, f0, e0), (t1, f1, e1) = ret[0], ret[1]
    assert t0 == t1 and f0 == f1 and e0 == e1        # identical step on every rank
    Z, Y = make_problem()
    ts, fs, _ = distributed.shared_theta_fit(oracle_local_cost(Z, Y), 2.0, [(0.6, 1.9)],
                                             distributed.Comm())
    assert t0 == pytest.approx(ts, rel=1e-7)
    assert f0 == pytest.approx(fs, rel=1e-7)
    assert t0 == pytest.approx(THETA_TRUE, abs=5e-3)   # recovers the generating dilation


def test_single_voxel_shared_fit_equals_reference_style_fit():
    """V = 1, one rank: the shared fit is the reference's per-voxel theta-step
    (bounded L-BFGS-B with a forward-difference gradient, bold_signal.py:329-333)."""
    from scipy.optimize import fmin_l_bfgs_b
    Z, Y = make_problem(n_vox=1)
    theta_ref, f_ref, _ = fmin_l_bfgs_b(func=orc.hrf_fit_err, x0=1.9, args=(Z[0], Y[0], T_R, HRF_DUR),
                                        bounds=[(0.6, 1.9)], approx_grad=True, maxiter=999,
                                        pgtol=1.0e-12)
    theta, f, _ = distributed.shared_theta_fit(oracle_local_cost(Z, Y), 2.0, [(0.6, 1.9)],
                                               distributed.Comm())
    assert theta == pytest.approx(float(theta_ref[0]), rel=1e-5)
    assert f == pytest.approx(float(f_ref), rel=1e-6)
