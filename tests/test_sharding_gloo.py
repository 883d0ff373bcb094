"""CPU-only, world_size 2, gloo backend: the multi-rank logic of
pybold_amd.distributed (contiguous voxel shards; shared-HRF blind step = per-rank
normal equations, ONE all-reduce per outer iteration, identical theta fit on every
rank; and the L-BFGS-B variant with its 16-byte all-reduce per cost evaluation).
The per-rank compute here comes from the CPU oracle standing in for the HIP
kernels (distributed.HipOps) so that the logic is exercised without a GPU."""
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


# ---- shared-HRF blind loop with the device theta-step (normal equations) -------------
from oracle.shared_ops import OracleOps as _OracleOps   # noqa: E402


def OracleOps(n):
    return _OracleOps(n, T_R, HRF_DUR)


def _bd_worker(rank, world, port, n_vox, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, Y = make_problem(n_vox=n_vox)
        lo, hi = distributed.shard_bounds(len(Y), world, rank)
        Yl = torch.from_numpy(Y[lo:hi].astype(np.float32))
        W, h, d = distributed.bd_shared(Yl, T_R, lbda=0.05, hrf_dur=HRF_DUR, nb_iter=3, nb_inner=30,
                                        comm=distributed.Comm(), ops=OracleOps(Y.shape[1]))
        ret[rank] = (lo, hi, W.numpy(), h, d["theta"], d["J"])
    finally:
        dist.destroy_process_group()


def _bd_single(n_vox):
    _, Y = make_problem(n_vox=n_vox)
    return distributed.bd_shared(torch.from_numpy(Y.astype(np.float32)), T_R, lbda=0.05,
                                 hrf_dur=HRF_DUR, nb_iter=3, nb_inner=30,
                                 comm=distributed.Comm(), ops=OracleOps(Y.shape[1]))


@pytest.mark.parametrize("n_vox", [6, 1])
def test_bd_shared_two_ranks_equal_one_process(n_vox):
    """Voxel shards + one all-reduce of the normal equations per outer iteration give the
    same theta trajectory, cost trace and iterates as one process; with n_vox = 1 < world
    size the second rank owns NO voxel and must still join every all-reduce (no hang)."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bd_worker, args=(world, free_port(), n_vox, ret), nprocs=world, join=True)
    Ws, hs, ds = _bd_single(n_vox)
    np.testing.assert_array_equal(ret[0][4], ret[1][4])          # identical theta on every rank
    np.testing.assert_array_equal(ret[0][5], ret[1][5])
    np.testing.assert_allclose(ret[0][4], ds["theta"], rtol=1e-9)
    np.testing.assert_allclose(ret[0][5], ds["J"], rtol=1e-9)
    assert len(ds["J"]) == 5 and ds["J"][0] == 1.0 and (np.diff(ds["J"]) < 0).all()
    for r in range(world):
        lo, hi, W = ret[r][0], ret[r][1], ret[r][2]
        assert W.shape[0] == hi - lo
        np.testing.assert_allclose(W, Ws.numpy()[lo:hi], rtol=1e-7, atol=1e-12)
    if n_vox == 1:
        assert ret[1][1] - ret[1][0] == 0                          # the empty rank
    assert THETA_TRUE < ds["theta"][-1] < 1.9           # moving from theta_0 = 2.0 towards the truth


def test_normal_equation_fit_equals_direct_cost_minimiser():
    """The quadratic-form theta-step returns the minimiser of the DIRECT shared cost
    sum_v 0.5||y_v - h(theta)*z_v||^2; for one voxel that is the reference's theta-step
    objective (hrf_fit_err, bold_signal.py:217-222), whose L-BFGS-B solution it matches to
    the optimiser's own accuracy."""
    from scipy.optimize import fmin_l_bfgs_b
    for n_vox in (1, 6):
        Z, Y = make_problem(n_vox=n_vox)
        K = len(orc.spm_hrf(1.0, T_R, HRF_DUR, False)[0])
        G, b, yy = orc.hrf_normal_eq(Z, Y, K)
        th, f, h = orc.theta_fit_normal_eq(G, b, yy, T_R, HRF_DUR, (0.6, 1.9))
        th_direct, f_direct = orc.shared_theta_argmin(Z, Y, T_R, HRF_DUR, (0.6, 1.9))
        assert th == pytest.approx(th_direct, abs=2e-7)
        assert f == pytest.approx(f_direct, rel=1e-8)
        assert f == pytest.approx(orc.shared_hrf_cost(th, Z, Y, T_R, HRF_DUR), rel=1e-9)
    Z, Y = make_problem(n_vox=1)
    theta_ref, f_ref, _ = fmin_l_bfgs_b(func=orc.hrf_fit_err, x0=1.9, args=(Z[0], Y[0], T_R, HRF_DUR),
                                        bounds=[(0.6, 1.9)], approx_grad=True, maxiter=999,
                                        pgtol=1.0e-12)
    G, b, yy = orc.hrf_normal_eq(Z, Y, K)
    th, f, _ = orc.theta_fit_normal_eq(G, b, yy, T_R, HRF_DUR, (0.6, 1.9))
    assert th == pytest.approx(float(theta_ref[0]), abs=2e-5)     # L-BFGS-B stops at ~1e-5
    assert f <= float(f_ref) * (1 + 1e-9)                         # and never finds a lower cost
