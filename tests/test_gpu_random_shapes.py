"""Randomised parity sweep: arbitrary (voxels, scans, taps, lambda, iterations,
warm start, per-problem lambda) through every kernel that accepts the shape,
against the float64 oracle.  Deterministic (fixed hypothesis seed)."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, seed, settings
from hypothesis import strategies as st

from oracle import pybold_oracle as orc

pytestmark = pytest.mark.gpu


@seed(20261003)
@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck))
@given(V=st.integers(1, 37), N=st.integers(1, 700), K=st.integers(1, 40),
       lbda=st.sampled_from([0.0, 1e-3, 0.05, 1.0, 50.0]), n_iter=st.integers(0, 25),
       warm=st.booleans(), per_problem=st.booleans(), rs=st.integers(0, 2 ** 16))
def test_random_shapes_all_kernels(V, N, K, lbda, n_iter, warm, per_problem, rs):
    from pybold_amd import solver
    rng = np.random.RandomState(rs)
    hrf = rng.randn(K) * 0.3
    Y = rng.randn(V, N)
    Y32 = Y.astype(np.float32).astype(np.float64)
    A = orc.toeplitz_from_kernel(hrf, N, N).dot(np.tril(np.ones((N, N))))
    lip = 1.05 * np.linalg.norm(A, 2) ** 2 + 1e-9
    lam = (lbda * (0.5 + rng.rand(V))) if per_problem else lbda
    W0 = (0.01 * rng.randn(V, N)) if warm else None
    ref = orc.fista_batch(Y32, hrf, lam, 1.0 / lip, n_iter, W0=W0)
    Yd = torch.from_numpy(Y.astype(np.float32)).cuda()
    W0d = torch.from_numpy(W0).cuda() if warm else None
    scale = np.abs(ref).max() + 1e-30
    for force in ("fast1", "fast2", "fast2d", "generic"):
        if force != "generic" and not solver.has_fast_path(N, K):
            continue
        W, _, n_done = solver.fista_solve(Yd, hrf, lam, 1.0 / lip, n_iter, W0=W0d, force=force)
        err = np.abs(W.cpu().numpy() - ref).max() / scale
        assert err < 1e-5, (force, V, N, K, lbda, n_iter, warm, per_problem, err)
        assert (n_done.cpu().numpy() == n_iter).all()
    if warm:
        assert torch.equal(W0d, torch.from_numpy(W0).cuda())     # warm start untouched
