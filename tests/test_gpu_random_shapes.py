"""Randomised parity sweep: arbitrary (voxels, scans, taps, lambda, iterations,
warm start, per-problem lambda) through every kernel that accepts the shape,
against the float64 oracle.  Deterministic: the cases are drawn from a seeded NumPy
generator (no third-party test dependency)."""
import numpy as np
import pytest
import torch

from oracle import pybold_oracle as orc

pytestmark = pytest.mark.gpu


def _cases(n=48, seed=20261003):
    rng = np.random.RandomState(seed)
    out = [(1, 1, 1, 0.05, 3, False, False, 0), (37, 700, 40, 1.0, 25, True, True, 1),
           (2, 16, 2, 0.0, 0, False, False, 2), (5, 304, 30, 50.0, 7, True, False, 3)]     # corners
    while len(out) < n:
        out.append((int(rng.randint(1, 38)), int(rng.randint(1, 701)), int(rng.randint(1, 41)),
                    float(rng.choice([0.0, 1e-3, 0.05, 1.0, 50.0])), int(rng.randint(0, 26)),
                    bool(rng.randint(2)), bool(rng.randint(2)), int(rng.randint(2 ** 16))))
    return out


@pytest.mark.parametrize("V,N,K,lbda,n_iter,warm,per_problem,rs", _cases())
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
