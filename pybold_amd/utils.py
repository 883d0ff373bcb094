"""Host helpers of the hot path (``pybold/utils.py``)."""
import numpy as np
from numpy.linalg import norm as norm_2


def spectral_radius_est(L, x_shape, nb_iter=30, tol=1.0e-6, verbose=False):
    """Power iteration on ``L.adj(L.op(.))`` -- same contract as
    pybold/utils.py:94-109, including the start vector drawn from NumPy's
    global RNG (seed it for reproducible runs).  ``L`` is any object with
    ``.op`` / ``.adj``; float64 on the host, the operator runs on the GPU."""
    x_old = np.random.randn(*x_shape)
    x_new = x_old
    stopped = False
    for _ in range(nb_iter):
        x_new = L.adj(L.op(x_old)) / norm_2(x_old)
        if np.abs(norm_2(x_new) - norm_2(x_old)) < tol:
            stopped = True
            break
        x_old = x_new
    if not stopped and verbose:
        print("Spectral radius estimation did not converge")
    return norm_2(x_new)


def gram_frobenius(hrf, n):
    """``|| A^T A ||_F`` for ``A = toeplitz(hrf) @ tril(ones)``: the Lipschitz
    constant ``_loops_deconv`` uses (pybold/bold_signal.py:249-253).  ``A`` is
    itself lower-triangular Toeplitz with kernel ``cumsum(hrf)`` (the step
    response), so it is assembled directly."""
    hrf = np.asarray(hrf, dtype=np.float64)
    step_resp = np.cumsum(np.concatenate([hrf, np.zeros(max(0, n - len(hrf)))]))[:n]
    lag = np.arange(n)[:, None] - np.arange(n)[None, :]
    A = np.where(lag >= 0, step_resp[np.clip(lag, 0, n - 1)], 0.0)
    return np.linalg.norm(A.T.dot(A))


def inf_norm(x):
    """``x / (max|x| + 1e-12)`` (pybold/utils.py:112-115), 1-D helper."""
    x = np.asarray(x, dtype=np.float64)
    return x / (np.max(np.abs(x)) + 1.0e-12)
