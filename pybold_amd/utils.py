"""Host helpers of the hot path (``pybold/utils.py``)."""
import numpy as np
from numpy.linalg import norm as norm_2


def spectral_radius_est(L, x_shape, nb_iter=30, tol=1.0e-6, verbose=False):
    """Power iteration on ``L.adj(L.op(.))`` -- same contract as
    pybold/utils.py:94-109, including the start vector drawn from NumPy's
    global RNG (seed it for reproducible runs).  ``L`` is any object with
    ``.op`` / ``.adj``; float64 on the host, the operator runs on the GPU."""
    v = np.random.randn(*x_shape)            # global RNG, as the reference (:97)
    fused = _fused_radius(L, v, nb_iter, tol)
    if fused is not None:
        rho, n_it = fused
        if verbose and n_it >= nb_iter:
            print("spectral_radius_est: no convergence after %d iterations" % nb_iter)
        return rho
    nv = norm_2(v)
    rho, converged = nv, False
    for _ in range(nb_iter):
        v = L.adj(L.op(v)) / nv                # one step of the power method on L^T L
        rho = norm_2(v)
        converged = abs(rho - nv) < tol
        if converged:
            break
        nv = rho
    if verbose and not converged:
        print("spectral_radius_est: no convergence after %d iterations" % nb_iter)
    return rho


def _fused_radius(L, v, nb_iter, tol):
    """The standard operator of the solvers (square ``ConvAndLinear`` over
    ``DiscretInteg``) runs the whole iteration in one kernel launch instead of
    2 * nb_iter operator calls with host round trips; any other ``.op/.adj`` object
    takes the generic host loop."""
    from . import linear, solver
    if (type(L) is linear.ConvAndLinear and type(L.M) is linear.DiscretInteg
            and L.dim_in == L.dim_out and v.ndim == 1 and v.shape[0] == L.dim_in):
        return solver.spectral_radius(v, L.k, nb_iter, tol)
    return None


def gram_frobenius(hrf, n):
    """``|| A^T A ||_F`` for ``A = toeplitz(hrf) @ tril(ones)``: the Lipschitz
    constant ``_loops_deconv`` uses (pybold/bold_signal.py:249-253).

    ``A`` is lower-triangular Toeplitz with kernel ``c = cumsum(hrf)`` (the step
    response), so ``(A^T A)[j, j+d] = R_d(n-1-j-d)`` with the partial
    autocorrelations ``R_d(T) = sum_{t<=T} c[t] c[t+d]``; the squared Frobenius
    norm is ``sum_d (2 - [d=0]) sum_T R_d(T)^2`` -- O(n^2) instead of the
    reference's two dense n^3 products, same value to rounding."""
    hrf = np.asarray(hrf, dtype=np.float64)
    c = np.cumsum(np.concatenate([hrf, np.zeros(max(0, n - len(hrf)))]))[:n]
    d = np.arange(n)[:, None]
    t = np.arange(n)[None, :]
    cpad = np.concatenate([c, np.zeros(n)])
    r = np.cumsum(c[None, :] * cpad[d + t], axis=1)          # r[d, T] = R_d(T)
    r = np.where(t <= n - 1 - d, r, 0.0)
    weight = np.where(d == 0, 1.0, 2.0)
    return np.sqrt(np.sum(weight * r * r))


def _inf_norm(arr, axis=1):
    """One array of :func:`inf_norm` (pybold/utils.py:118-130) on the GPU: 2-D arrays are
    normalised along ``axis``, 1-D and 3-D arrays as a whole."""
    import torch
    from . import solver
    on_device = torch.is_tensor(arr) and arr.is_cuda
    t = arr if torch.is_tensor(arr) else torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
    t = t.to(device=solver.device(t.device if on_device else None), dtype=torch.float64)
    if t.dim() == 2:
        if axis in (1, -1):
            out = solver.inf_norm_rows(t)
        elif axis in (0, -2):
            out = solver.inf_norm_rows(t.t().contiguous()).t()
        else:
            raise ValueError("axis out of range for a 2-D array")
    elif t.dim() in (1, 3):
        out = solver.inf_norm_rows(t.reshape(1, -1)).reshape(t.shape)
    else:
        raise ValueError("inf-norm normalization only handle 1D, 2D or 3D arrays")
    return out if on_device else out.cpu().numpy()


def inf_norm(arrays, axis=1):
    """Inf-norm normalisation ``x / (max|x| + 1e-12)`` of an array or of each array of a
    list (pybold/utils.py:112-138), computed on the GPU (``pb_inf_norm``): NumPy in ->
    NumPy float64 out, CUDA tensors in -> CUDA tensors out (no host round trip, e.g. on
    the ``(V, N)`` outputs of a batched ``bd``)."""
    if isinstance(arrays, list):
        return [_inf_norm(a, axis=axis) for a in arrays]
    return _inf_norm(arrays, axis=axis)


# db3 decomposition high-pass filter (PyWavelets' Wavelet('db3').dec_hi)
_DB3_DEC_HI = np.array([-0.3326705529509569, 0.8068915093133388, -0.4598775021193313,
                        -0.13501102001039084, 0.08544127388224149, 0.035226291882100656])


def mad(x, c=0.6744):
    """Median absolute deviation / c along the last axis (pybold/utils.py:10-13)."""
    x = np.asarray(x, dtype=np.float64)
    med = np.median(x, axis=-1, keepdims=True)
    return np.median(np.abs(x - med), axis=-1) / c


def mad_daub_noise_est(x, c=0.6744):
    """Noise level from the MAD of the level-1 db3 detail coefficients
    (pybold/utils.py:16-25), for a 1-D signal or every row of a 2-D batch.  The
    reference calls PyWavelets; here the level-1 detail band is computed directly
    (6-tap high-pass, half-sample symmetric extension, stride 2:
    ``cD[k] = sum_j g[j] xe[2k + 1 - j]``, ``(N + 5) // 2`` coefficients).
    Parity of this function is unpinned (PyWavelets is absent from the build image)."""
    x = np.asarray(x, dtype=np.float64)
    F = len(_DB3_DEC_HI)
    n = x.shape[-1]
    if n < F - 1:
        raise ValueError("signal too short for a db3 decomposition")
    xe = np.concatenate([np.flip(x[..., :F - 1], -1), x, np.flip(x, -1)[..., :F - 1]], axis=-1)
    n_out = (n + F - 1) // 2
    cD = np.zeros(x.shape[:-1] + (n_out,))
    base = 1 + (F - 1) + 2 * np.arange(n_out)        # index of x_ext[2k + 1] in xe
    for j in range(F):
        cD += _DB3_DEC_HI[j] * xe[..., base - j]
    return mad(cD, c=c)
