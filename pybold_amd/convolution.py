"""Host-side helpers mirroring ``pybold/convolution.py``.

Only what the deconvolution hot path needs: the dense Toeplitz builder (the
reference API hands dense ``H`` matrices to ``_loops_deconv``) and its inverse
(recover the taps from such a matrix so the GPU can work matrix-free).  The
FFT-based ``spectral_*`` functions are out of scope: at the
hot path's sizes they equal the causal truncated convolution implemented by
the kernels.
"""
import numpy as np
import torch


def toeplitz_from_kernel(k, dim_in, dim_out=None):
    """``T[i, j] = k[i - j]`` for ``0 <= i - j < len(k)``, shape
    ``(dim_out, dim_in)`` -- same contract as pybold/convolution.py:105-132."""
    k = np.asarray(k, dtype=np.float64)
    if dim_out is None:
        dim_out = dim_in
    T = np.zeros((dim_out, dim_in))
    for m in range(min(len(k), dim_out)):
        n = min(dim_in, dim_out - m)
        if n <= 0:
            break
        idx = np.arange(n)
        T[idx + m, idx] = k[m]
    return T


def kernel_from_toeplitz(H):
    """Taps of a square causal Toeplitz matrix built by :func:`toeplitz_from_kernel`
    (first column, trailing zeros trimmed).  Raises ``ValueError`` if ``H`` is
    not such a matrix: the GPU solver is matrix-free and supports nothing else."""
    H = np.asarray(H, dtype=np.float64)
    if H.ndim != 2:
        raise ValueError("H must be a 2-D causal Toeplitz matrix")
    col = H[:, 0]
    nz = np.nonzero(col)[0]
    K = int(nz[-1]) + 1 if nz.size else 1
    taps = col[:K].copy()
    if not np.array_equal(H, toeplitz_from_kernel(taps, H.shape[1], H.shape[0])):
        raise ValueError("H is not a causal Toeplitz (convolution) matrix; the "
                         "matrix-free GPU solver cannot represent it")
    return taps


def _rows_on_device(x):
    from . import solver
    a = np.asarray(x, dtype=np.float64)
    one_d = a.ndim == 1
    t = torch.from_numpy(np.ascontiguousarray(a.reshape(1, -1) if one_d else a)).to(solver.device())
    return t, one_d


def simple_convolve(k, x, dim_out=None):
    """``out[i] = sum_m k[m] x[i - m]`` truncated to ``dim_out`` samples -- the
    loop-form definition of pybold/convolution.py:135-164 (== ``toeplitz_from_kernel(k,
    len(x), dim_out) @ x``), evaluated by the GPU Toeplitz-product kernel.  1-D ``x``
    like the reference, or a 2-D batch of rows."""
    from . import solver
    t, one_d = _rows_on_device(x)
    out = solver.conv(t, k, dim_out=t.shape[1] if dim_out is None else int(dim_out)).cpu().numpy()
    return out[0] if one_d else out


def simple_retro_convolve(k, x, dim_out=None):
    """Adjoint form ``out[j] = sum_m k[m] x[j + m]`` (pybold/convolution.py:167-196,
    == ``toeplitz_from_kernel(k, dim_out, len(x)).T @ x``)."""
    from . import solver
    t, one_d = _rows_on_device(x)
    out = solver.corr(t, k, dim_in=t.shape[1] if dim_out is None else int(dim_out)).cpu().numpy()
    return out[0] if one_d else out
