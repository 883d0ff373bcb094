"""Host-side helpers mirroring ``pybold/convolution.py``.

Only what the deconvolution hot path needs: the dense Toeplitz builder (the
reference API hands dense ``H`` matrices to ``_loops_deconv``) and its inverse
(recover the taps from such a matrix so the GPU can work matrix-free).  The
FFT-based ``spectral_*`` functions are out of scope: at the
hot path's sizes they equal the causal truncated convolution implemented by
the kernels.
"""
import numpy as np


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
