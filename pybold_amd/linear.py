"""Linear operators with the reference's duck-typed ``.op(x)`` / ``.adj(x)``
surface (``pybold/linear.py``), executed by the HIP kernels.

Inputs may be 1-D (reference behaviour) or 2-D ``(V, N)`` batches, NumPy
arrays (returned as NumPy float64) or float64 CUDA tensors (returned as
tensors, no host round trip).
"""
import numpy as np
import torch

from . import solver
from .convolution import toeplitz_from_kernel


def _to_dev(x):
    """-> (float64 CUDA tensor (V, N), restore function)"""
    if torch.is_tensor(x):
        one_d = x.dim() == 1
        t = x.to(device=solver.device(x.device if x.is_cuda else None), dtype=torch.float64)
        t = t.reshape(1, -1) if one_d else t
        return t, (lambda r: r.reshape(-1) if one_d else r)
    a = np.asarray(x, dtype=np.float64)
    one_d = a.ndim == 1
    t = torch.from_numpy(np.ascontiguousarray(a.reshape(1, -1) if one_d else a)).to(solver.device())
    return t, (lambda r: r.cpu().numpy().reshape(-1) if one_d else r.cpu().numpy())


class DiscretInteg:
    """Integration operator (pybold/linear.py:9-43): ``op`` = cumulative sum,
    ``adj`` = reverse cumulative sum."""

    def __init__(self):
        pass

    def op(self, x):
        t, back = _to_dev(x)
        return back(solver.integ_op(t))

    def adj(self, x):
        t, back = _to_dev(x)
        return back(solver.integ_adj(t))


class ConvAndLinear:
    """``op(x) = K M.op(x)``, ``adj(r) = M.adj(K^T r)`` with ``K`` the causal
    Toeplitz matrix of ``kernel`` (pybold/linear.py:46-113).  With
    ``M = DiscretInteg()`` both directions are one fused kernel launch."""

    def __init__(self, M, kernel, dim_in, dim_out=None, spectral_conv=False):
        if spectral_conv:
            raise NotImplementedError(
                "spectral_conv=True (FFT path, pybold/linear.py:88-89,108-109) is out "
                "of scope; the solvers never use it")
        self.M = M
        self.k = np.asarray(kernel, dtype=np.float64)
        self.spectral_conv = spectral_conv
        self.dim_in = int(dim_in)
        self.dim_out = int(dim_in if dim_out is None else dim_out)
        self._K = None

    @property
    def K(self):
        """Dense matrix, built on first use (the kernels never need it)."""
        if self._K is None:
            self._K = toeplitz_from_kernel(self.k, self.dim_in, self.dim_out)
        return self._K

    @property
    def K_T(self):
        return self.K.T

    def op(self, x):
        if isinstance(self.M, DiscretInteg):
            t, back = _to_dev(x)
            return back(solver.op_forward(t, self.k, self.dim_out))
        t, back = _to_dev(self.M.op(x))
        return back(solver.conv(t, self.k, self.dim_out))

    def adj(self, x):
        t, back = _to_dev(x)
        if isinstance(self.M, DiscretInteg):
            return back(solver.op_adjoint(t, self.k, self.dim_in))
        return self.M.adj(back(solver.corr(t, self.k, self.dim_in)))
