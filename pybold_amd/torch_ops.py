"""``torch.ops.pybold_hip``: the solver entry points as registered PyTorch operators
(``pybold_amd/csrc/torch_ops.cpp``, a ``TORCH_LIBRARY`` shim over the C ABI of
``libpybold_hip.so``): tensors in, launches on PyTorch's current stream of the tensors' device,
``RuntimeError`` with the library's message on a bad call.  ``load()`` registers them (once) and
returns the ``torch.ops.pybold_hip`` namespace; the ctypes path (``pybold_amd.solver``) and this one
call the same kernels."""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
OPS_PATH = os.path.join(_HERE, "libpybold_torch_ops.so")
_loaded = False


def load():
    global _loaded
    if not _loaded:
        if not os.path.exists(OPS_PATH):
            raise ImportError("pybold_amd: %s not found; build it with `make -C pybold_amd/csrc` "
                              "(needs PyTorch's headers)" % OPS_PATH)
        torch.ops.load_library(OPS_PATH)
        _loaded = True
    return torch.ops.pybold_hip


def fista_solve(Y, hrf, lbda, step, n_iter, W0=None, want_J=False, stop_mode=0, tol=0.0, wind=6, y_rep=1, flags=0):
    """The same contract as :func:`pybold_amd.solver.fista_solve` through ``torch.ops.pybold_hip.fista_solve``
    (scalar ``lbda``); returns ``(W, J or None, n_done)``."""
    import numpy as np
    from . import solver
    ops = load()
    dev = Y.device
    taps = torch.from_numpy(np.ascontiguousarray(np.asarray(hrf, dtype=np.float64).ravel()))
    P, N = Y.shape[0] * int(y_rep), Y.shape[1]
    if W0 is None:
        W = torch.empty((P, N), dtype=torch.float64, device=dev)
        flags |= solver.PB_FLAG_COLD_START
    else:
        W = W0.clone()
    J = torch.full((P, max(int(n_iter), 1)), float("nan"), dtype=torch.float32, device=dev) if want_J else None
    n_done = torch.empty((P,), dtype=torch.int32, device=dev)
    lbda_vec = None if np.ndim(lbda) == 0 and not torch.is_tensor(lbda) else torch.as_tensor(lbda, dtype=torch.float64).to(dev)
    ops.fista_solve(Y, W, taps, taps.to(dev), float(step), 0.0 if lbda_vec is not None else float(lbda), lbda_vec,
                    solver._betas_on(dev, int(n_iter)), int(n_iter), J, int(stop_mode), float(tol), int(wind), n_done,
                    int(y_rep), int(flags))
    return W, J, n_done
