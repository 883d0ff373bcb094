"""Multi-GPU layer: one process per GPU, voxels sharded contiguously over ranks.

Voxels (and (voxel, lambda) problems) are independent, so the fixed-HRF solves
need NO collective: each rank runs :func:`solver.fista_solve` on
its shard.  The only exchange step of the path is the shared-HRF variant of the
blind step: one all-reduce (SUM) of two float64 values per evaluation of

    F(theta) = sum_ranks sum_v 0.5 || y_v - h(theta) * z_v ||^2

(the reference fits one theta per voxel with ``hrf_fit_err``,
pybold/bold_signal.py:217-222, :329-333; for a single voxel on a single rank
the functions below reduce to exactly that).  ``torch.distributed`` carries it:
backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  The
message is 16 bytes, i.e. latency-bound; link bandwidth is irrelevant.
"""
import numpy as np
import torch
from scipy.optimize import fmin_l_bfgs_b

from .hrf_model import MAX_DELTA, MIN_DELTA, spm_hrf

FD_EPS = 1.0e-8     # forward-difference step of scipy's approx_grad


def shard_bounds(n, world_size, rank):
    """Contiguous block ``[lo, hi)`` of ``ceil(n / world_size)`` rows for ``rank``
    (last ranks may get fewer, possibly none)."""
    per = -(-int(n) // int(world_size))
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


class Comm:
    """Thin wrapper over ``torch.distributed`` (or nothing, for one process)."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.group = group
        self.device = device
        if self.dist is not None and device is None:
            backend = self.dist.get_backend(group)
            self.device = (torch.device("cuda", torch.cuda.current_device())
                           if backend == "nccl" else torch.device("cpu"))

    @property
    def world_size(self):
        return self.dist.get_world_size(self.group) if self.dist else 1

    @property
    def rank(self):
        return self.dist.get_rank(self.group) if self.dist else 0

    def allreduce_sum(self, values):
        """SUM of a small float64 vector over ranks; returns a NumPy array."""
        v = np.atleast_1d(np.asarray(values, dtype=np.float64))
        if self.dist is None:
            return v.copy()
        t = torch.from_numpy(v.copy()).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()


def shared_theta_fit(local_cost, theta0, bounds, comm, maxiter=999, pgtol=1.0e-12):
    """Bounded L-BFGS-B on the shared dilation ``theta``.

    ``local_cost(thetas)`` returns this rank's ``sum_v f_v(theta)`` for each
    theta of a short list.  Every evaluation prices ``theta`` and the
    forward-difference point together (one kernel launch, ONE all-reduce of two
    float64) and every rank then takes the identical scalar L-BFGS-B step, so
    no broadcast is needed.  Returns ``(theta, F(theta), n_evals)``.
    """
    lo, hi = bounds[0]
    n_evals = [0]

    def fun(x):
        th = float(np.ravel(x)[0])
        h = FD_EPS if th + FD_EPS <= hi else -FD_EPS      # stay inside the bounds
        f = comm.allreduce_sum(local_cost([th, th + h]))
        n_evals[0] += 1
        return float(f[0]), np.array([(f[1] - f[0]) / h])

    x0 = min(max(float(theta0), lo), hi)                   # L-BFGS-B clips x0 itself
    theta, f, _ = fmin_l_bfgs_b(func=fun, x0=np.array([x0]), bounds=bounds, maxiter=maxiter,
                                pgtol=pgtol)
    return float(theta[0]), float(f), n_evals[0]


def bd_shared(Y, t_r, lbda=1.0, theta_0=None, hrf_dur=20.0, bounds=None, nb_iter=20,
              nb_inner=100, comm=None, verbose=0):
    """Semi-blind deconvolution with ONE HRF dilation shared by all voxels of
    all ranks (BASELINE config 4).  ``Y`` is this rank's shard, float32 CUDA
    ``(V_local, N)``.  Structure of ``bd`` (pybold/bold_signal.py:281-382): outer
    loop of z-step (``nb_inner`` iterations of the ``_loops_deconv`` recurrence,
    step ``1/||A^T A||_F``, warm-started) and theta-step (:func:`shared_theta_fit`).

    Returns ``(W float64 CUDA (V_local, N), h, d)`` with ``d['theta']``,
    ``d['J']`` (global normalised cost per outer iteration) and ``d['evals']``.
    """
    from . import solver
    from .utils import gram_frobenius
    comm = comm or Comm()
    dev = Y.device
    V, n = Y.shape
    theta = MAX_DELTA if theta_0 is None else float(theta_0)
    if bounds is None:
        bounds = [(MIN_DELTA + 1.0e-1, MAX_DELTA - 1.0e-1)]
    h, _ = spm_hrf(theta, t_r, hrf_dur, False)
    W = torch.zeros((V, n), dtype=torch.float64, device=dev)
    y2 = comm.allreduce_sum([float((Y.double() ** 2).sum().item())])[0]
    d = {"theta": [theta], "J": [1.0], "evals": []}

    def local_cost_for(Z):
        def local_cost(thetas):
            taps = np.stack([spm_hrf(t, t_r, hrf_dur, False)[0] for t in thetas])
            return solver.hrf_cost(Z, Y, taps).sum(dim=1).cpu().numpy()
        return local_cost

    for it in range(nb_iter + 1):
        step = 1.0 / gram_frobenius(h, n)
        W, _, _ = solver.fista_solve(Y, h, lbda, step, nb_inner, W0=W)
        Z = solver.integ_op(W)
        if it == nb_iter:              # last (long) deconvolution of the reference, :365-369
            f = comm.allreduce_sum(local_cost_for(Z)([theta]))[0]
        else:
            theta, f, evals = shared_theta_fit(local_cost_for(Z), theta, bounds, comm)
            h, _ = spm_hrf(theta, t_r, hrf_dur, False)
            d["theta"].append(theta)
            d["evals"].append(evals)
        g = comm.allreduce_sum([float(W.abs().sum().item())])[0]
        d["J"].append((2.0 * f + lbda * g) / y2)
        if verbose > 0 and comm.rank == 0:
            print("bd_shared outer %d: theta=%.6f J=%.6f" % (it, theta, d["J"][-1]))
    d["J"] = np.array(d["J"])
    d["theta"] = np.array(d["theta"])
    return W, h, d
