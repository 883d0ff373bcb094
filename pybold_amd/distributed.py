"""Multi-GPU layer: one process per GPU, voxels sharded contiguously over ranks.

Voxels (and (voxel, lambda) problems) are independent, so the fixed-HRF solves
need NO collective: each rank runs :func:`solver.fista_solve` on
its shard.  The only exchange step of the path is the shared-HRF variant of the
blind step, which minimises

    F(theta) = sum_ranks sum_v 0.5 || y_v - h(theta) * z_v ||^2

(the reference fits one theta per voxel with ``hrf_fit_err``,
pybold/bold_signal.py:217-222, :329-333; for a single voxel on a single rank
the functions below reduce to exactly that).  ``F`` is a quadratic form in the K
taps, so each rank makes ONE pass over its shard for the normal equations
``(G, b, yy)`` (``pb_hrf_normal_eq``), ONE all-reduce (SUM) of ``K*K + K + 2``
float64 (~6 kB, latency-bound: xGMI link bandwidth is irrelevant) makes them
global, and every rank runs the identical 1-D search on those numbers
(``pb_theta_fit``, one wave) -- no per-candidate pass, no host round trip: theta,
the new taps and the step constant stay in HBM and feed the next z-step
(``pb_fista_solve_pp`` with shared taps).  ``torch.distributed`` carries the
all-reduce: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
``theta_solver="lbfgsb"`` keeps the reference's optimiser (SciPy L-BFGS-B with a
finite-difference gradient, one all-reduce of two float64 per evaluation) for
the single-voxel equivalence tests.
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

    def allreduce_(self, t):
        """In-place SUM over ranks of a float64 tensor living on the compute device; no
        host synchronisation on the RCCL path (gloo reduces a host copy)."""
        if self.dist is None:
            return t
        if t.device.type == self.device.type:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        else:
            c = t.to(self.device)
            self.dist.all_reduce(c, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(c)
        return t

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


class HipOps:
    """The compute steps of :func:`bd_shared`, each one (or two) launches of the HIP
    library on the current stream; nothing returns to the host.  (The gloo test of the
    multi-rank logic substitutes the CPU oracle for this class.)

    ``fused`` (the default): an outer iteration is z-step + ``normal_eq_msg`` (cumulative sum and
    ``||w||_1`` folded into the pass over the data) + all-reduce + ``theta_step`` (the step constant
    of the next z-step and the normalised cost folded into the fit) -- no other launch."""
    fused = True

    def __init__(self, t_r, hrf_dur, n):
        from . import solver
        self.s, self.t_r, self.hrf_dur, self.n = solver, t_r, hrf_dur, n
        self.work = None

    def hrf(self, theta):                       # (1,) -> (K,)
        return self.s.spm_hrf_batch(theta, self.t_r, self.hrf_dur)[0]

    def z_step(self, Y, taps, lbda, nb_inner, W, step=None, last=True):
        """``last=False``: an intermediate z-step of the outer loop -- the matrix-pipe kernel keeps its
        result also where the iterate is still tiny against the threshold (its accuracy guard would
        send those voxels to the vector kernels; the next, warm-started z-step forgets such errors)."""
        if step is None:
            step = 1.0 / self.s.gram_frobenius_batch(taps.reshape(1, -1), self.n)    # (1,)
        W, _ = self.s.fista_solve_pp(Y, taps, step, lbda, nb_inner, W0=W, inplace=True,
                                     force=None if last else "intermediate")
        return W

    def normal_eq(self, W, Y, K):
        Z = self.s.integ_op(W)
        if self.work is None:
            self.work = torch.empty((2048 * (K * K + K + 2),), dtype=torch.float64, device=Y.device)
        return self.s.hrf_normal_eq(Z, Y, K, work=self.work)

    def normal_eq_msg(self, W, Y, K, out):      # -> out (K*K + K + 2,): normal equations, ||w||_1
        if self.work is None:
            self.work = torch.empty((2048 * (K * K + K + 2),), dtype=torch.float64, device=Y.device)
        return self.s.hrf_normal_eq_w(W, Y, K, work=self.work, out=out)

    def theta_fit(self, ne, bounds):            # -> theta (1,), F(theta) (1,), taps (K,)
        theta, f, taps = self.s.theta_fit(ne, self.t_r, self.hrf_dur, bounds)
        return theta, f, taps[0]

    def theta_step(self, msg, bounds, lbda):    # -> theta, F, taps, step of the next z-step, normalised cost
        return self.s.theta_fit_step(msg, self.t_r, self.hrf_dur, bounds, self.n, lbda)


def _bd_shared_device_loop(Y, t_r, lbda, theta0, hrf_dur, bounds, nb_iter, nb_inner, comm, ops):
    """The outer loop of :func:`bd_shared` with the theta-step on the device: enqueues everything on the current
    stream and returns device tensors only -- ``(W, taps, thetas [nb_iter + 1 tensors (1,)], costs [nb_iter + 1])``.
    No host synchronisation, so the whole loop can be captured into one HIP graph (:class:`BdSharedGraph`)."""
    dev = Y.device
    V, n = Y.shape
    theta = torch.full((1,), theta0, dtype=torch.float64, device=dev)
    taps = ops.hrf(theta)                                        # (K,), stays on the device
    K = taps.numel()
    ne_len = K * K + K + 1
    W = torch.zeros((V, n), dtype=torch.float64, device=dev)
    msg = torch.empty((ne_len + 1,), dtype=torch.float64, device=dev)
    thetas, costs = [theta], []
    fused = bool(getattr(ops, "fused", False))
    step = None                                  # fused: 1 / ||A^T A||_F comes out of the theta step
    for it in range(nb_iter + 1):
        W = ops.z_step(Y, taps, lbda, nb_inner, W, step, last=(it == nb_iter)) if fused else ops.z_step(Y, taps, lbda, nb_inner, W)
        if fused:
            ops.normal_eq_msg(W, Y, K, msg)      # cumulative sum and ||w||_1 inside the one pass
        else:
            msg[:ne_len] = ops.normal_eq(W, Y, K)
            msg[ne_len] = W.abs().sum()
        comm.allreduce_(msg)                     # the ONE collective of the outer iteration
        ne = msg[:ne_len]
        if it < nb_iter and fused:
            theta, f, taps, step, jc = ops.theta_step(msg, bounds[0], lbda)
            thetas.append(theta)
            costs.append(jc)
        elif it < nb_iter:
            theta, f, taps = ops.theta_fit(ne, bounds[0])
            thetas.append(theta)
            costs.append((2.0 * f + lbda * msg[ne_len]) / ne[ne_len - 1])
        else:                                    # last z-step: price the current HRF
            G, b = ne[:K * K].reshape(K, K), ne[K * K:K * K + K]
            f = (0.5 * ne[ne_len - 1] - taps.dot(b) + 0.5 * taps.dot(G.mv(taps))).reshape(1)
            costs.append((2.0 * f + lbda * msg[ne_len]) / ne[ne_len - 1])
    return W, taps, thetas, costs


def _check_bd_shared_args(theta_0, bounds):
    theta0 = MAX_DELTA if theta_0 is None else float(theta_0)
    if theta0 < MIN_DELTA or theta0 > MAX_DELTA:
        raise ValueError("theta_0 must lie in [%g, %g]" % (MIN_DELTA, MAX_DELTA))
    if bounds is None:
        bounds = [(MIN_DELTA + 1.0e-1, MAX_DELTA - 1.0e-1)]
    return theta0, bounds


def bd_shared(Y, t_r, lbda=1.0, theta_0=None, hrf_dur=20.0, bounds=None, nb_iter=20,
              nb_inner=100, comm=None, verbose=0, theta_solver="device", ops=None):
    """Semi-blind deconvolution with ONE HRF dilation shared by all voxels of
    all ranks (BASELINE config 4).  ``Y`` is this rank's shard, float32 CUDA
    ``(V_local, N)`` (may be empty).  Structure of ``bd``
    (pybold/bold_signal.py:281-382): outer loop of z-step (``nb_inner`` iterations
    of the ``_loops_deconv`` recurrence, step ``1/||A^T A||_F``, warm-started) and
    theta-step (see the module docstring), then a last z-step (:365-369).

    Returns ``(W float64 CUDA (V_local, N), h, d)`` with ``d['theta']``,
    ``d['J']`` (global normalised cost per outer iteration) and ``d['evals']``
    (cost evaluations per theta-step: passes over the data for "lbfgsb", 1 for
    "device").  With ``theta_solver="device"`` nothing is copied to the host before
    the loop ends (:class:`BdSharedGraph` submits that loop as ONE HIP graph).
    """
    if theta_solver == "lbfgsb":
        return _bd_shared_lbfgsb(Y, t_r, lbda, theta_0, hrf_dur, bounds, nb_iter, nb_inner, comm,
                                 verbose)
    if theta_solver != "device":
        raise ValueError("theta_solver must be 'device' or 'lbfgsb'")
    comm = comm or Comm()
    ops = ops or HipOps(t_r, hrf_dur, Y.shape[1])
    theta0, bounds = _check_bd_shared_args(theta_0, bounds)
    W, taps, thetas, costs = _bd_shared_device_loop(Y, t_r, lbda, theta0, hrf_dur, bounds, nb_iter, nb_inner, comm, ops)
    d = {"theta": torch.cat(thetas).cpu().numpy(),
         "J": np.concatenate([[1.0], torch.cat(costs).cpu().numpy()]),
         "evals": [1] * nb_iter}
    if verbose > 0 and comm.rank == 0:
        for it in range(nb_iter + 1):
            print("bd_shared outer %d: theta=%.6f J=%.6f" % (it, d["theta"][min(it + 1, nb_iter)], d["J"][it + 1]))
    return W, taps.cpu().numpy(), d


class BdSharedGraph:
    """:func:`bd_shared` (``theta_solver="device"``) as ONE submission: the whole outer loop -- ``nb_iter + 1`` z-steps,
    normal equations, the all-reduce of each outer iteration (RCCL collectives can be captured) and the theta fits,
    ~6 launches per outer iteration -- is captured once into a HIP graph on the data it was built for and replayed by
    :meth:`launch`.  At small shards the loop is bound by its ~130 dependent launches (21 x 3 ... 6), not by the kernels:
    6 250 voxels 6.6 ms eager.  ``Y`` may be refilled in place between launches (same shape).

    Falls back to the eager loop (``self.graph is None``, reason in ``self.fallback``) when capture is not possible:
    a communicator that cannot be captured (gloo), or a capture error."""

    def __init__(self, Y, t_r, lbda=1.0, theta_0=None, hrf_dur=20.0, bounds=None, nb_iter=20, nb_inner=100, comm=None):
        self.comm = comm or Comm()
        self.Y, self.t_r, self.lbda, self.hrf_dur = Y, t_r, lbda, hrf_dur
        self.nb_iter, self.nb_inner = nb_iter, nb_inner
        self.theta0, self.bounds = _check_bd_shared_args(theta_0, bounds)
        self.ops = HipOps(t_r, hrf_dur, Y.shape[1])
        self.graph, self.fallback, self.out = None, None, None
        self._eager()                            # warm-up: every cache (momentum factors, sample times, work buffers) filled
        torch.cuda.synchronize(Y.device)
        capturable = self.comm.dist is None or self.comm.device.type == "cuda"
        if not capturable:
            self.fallback = "communicator is not on the GPU (gloo): eager loop"
            return
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._loop()
            self.graph, self.out = g, out
        except Exception as e:                   # capture refused: keep the eager loop
            self.graph, self.fallback = None, "capture failed: %s" % (str(e).splitlines()[0] if str(e) else type(e).__name__)
            torch.cuda.synchronize(Y.device)
        # Every rank must replay the SAME sequence of collectives: if the capture failed on one rank only, the others
        # would replay captured all-reduces against its eager ones.  The ranks agree here (one MIN all-reduce of a flag,
        # outside any capture: during a capture collectives are only recorded, so no rank is waiting inside one) and fall
        # back together.
        if self.comm.dist is not None and self.comm.world_size > 1:
            ok = torch.tensor([1.0 if self.graph is not None else 0.0], dtype=torch.float64, device=self.comm.device)
            self.comm.dist.all_reduce(ok, op=self.comm.dist.ReduceOp.MIN, group=self.comm.group)
            if float(ok.item()) == 0.0 and self.graph is not None:
                self.graph, self.out = None, None
                self.fallback = "capture failed on another rank: eager loop on every rank"
                self._eager()

    def _loop(self):
        return _bd_shared_device_loop(self.Y, self.t_r, self.lbda, self.theta0, self.hrf_dur, self.bounds, self.nb_iter,
                                      self.nb_inner, self.comm, self.ops)

    def _eager(self):
        self.out = self._loop()

    def launch(self):
        """Enqueue one whole ``bd_shared`` call (one graph launch, or the eager loop) on the current stream."""
        if self.graph is not None:
            self.graph.replay()
        else:
            self._eager()

    def result(self):
        """``(W, h, d)`` of the last :meth:`launch`, as :func:`bd_shared` returns them (synchronises).  Under a graph
        ``W`` is a COPY: the graph's own buffer is overwritten in place by the next :meth:`launch`."""
        W, taps, thetas, costs = self.out
        if self.graph is not None:
            W = W.clone()
        d = {"theta": torch.cat(thetas).cpu().numpy(), "J": np.concatenate([[1.0], torch.cat(costs).cpu().numpy()]),
             "evals": [1] * self.nb_iter}
        return W, taps.cpu().numpy(), d


def _bd_shared_lbfgsb(Y, t_r, lbda, theta_0, hrf_dur, bounds, nb_iter, nb_inner, comm, verbose):
    """``bd_shared`` with the reference's optimiser in the loop: SciPy L-BFGS-B on the host,
    every cost evaluation a pass over the data (``pb_hrf_cost``) + a 16-byte all-reduce."""
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
