"""Drop-in counterparts of ``pybold/bold_signal.py`` running on MI355X.

Same function names, argument order, defaults and return tuples as the
reference.  ``y`` may be 1-D (reference behaviour, NumPy in / NumPy float64 out)
or a 2-D ``(V, N)`` batch of voxels solved in one kernel launch (the axis the
reference fans out with joblib, examples/icassp_2019/simulation.py:62-72).  A
CUDA tensor in gives CUDA tensors out (x, z, diff_z stay in HBM).

Deviations from the reference, all deliberate:
  * the fixed-lambda ``deconv`` does not print one line per iteration
    (pybold/bold_signal.py:79-80 does, unconditionally);
  * a 2-D batch ``y`` is stored in float32 in HBM (the iterate and its update stay
    float64 on chip); a 1-D ``y`` is computed in float64 end to end; outputs are
    float64 like the reference's;
  * ``_loops_deconv`` does not overwrite the caller's ``diff_z`` (the reference
    does at :261; every caller rebinds the returned array);
  * ``deconv(lbda=None)`` (noise-driven lambda search, :99-214) estimates the
    noise level with an in-package db3 detail band instead of PyWavelets
    (``utils.mad_daub_noise_est``; parity of that estimate is unpinned -- everything
    else of the branch is pinned against the reference with the estimate injected,
    tests/golden/auto_lbda.npz); it computes in float64 end to end, batches included
    (see ``_deconv_auto_lbda``).
There is no CPU fallback: without the HIP library or a GPU these raise.
"""
import os

import numpy as np
import torch
from scipy.optimize import fmin_l_bfgs_b

from . import solver
from .convolution import kernel_from_toeplitz, toeplitz_from_kernel
from .hrf_model import MAX_DELTA, MIN_DELTA, spm_hrf
from .linear import ConvAndLinear, DiscretInteg
from .utils import gram_frobenius, mad_daub_noise_est, spectral_radius_est


class _Shape:
    """How to hand results back: 1-D or batch, NumPy (reference behaviour) or the
    caller's CUDA tensors (no host round trip)."""

    def __init__(self, one_d, on_device):
        self.one_d, self.on_device = one_d, on_device

    def __bool__(self):             # truthiness = "input was 1-D"
        return self.one_d


def _y_to_device(y):
    """-> (CUDA (V, N), _Shape).  A 2-D batch is stored as float32 (the layout of the
    register-resident kernels).  A 1-D series -- the reference's own call pattern -- stays
    float64 and runs on the all-float64 kernels (``pb_fista_solve_d`` & co.): same
    arithmetic as the reference end to end, which also keeps the finite-difference
    L-BFGS-B of the theta-step on the reference's trajectory."""
    if torch.is_tensor(y):
        one_d = y.dim() == 1
        on_device = y.is_cuda
        t = y.to(device=solver.device(y.device if y.is_cuda else None),
                 dtype=torch.float64 if one_d else torch.float32)
    else:
        a = np.asarray(y)
        one_d, on_device = a.ndim == 1, False
        t = torch.from_numpy(np.ascontiguousarray(
            a, dtype=np.float64 if one_d else np.float32)).to(solver.device())
    return (t.reshape(1, -1) if one_d else t), _Shape(one_d, on_device)


def _host(t, shape):
    """Result in the caller's world: float64 NumPy, or the CUDA tensor itself when
    the input was one."""
    if isinstance(shape, _Shape) and shape.on_device:
        return t[0] if shape.one_d else t
    a = t.cpu().numpy()
    return a[0] if shape else a


# How the batch path of `deconv` reaches the library: "ctypes" (pybold_amd._lib, the default) or "torch_ops"
# (torch.ops.pybold_hip: the TORCH_LIBRARY shim over the same C ABI) -- same kernels, bit-identical results
# (tests/test_torch_ops.py).  Initialised from the environment variable PYBOLD_AMD_DISPATCH.
DISPATCH = os.environ.get("PYBOLD_AMD_DISPATCH", "ctypes")


def deconv(y, t_r, hrf, lbda=None, early_stopping=True, tol=1.0e-6,  # noqa
           wind=6, nb_iter=1000, nb_sub_iter=1000, verbose=0):
    """Deconvolve BOLD signal(s) ``y`` with the HRF ``hrf``
    (pybold/bold_signal.py:13-97): ``min_w 0.5 ||h * cumsum(w) - y||^2 +
    lbda ||w||_1`` by the reference's FISTA-like loop, constant step
    ``1 / (0.9 rho)``.

    Returns ``(x, z, diff_z, J, None, None)`` with ``J`` normalised by ``J[0]``.
    For a 2-D ``y`` every output gains a leading voxel axis and ``J`` is
    ``(V, max iterations run)`` padded with NaN after a voxel's early stop.
    """
    if lbda is None:
        return _deconv_auto_lbda(y, hrf, early_stopping, tol, wind, nb_iter, nb_sub_iter, verbose)
    Y, one_d = _y_to_device(y)
    n = Y.shape[1]
    if Y.shape[0] == 0:
        raise ValueError("deconv: empty voxel batch")
    hrf = np.asarray(hrf, dtype=np.float64)
    H = ConvAndLinear(DiscretInteg(), hrf, dim_in=n, dim_out=n)
    grad_lipschitz_cst = 0.9 * spectral_radius_est(H, (n,))
    step = 1.0 / grad_lipschitz_cst
    if DISPATCH == "torch_ops" and Y.dtype == torch.float32:
        # the same kernels through the registered PyTorch operators (torch.ops.pybold_hip, csrc/torch_ops.cpp)
        from . import torch_ops
        W, J, n_done = torch_ops.fista_solve(Y, hrf, lbda, step, int(nb_iter), want_J=True,
                                             stop_mode=solver._STOP["window" if early_stopping else None],
                                             tol=tol, wind=wind)
        X, Z = torch_ops.load().fista_outputs(W, torch.from_numpy(np.ascontiguousarray(hrf)).to(W.device))
    else:
        W, J, n_done = solver.fista_solve(
            Y, hrf, lbda, step, int(nb_iter), want_J=True,
            stop="window" if early_stopping else None, tol=tol, wind=wind)
        X, Z = solver.fista_outputs(W, hrf)
    n_max = max(int(n_done.max()), 1)
    if one_d.on_device:
        # CUDA in -> CUDA out: the cost trace is normalised on the device and stays there
        # (a (V, n_iter) trace of a large batch is hundreds of MB: no PCIe round trip)
        Jd = J[:, :n_max].to(torch.float64)
        Jd = Jd / (Jd[:, :1] + 1.0e-30)
        if verbose > 0:
            last = Jd.gather(1, (n_done.long() - 1).clamp(min=0)[:, None])
            print("deconv: {0} voxel(s), {1} iteration(s), final normalised cost "
                  "{2:.6f}".format(Y.shape[0], n_max, float(last.nanmean())))
        if one_d:
            return X[0], Z[0], W[0], Jd[0, :int(n_done[0])], None, None
        return X, Z, W, Jd, None, None
    n_done = n_done.cpu().numpy()
    J = J.cpu().numpy().astype(np.float64)[:, :n_max]
    J = J / (J[:, :1] + 1.0e-30)
    if verbose > 0:
        print("deconv: {0} voxel(s), {1} iteration(s), final normalised cost "
              "{2:.6f}".format(Y.shape[0], int(n_done.max()),
                               float(np.nanmean(J[np.arange(len(n_done)), n_done - 1]))))
    if one_d:
        return _host(X, one_d), _host(Z, one_d), _host(W, one_d), J[0, :n_done[0]], None, None
    return _host(X, one_d), _host(Z, one_d), _host(W, one_d), J, None, None


def _deconv_auto_lbda(y, hrf, early_stopping, tol, wind, nb_iter, nb_sub_iter, verbose):
    """``lbda=None`` branch of ``deconv`` (pybold/bold_signal.py:99-214), batched.

    Per voxel: ``sigma`` = db3 MAD noise level (:103); ``alpha_0 = 1``,
    ``lbda = 1/(2 alpha)``, ``mu = 1e-4`` (:104-106); each outer iteration runs a
    warm-started inner solve of at most ``nb_sub_iter`` iterations with the
    window rule (ONE kernel launch for all voxels, per-voxel lambda), then
    ``alpha += mu (||x - y||^2 - N sigma^2)`` (:141-145); a voxel leaves the
    outer loop on the windowed ``alpha`` rule (:164-178) and keeps its iterate;
    a last inner solve (:181-209) ends the run.  Returns ``(x, z, diff_z, J, R, G)``:
    lists of floats for a 1-D ``y`` (as the reference), ``(n_outer, V)`` arrays
    padded with NaN for a batch.
    """
    y_host = y.detach().cpu().numpy() if torch.is_tensor(y) else np.asarray(y, dtype=np.float64)
    sigma = np.atleast_1d(mad_daub_noise_est(y_host))
    Y, one_d = _y_to_device(y)
    if Y.dtype != torch.float64:
        if Y.shape[0] >= 1024:
            solver.warn_once("auto-lambda-f64",
                             "deconv(lbda=None) on %d voxels runs on the all-float64 kernel (one problem per wave, ~3x "
                             "slower than the float32-FIR batch kernels; the LDS kernel beyond 640 scans / 32 taps): "
                             "the decisions of this branch sit on rounding knife edges." % Y.shape[0])
        # Two knife edges (both pinned against the reference: tests/golden/auto_lbda.npz).  (1) The inner window rule
        # compares iterates with gradient points (the aliasing of :65/:72), so its criterion tends to a CONSTANT
        # proportional to lambda instead of 0; the search drives lambda down until that constant crosses `tol`, i.e.
        # "stop the inner solve after 8 iterations or run all of them" flips on the 1e-7 rounding of the float32-FIR
        # kernels (measured: whole trajectories diverge).  (2) alpha -- hence lambda = 1/(2 alpha) -- goes NEGATIVE
        # in the reference (:141-145; it does in its default call on the golden series) and :66 with a negative
        # threshold is discontinuous at 0: sign(u) |th|.  The reference keeps the last sample of diff_z EXACTLY 0
        # (h[0] = 0); a float32 residue there is blown up to |th| per iteration.  So the whole branch runs on the
        # float64 kernels, which restate :66 and keep that zero (fista_exact.h, generic.h), batches included.
        if torch.is_tensor(y) and y.is_cuda:                 # already on the device: widen there
            Y = Y.double() if y.dtype != torch.float64 else torch.atleast_2d(y).contiguous()
        else:
            Y = torch.from_numpy(np.ascontiguousarray(np.atleast_2d(y_host), dtype=np.float64)).to(Y.device)
    V, n = Y.shape
    dev = Y.device
    hrf = np.asarray(hrf, dtype=np.float64)
    H = ConvAndLinear(DiscretInteg(), hrf, dim_in=n, dim_out=n)
    grad_lipschitz_cst = 0.9 * spectral_radius_est(H, (n,))
    step = 1.0 / grad_lipschitz_cst
    stop = "window" if early_stopping else None

    alpha = np.ones(V)
    lbda = 1.0 / (2.0 * alpha)
    mu = 1.0e-4
    active = np.ones(V, dtype=bool)
    l_alpha = []                                   # last `wind` alpha vectors
    J, R, G = [], [], []
    W = torch.zeros((V, n), dtype=torch.float64, device=dev)
    for i in range(nb_iter):
        W_new, _, _ = solver.fista_solve(Y, hrf, lbda, step, int(nb_sub_iter), W0=W, stop=stop,
                                         tol=tol, wind=wind)
        if active.all():
            W = W_new
        else:
            W = torch.where(torch.from_numpy(active).to(dev)[:, None], W_new, W)
        r2, l1 = solver.fista_stats(W, Y, hrf)
        r, g = r2.cpu().numpy(), l1.cpu().numpy()
        grad = r - n * sigma ** 2
        alpha = np.where(active, alpha + mu * grad, alpha)
        lbda = 1.0 / (2.0 * alpha)
        l_alpha.append(alpha.copy())
        if len(l_alpha) > wind:
            l_alpha = l_alpha[1:]
        nan = np.where(active, 0.0, np.nan)
        R.append(r + nan)
        G.append(g + nan)
        J.append(0.5 * r + lbda * g + nan)
        if verbose > 0:
            print("Main loop: iteration {0:03d}, |grad| = {1:0.6f}, lbda = {2:0.6f},".format(
                i + 1, float(np.abs(grad[active]).mean()), float(lbda[active].mean())))
        if early_stopping and i > wind:
            sub_wind_len = int(wind / 2)
            old_iter = np.mean(l_alpha[:-sub_wind_len], axis=0)
            new_iter = np.mean(l_alpha[-sub_wind_len:], axis=0)
            diff = np.abs(new_iter - old_iter) / np.abs(new_iter)
            active &= ~(diff < tol)
            if not active.any():
                break
    W, _, _ = solver.fista_solve(Y, hrf, lbda, step, int(nb_sub_iter), W0=W, stop=stop, tol=tol,
                                 wind=wind)
    X, Z = solver.fista_outputs(W, hrf)
    if one_d:
        keep = [k for k in range(len(J)) if not np.isnan(J[k][0])]
        return (_host(X, one_d), _host(Z, one_d), _host(W, one_d), [float(J[k][0]) for k in keep],
                [float(R[k][0]) for k in keep], [float(G[k][0]) for k in keep])
    return (_host(X, one_d), _host(Z, one_d), _host(W, one_d), np.array(J), np.array(R),
            np.array(G))


def hrf_fit_err(theta, z, y, t_r, hrf_dur):
    """``0.5 || y - h(theta) * z ||^2`` (pybold/bold_signal.py:217-222); the
    convolution and the reduction run on the GPU."""
    theta = float(np.ravel(theta)[0])
    h, _ = spm_hrf(theta, t_r, hrf_dur, False)
    Y, _ = _y_to_device(y)
    Z = torch.from_numpy(np.ascontiguousarray(np.atleast_2d(np.asarray(z, dtype=np.float64)))).to(Y.device)
    return float(solver.hrf_cost(Z, Y, h).sum().item())


class _Tracker:
    """L-BFGS-B callback recording the cost (pybold/utils.py:28-45)."""

    def __init__(self, f, args, verbose=0):
        self.J, self.f, self.args, self.verbose, self.idx = [], f, list(args), verbose, 0

    def __call__(self, x):
        self.idx += 1
        j = self.f(*([x] + self.args))
        if self.verbose > 2:
            print("At iterate {0}, tracked function = {1:.6f}".format(self.idx, j))
        self.J.append(j)


def hrf_estim(z, y, t_r, dur, verbose=0):
    """HRF dilation fit for a known block signal (pybold/bold_signal.py:225-239)."""
    args = (z, y, t_r, dur)
    bounds = [(MIN_DELTA + 1.0e-1, MAX_DELTA - 1.0e-1)]
    f_cost = _Tracker(hrf_fit_err, args, verbose)
    theta, _, _ = fmin_l_bfgs_b(func=hrf_fit_err, x0=MAX_DELTA, args=args, bounds=bounds,
                                approx_grad=True, callback=f_cost, maxiter=99999,
                                pgtol=1.0e-12)
    h, _ = spm_hrf(float(np.ravel(theta)[0]), t_r, dur, False)
    return h, f_cost.J


class _BlindTrace:
    """Normalised cost bookkeeping of ``bd`` (pybold/bold_signal.py:306-313,
    :337-342, :371-380): residual ``r`` and cost ``J`` relative to their initial
    values, sparsity ``g`` raw."""

    def __init__(self, x, y, w, lbda):
        self.y, self.lbda = y, lbda
        self.r0 = float(np.sum(np.square(x - y)))
        g0 = float(np.sum(np.abs(w)))
        self.j0 = self.r0 + lbda * g0
        self.J, self.r, self.g, self.theta = [1.0], [1.0], [g0], []

    def record(self, x, w, eps):
        r = float(np.sum(np.square(x - self.y)))
        g = float(np.sum(np.abs(w)))
        self.J.append((r + self.lbda * g) / self.j0 + eps)
        self.r.append(r / self.r0 + eps)
        self.g.append(g)

    def stalled(self, wind, tol):
        half = int(wind / 2)                          # rule of :350-356
        older, newer = np.mean(self.J[:-half]), np.mean(self.J[-half:])
        return (newer - older) / newer < tol

    def as_dict(self):
        return {'J': np.array(self.J), 'r': np.array(self.r), 'g': np.array(self.g),
                'l_alpha': [], 'theta': np.array(self.theta)}     # 'theta': extra key (per outer iteration)


def _loops_deconv(y, diff_z, H, lbda, nb_iter, early_stopping, wind, tol):
    """Inner FISTA loop of the blind solver (pybold/bold_signal.py:246-278):
    warm start ``diff_z``, step ``1 / ||A^T A||_F`` with ``A = H tril(1)``.
    ``H`` must be the causal Toeplitz matrix of an HRF (what ``bd`` passes);
    the GPU works matrix-free from its first column.  Returns ``diff_z``."""
    taps = kernel_from_toeplitz(H)
    Y, one_d = _y_to_device(y)
    n = Y.shape[1]
    step = 1.0 / gram_frobenius(taps, n)
    W0 = torch.from_numpy(np.ascontiguousarray(
        np.atleast_2d(np.asarray(diff_z, dtype=np.float64)))).to(Y.device)
    W, _, _ = solver.fista_solve(Y, taps, lbda, step, int(nb_iter), W0=W0,
                                 stop="loops" if early_stopping else None, tol=tol, wind=wind)
    return _host(W, one_d)


def bd(y, t_r, lbda=1.0, theta_0=None, z_0=None, hrf_dur=20.0,  # noqa
       bounds=None, nb_iter=100, nb_sub_iter=1000, nb_last_iter=10000,
       print_period=50, early_stopping=False, wind=4, tol=1.0e-12, verbose=0):
    """Blind deconvolution of one voxel (pybold/bold_signal.py:281-382):
    alternate the GPU z-step (:func:`_loops_deconv`, ``nb_iter`` inner
    iterations exactly like the reference's call at :324) with a bounded
    L-BFGS-B fit of the HRF dilation ``theta`` on the GPU-evaluated cost.
    ``nb_sub_iter`` and ``nb_last_iter`` are accepted and unused, as in the
    reference.  Returns ``(x, z, diff_z, h, d)``.

    A 2-D ``y`` (voxels, scans) is solved for all voxels at once with one HRF
    dilation per voxel (``pybold_amd.blind.bd_batch``; the theta-step is then a
    batched section search instead of SciPy's L-BFGS-B, see that module)."""
    if (torch.is_tensor(y) and y.dim() == 2) or (not torch.is_tensor(y) and np.ndim(y) == 2):
        from .blind import bd_batch
        Y, shape = _y_to_device(y)
        X, Z, W, taps, d = bd_batch(Y, t_r, lbda=lbda, theta_0=theta_0, z_0=z_0, hrf_dur=hrf_dur,
                                    bounds=bounds, nb_iter=nb_iter, early_stopping=early_stopping,
                                    wind=wind, tol=tol, verbose=verbose)
        return _host(X, shape), _host(Z, shape), _host(W, shape), _host(taps, shape), d
    y = np.asarray(y).astype(np.float64)
    n = len(y)
    dev = solver.device()
    theta = MAX_DELTA if theta_0 is None else theta_0
    h, _ = spm_hrf(theta, t_r, hrf_dur, False)       # may lie outside `bounds` (:291-292)
    if bounds is None:
        bounds = [(MIN_DELTA + 1.0e-1, MAX_DELTA - 1.0e-1)]

    def outputs(w, taps):
        X, Z = solver.fista_outputs(torch.from_numpy(np.ascontiguousarray(w[None])).to(dev), taps)
        return X.cpu().numpy()[0], Z.cpu().numpy()[0]

    if z_0 is None:
        diff_z = np.zeros(n)
        x = np.zeros(n)
    else:                                            # warm start from a block signal (:299-301)
        z_0 = np.asarray(z_0, dtype=np.float64)
        diff_z = np.concatenate([[0.0], np.diff(z_0)])
        x = solver.conv(torch.from_numpy(z_0[None].copy()).to(dev), h).cpu().numpy()[0]

    trace = _BlindTrace(x, y, diff_z, lbda)
    if verbose > 0:
        print("normalized global cost-function (init): {0:.6f}".format(trace.J[-1]))

    def z_step(w, taps):
        return _loops_deconv(y, w, toeplitz_from_kernel(taps, dim_in=n, dim_out=n), lbda, nb_iter,
                             early_stopping, wind, tol)

    for idx in range(nb_iter):
        diff_z = z_step(diff_z, h)
        z = DiscretInteg().op(diff_z)                 # block signal of this iterate (:326)
        theta, _, _ = fmin_l_bfgs_b(func=hrf_fit_err, x0=theta, args=(z, y, t_r, hrf_dur),
                                    bounds=bounds, approx_grad=True, maxiter=999, pgtol=1.0e-12)
        trace.theta.append(float(np.ravel(theta)[0]))
        h, _ = spm_hrf(float(np.ravel(theta)[0]), t_r, hrf_dur, False)
        x, z = outputs(diff_z, h)
        trace.record(x, diff_z, eps=1.0e-30)
        if (verbose > 0) and ((idx + 1) % print_period == 0):
            print("normalized global cost-function ({0:03d}/{1:03d}): "
                  "{2:.6f}".format(idx + 1, nb_iter, trace.J[-1]))
        if early_stopping and idx > wind and trace.stalled(wind, tol):
            break

    diff_z = z_step(diff_z, h)                        # last solve with the final HRF (:365-369)
    x, z = outputs(diff_z, h)
    trace.record(x, diff_z, eps=0.0)
    d = trace.as_dict()
    return x, z, diff_z, h, d
