"""Batched blind deconvolution with ONE HRF DILATION PER VOXEL.

This is the loop the reference runs voxel by voxel under joblib
(``bd``, pybold/bold_signal.py:281-382; fan-out at
examples/icassp_2019/simulation.py:62-72), executed for all voxels at once:

  z-step   ``_loops_deconv`` recurrence for every voxel with its own HRF taps and
           its own step ``1/||A_v^T A_v||_F`` -- one ``pb_fista_solve_pp`` launch
           (``pb_gram_frobenius`` gives the V Lipschitz constants);
  theta-step  per voxel ``argmin_theta 0.5 ||y_v - h(theta) * z_v||^2`` over the
           bounds.  The reference calls SciPy's L-BFGS-B once per voxel with a
           finite-difference gradient (:329-333), every evaluation a pass over the
           voxel's data; a per-voxel Python optimiser cannot be batched.  The cost is
           a quadratic form in the K taps, so ONE pass (``pb_hrf_normal_eq``) yields
           every voxel's ``(G, b, yy)`` and the bounded 1-D problem is solved on those
           numbers by a section search closed with a parabola vertex
           (``pb_theta_fit``).  It returns the minimiser over the whole interval,
           which is L-BFGS-B's answer whenever the cost is unimodal on the bounds, to
           better accuracy than the reference's own early termination (factr = 1e7
           leaves theta accurate to ~1e-5).

The single-voxel ``bd`` keeps SciPy in the loop and is the exact reference
semantic; ``bd`` dispatches 2-D inputs here.
"""
import numpy as np
import torch

from . import solver
from .hrf_model import MAX_DELTA, MIN_DELTA


def fit_dilations(Z, Y, t_r, hrf_dur, bounds, n_refine=3):
    """Per-voxel ``argmin`` of ``hrf_fit_err`` (pybold/bold_signal.py:217-222) over
    ``[lo, hi]``.  ``Z`` float64, ``Y`` float32 (or float64), CUDA ``(V, N)``.  Two launches
    whatever the number of candidates: ``pb_hrf_normal_eq`` (one pass over the data: the cost
    is a quadratic form in the taps) and ``pb_theta_fit`` (section search + parabola vertex
    on those numbers, 64 dilations per refinement).  Returns ``(theta (V,), cost (V,),
    taps (V, K))`` float64 CUDA tensors."""
    K = len(solver.hrf_sample_times(t_r, hrf_dur))
    ne = solver.hrf_normal_eq(Z, Y, K, per_voxel=True)
    return solver.theta_fit(ne, t_r, hrf_dur, bounds[0], n_refine=n_refine)


def fit_dilations_grid(Z, Y, t_r, hrf_dur, bounds, n_grid=17, n_refine=9):
    """The same minimiser by pricing every candidate with a pass over the data
    (``pb_spm_hrf`` + ``pb_hrf_cost_pv``, 17 dilations per launch, bracket / 8 per launch):
    the round-1 form, kept as an independent check of :func:`fit_dilations`."""
    dev = Z.device
    V = Z.shape[0]
    lo, hi = bounds[0]
    a = torch.full((V,), float(lo), dtype=torch.float64, device=dev)
    b = torch.full((V,), float(hi), dtype=torch.float64, device=dev)
    frac = torch.linspace(0.0, 1.0, n_grid, dtype=torch.float64, device=dev)[:, None]
    rows = torch.arange(V, device=dev)
    best_t, best_c = a.clone(), None
    for _ in range(n_refine):
        grid = a[None, :] + (b - a)[None, :] * frac                     # (C, V)
        taps = solver.spm_hrf_batch(grid, t_r, hrf_dur)                 # (C, V, K)
        cost = solver.hrf_cost_pv(Z, Y, taps)                           # (C, V)
        m = cost.argmin(dim=0)
        best_t, best_c = grid[m, rows], cost[m, rows]
        a = grid[(m - 1).clamp(min=0), rows]
        b = grid[(m + 1).clamp(max=n_grid - 1), rows]
    return best_t, best_c


def bd_batch(Y, t_r, lbda=1.0, theta_0=None, z_0=None, hrf_dur=20.0, bounds=None, nb_iter=100,
             early_stopping=False, wind=4, tol=1.0e-12, verbose=0):
    """``bd`` for a batch: ``Y`` float32 CUDA ``(V, N)``.  Same outer structure and
    cost bookkeeping as the reference (z-step with ``nb_iter`` inner iterations as
    at :324, theta-step, normalised ``J``/``r`` and raw ``g`` per outer iteration,
    final z-step).  Returns ``(X, Z, W, H, d)``: float64 CUDA ``(V, N)`` x3, the
    per-voxel HRFs ``(V, K)`` and ``d`` with ``'J'``, ``'r'``, ``'g'`` as
    ``(nb_iter + 2, V)`` arrays and ``'theta'`` ``(V,)``.  ``lbda`` and ``theta_0`` may
    hold one value per voxel.  With ``early_stopping=False`` and ``verbose=0`` the whole loop
    runs without a single host synchronisation."""
    dev = Y.device
    V, n = Y.shape
    if bounds is None:
        bounds = [(MIN_DELTA + 1.0e-1, MAX_DELTA - 1.0e-1)]
    if theta_0 is not None and (torch.is_tensor(theta_0) or np.ndim(theta_0) > 0):
        theta = torch.as_tensor(theta_0, dtype=torch.float64).to(dev).reshape(-1).clone()
        if theta.numel() != V:
            raise ValueError("theta_0 must be a scalar or hold one dilation per voxel "
                             "(%d), got %d" % (V, theta.numel()))
        if V and (float(theta.min()) < MIN_DELTA or float(theta.max()) > MAX_DELTA):
            raise ValueError("theta_0 must lie in [%g, %g]" % (MIN_DELTA, MAX_DELTA))   # hrf_model.py:17-21
    else:
        t0 = MAX_DELTA if theta_0 is None else float(theta_0)
        if t0 < MIN_DELTA or t0 > MAX_DELTA:
            raise ValueError("theta_0 must lie in [%g, %g]" % (MIN_DELTA, MAX_DELTA))
        theta = torch.full((V,), t0, dtype=torch.float64, device=dev)
    taps = solver.spm_hrf_batch(theta, t_r, hrf_dur)                    # (V, K)
    if z_0 is None:
        W = torch.zeros((V, n), dtype=torch.float64, device=dev)
        X = torch.zeros((V, n), dtype=torch.float64, device=dev)
    else:
        Z0 = torch.as_tensor(z_0, dtype=torch.float64).to(dev)
        W = torch.cat([torch.zeros((V, 1), dtype=torch.float64, device=dev),
                       Z0[:, 1:] - Z0[:, :-1]], dim=1)
        X, _ = solver.fista_outputs_pp(W, taps)
    if torch.is_tensor(lbda) or np.ndim(lbda) > 0:          # one lambda per voxel
        lbda = torch.as_tensor(lbda, dtype=torch.float64).to(dev).reshape(-1)
        if lbda.numel() != V:
            raise ValueError("lbda must be a scalar or hold one value per voxel")
    Yd = Y.double()
    r0 = ((X - Yd) ** 2).sum(dim=1)
    g0 = W.abs().sum(dim=1)
    j0 = r0 + lbda * g0
    J, R, G = [torch.ones_like(r0)], [torch.ones_like(r0)], [g0]
    active = torch.ones((V,), dtype=torch.bool, device=dev)

    def z_step(W, taps):
        steps = 1.0 / solver.gram_frobenius_batch(taps, n)
        # without early stopping every voxel advances: iterate in place, no host sync at all
        Wn, _ = solver.fista_solve_pp(Y, taps, steps, lbda, int(nb_iter), W0=W,
                                      stop="loops" if early_stopping else None, tol=tol,
                                      inplace=not early_stopping)
        return Wn

    def record(W, taps, eps, r=None):
        """Cost bookkeeping of :337-342.  ``r`` given: the data term is already known (twice
        the minimum the theta-fit returns, for the same ``W`` and the new taps) and no output
        pass is needed; else ``x`` and ``z`` are formed and returned."""
        X = Z = None
        if r is None:
            X, Z = solver.fista_outputs_pp(W, taps)
            r = ((X - Yd) ** 2).sum(dim=1)
        g = W.abs().sum(dim=1)
        J.append((r + lbda * g) / j0 + eps)
        R.append(r / r0 + eps)
        G.append(g)
        return X, Z

    for idx in range(nb_iter):
        Wn = z_step(W, taps)
        all_active = (not early_stopping) or bool(active.all())
        W = Wn if all_active else torch.where(active[:, None], Wn, W)
        Z = solver.integ_op(W)
        th_new, cost_new, taps_new = fit_dilations(Z, Y, t_r, hrf_dur, bounds)
        if all_active:
            theta, taps = th_new, taps_new
            record(W, taps, 1.0e-30, r=2.0 * cost_new)
        else:
            theta = torch.where(active, th_new, theta)
            taps = torch.where(active[:, None], taps_new, taps)
            record(W, taps, 1.0e-30)
        if verbose > 0:
            print("bd_batch outer %d: median theta %.4f, median J %.6f"
                  % (idx + 1, float(theta.median()), float(J[-1].median())))
        if early_stopping and idx > wind:
            half = int(wind / 2)
            Js = torch.stack(J)
            older, newer = Js[:-half].mean(dim=0), Js[-half:].mean(dim=0)
            active &= ~((newer - older) / newer < tol)
            if not bool(active.any()):
                break
    W = z_step(W, taps)
    X, Z = record(W, taps, 0.0)
    d = {"J": torch.stack(J).cpu().numpy(), "r": torch.stack(R).cpu().numpy(),
         "g": torch.stack(G).cpu().numpy(), "l_alpha": [], "theta": theta.cpu().numpy()}
    return X, Z, W, taps, d
