"""CPU oracle: float64 NumPy restatement of pyBOLD's deconvolution hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pybold_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.  The shipped product path is
the HIP library behind ``include/pybold_hip.h``.

Parity status: PINNED.  Every function below is checked in
``tests/test_oracle_golden.py`` against fixtures under ``tests/golden/`` that
were produced by importing the real reference (``/root/reference``) in the
build container with ``tests/golden/make_golden.py`` (``_r2``, ``_r5``).  Two
things could not be executed there and are therefore *unpinned*: the noise
estimate ``mad_daub_noise_est`` (needs ``pywt``; the ``deconv(lbda=None)``
branch that calls it IS pinned since round 5 -- the reference's own loop run
with that one scalar injected, ``tests/golden/auto_lbda.npz``: alpha / lambda
updates incl. negative lambdas, warm-started inner solves, both stop windows,
the default 1000 x 1000 call) and true Numba code generation for
``_loops_deconv`` (its body was executed as plain NumPy).

All ``file:line`` citations are relative to the reference checkout.
"""
import numpy as np
from scipy.optimize import fmin_l_bfgs_b
from scipy.stats import gamma

MIN_DELTA = 0.5   # pybold/hrf_model.py:8
MAX_DELTA = 2.0   # pybold/hrf_model.py:9


# --------------------------------------------------------------------------
# L1: convolution kernels
# --------------------------------------------------------------------------
def toeplitz_from_kernel(k, dim_in, dim_out=None):
    """Dense causal Toeplitz matrix, ``T[i, j] = k[i - j]`` for
    ``0 <= i - j < len(k)`` and 0 elsewhere, shape ``(dim_out, dim_in)``.

    Follows pybold/convolution.py:105-132 (row ``i`` is a window of the
    zero-padded, flipped kernel starting at ``dim_in + len(k) - 1 - i``).
    """
    k = np.asarray(k, dtype=np.float64)
    dim_out = dim_in if dim_out is None else dim_out
    lag = np.arange(dim_out)[:, None] - np.arange(dim_in)[None, :]
    valid = (lag >= 0) & (lag < len(k))
    T = np.zeros((dim_out, dim_in))
    T[valid] = k[lag[valid]]
    return T


def simple_convolve(k, x, dim_out=None):
    """``out[i] = sum_m k[m] x[i-m]`` truncated to ``dim_out`` samples.

    Loop-form definition at pybold/convolution.py:135-164.
    """
    x = np.asarray(x, dtype=np.float64)
    dim_out = len(x) if dim_out is None else dim_out
    return toeplitz_from_kernel(k, len(x), dim_out).dot(x)


def simple_retro_convolve(k, x, dim_out=None):
    """Adjoint of :func:`simple_convolve`: ``out[j] = sum_m k[m] x[j+m]``.

    Loop-form definition at pybold/convolution.py:167-196 (window of the
    kernel padded by ``dim_out - 1`` zeros on both sides, start
    ``dim_out - 1 - i``; the window has ``len(x)`` entries).
    """
    k = np.asarray(k, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    dim_out = len(x) if dim_out is None else dim_out
    padded = np.concatenate([np.zeros(dim_out - 1), k, np.zeros(dim_out - 1)])
    out = np.empty(dim_out)
    for i in range(dim_out):
        s = dim_out - 1 - i
        win = padded[s:s + len(x)]
        out[i] = np.dot(win, x[:len(win)])
    return out


def causal_conv(k, z):
    """Matrix-free form of ``toeplitz_from_kernel(k, N, N) @ z`` for
    ``z`` of shape ``(..., N)``.  Equals ``spectral_convolve`` (FFT with
    zero/mirror/zero padding, pybold/convolution.py:9-30) to ~3e-14 at the
    hot path's sizes; that equality is re-checked against the
    golden ``x`` outputs.
    """
    z = np.asarray(z, dtype=np.float64)
    n = z.shape[-1]
    out = np.zeros_like(z)
    for m in range(min(len(k), n)):
        out[..., m:] += k[m] * z[..., :n - m]
    return out


def causal_corr(k, r):
    """Matrix-free form of ``toeplitz_from_kernel(k, N, N).T @ r``."""
    r = np.asarray(r, dtype=np.float64)
    n = r.shape[-1]
    out = np.zeros_like(r)
    for m in range(min(len(k), n)):
        out[..., :n - m] += k[m] * r[..., m:]
    return out


# --------------------------------------------------------------------------
# L2: linear operators
# --------------------------------------------------------------------------
def integ_op(x):
    """DiscretInteg.op, pybold/linear.py:15-28 (cumulative sum)."""
    return np.cumsum(x, axis=-1)


def integ_adj(x):
    """DiscretInteg.adj, pybold/linear.py:30-43 (reverse cumulative sum)."""
    return np.flip(np.cumsum(np.flip(x, axis=-1), axis=-1), axis=-1)


class DenseH:
    """ConvAndLinear(DiscretInteg(), kernel, dim_in, dim_out) with the dense
    Toeplitz product, pybold/linear.py:49-113: ``op(x) = K cumsum(x)``,
    ``adj(r) = revcumsum(K^T r)``.  Works on ``(N,)`` and ``(V, N)``.
    """

    def __init__(self, kernel, dim_in, dim_out=None):
        self.k = np.asarray(kernel, dtype=np.float64)
        self.K = toeplitz_from_kernel(self.k, dim_in, dim_out)
        self.K_T = self.K.T

    def op(self, x):
        z = integ_op(x)
        return self.K.dot(z) if z.ndim == 1 else z.dot(self.K_T)

    def adj(self, r):
        c = self.K_T.dot(r) if r.ndim == 1 else r.dot(self.K)
        return integ_adj(c)


def spectral_radius_est(H, x0, nb_iter=30, tol=1.0e-6):
    """Power iteration of pybold/utils.py:94-109 from an explicit start
    vector ``x0`` (the reference draws it with ``np.random.randn`` from the
    global RNG at :97).  Returns ``norm(x_new)`` of the last step.
    """
    x_old = np.asarray(x0, dtype=np.float64)
    x_new = x_old
    for _ in range(nb_iter):
        x_new = H.adj(H.op(x_old)) / np.linalg.norm(x_old)
        if abs(np.linalg.norm(x_new) - np.linalg.norm(x_old)) < tol:
            break
        x_old = x_new
    return np.linalg.norm(x_new)


def soft_threshold(u, th):
    """Prox of ``th * |.|_1``, as written at pybold/bold_signal.py:66."""
    return np.sign(u) * np.maximum(np.abs(u) - th, 0)


def momentum_sequence(nb_iter):
    """``beta_k = (t_k - 1) / t_{k+1}``, ``t_0 = 1`` (bold_signal.py:60,68-71)."""
    betas = np.empty(nb_iter)
    t_old = 1.0
    for k in range(nb_iter):
        t = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t_old ** 2))
        betas[k] = (t_old - 1.0) / t
        t_old = t
    return betas


# --------------------------------------------------------------------------
# L3: solvers
# --------------------------------------------------------------------------
def _window_stop(hist, wind, tol):
    """Windowed criterion of pybold/bold_signal.py:86-95 on the list of the
    last ``wind`` stored iterates."""
    half = int(wind / 2)
    old = np.mean(hist[:-half], axis=0)
    new = np.mean(hist[-half:], axis=0)
    return np.linalg.norm(new - old) / (np.linalg.norm(new) + 1.0e-10) < tol


def deconv_fixed_lbda(y, hrf, lbda, nb_iter=1000, early_stopping=True,
                      tol=1.0e-6, wind=6, lipschitz=None, x0_power=None,
                      w0=None, dense=True):
    """Fixed-lambda branch of ``deconv`` (pybold/bold_signal.py:49-97) for one
    voxel ``y (N,)``.

    Recurrence (the in-place ``-=`` at :65 together with the alias at :72
    make the momentum term use the *gradient-step point* of the current
    iteration, not the previous prox point):

        u_k     = w_k - s (H^T H w_k - H^T y)            (:64-65)
        p_k     = soft(u_k, lbda * s)                     (:66)
        w_{k+1} = p_k + beta_k (p_k - prev_k)             (:68-69)
        prev_0 = 0,  prev_k = u_k for k >= 1              (:58, :72)

    ``lipschitz`` = the constant ``0.9 * rho`` of :52; when None it is
    estimated from ``x0_power`` exactly like the reference.  Returns
    ``(x, z, w, J_normalised, n_done, lipschitz)``.
    """
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    H = DenseH(hrf, n, n) if dense else _MatrixFreeH(hrf)
    H_adj_y = H.adj(y)
    if lipschitz is None:
        lipschitz = 0.9 * spectral_radius_est(H, x0_power)
    step = 1.0 / lipschitz
    th = lbda / lipschitz

    w = np.zeros(n) if w0 is None else np.array(w0, dtype=np.float64)
    J, hist = [], []
    t_old = 1.0
    n_done = 0
    x = z = None
    for k in range(nb_iter):
        u = w - step * (H.adj(H.op(w)) - H_adj_y)
        if k > 0 and hist:
            hist[-1] = u            # the stored alias of w_k was overwritten
        prev = u if k > 0 else 0.0
        p = soft_threshold(u, th)
        t = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t_old ** 2))
        w = p + (t_old - 1.0) / t * (p - prev)
        t_old = t

        z = np.cumsum(w)
        x = causal_conv(np.asarray(hrf, dtype=np.float64), z)
        J.append(0.5 * np.sum(np.square(x - y)) + lbda * np.sum(np.abs(w)))
        n_done = k + 1

        hist.append(w)
        if len(hist) > wind:
            hist = hist[1:]
        if early_stopping and k > wind and _window_stop(hist, wind, tol):
            break
    J = np.array(J)
    return x, z, w, J / (J[0] + 1.0e-30), n_done, lipschitz


class _MatrixFreeH:
    def __init__(self, kernel):
        self.k = np.asarray(kernel, dtype=np.float64)

    def op(self, x):
        return causal_conv(self.k, integ_op(x))

    def adj(self, r):
        return integ_adj(causal_corr(self.k, r))


def fista_batch(Y, hrf, lbda, step, nb_iter, W0=None, dense=False):
    """Batched form of the same recurrence with *given* step, no cost trace
    and no early stop: ``Y (V, N)``, ``lbda`` scalar or ``(V,)``.  Residual
    form ``H^T (H w - y)`` (algebraically identical to :64).  Returns
    ``W (V, N)``.  This is what the HIP solver is compared with.
    """
    Y = np.atleast_2d(np.asarray(Y, dtype=np.float64))
    V, n = Y.shape
    H = DenseH(hrf, n, n) if dense else _MatrixFreeH(hrf)
    th = (np.broadcast_to(np.asarray(lbda, dtype=np.float64), (V,)) * step)[:, None]
    W = np.zeros((V, n)) if W0 is None else np.array(W0, dtype=np.float64)
    betas = momentum_sequence(nb_iter)
    for k in range(nb_iter):
        U = W - step * H.adj(H.op(W) - Y)
        P = soft_threshold(U, th)
        W = P + betas[k] * (P - (U if k > 0 else 0.0))
    return W


def fista_backtrack_batch(Y, hrf, lbda, step0, nb_iter, eta=0.5, max_bt=40, W0=None):
    """NOT a restatement of the reference (which has a constant step only, pybold/bold_signal.py:52-53): the float64
    statement of the opt-in backtracking mode (``pb_fista_solve_backtrack_d``, include/pybold_hip.h), its only checker.
    Returns ``(W, step, halvings, margin)``; ``margin`` = the smallest relative distance of an acceptance test from
    equality (a decision closer than rounding to equality cannot be compared across arithmetics)."""
    Y = np.atleast_2d(np.asarray(Y, dtype=np.float64))
    H = _MatrixFreeH(np.asarray(hrf, dtype=np.float64))
    V, n = Y.shape
    lb = np.broadcast_to(np.asarray(lbda, dtype=np.float64), (V,))
    W = np.zeros((V, n)) if W0 is None else np.array(W0, dtype=np.float64)
    steps, halv, margin = np.full(V, float(step0)), np.zeros(V, dtype=np.int64), np.full(V, np.inf)
    betas = momentum_sequence(nb_iter)
    for v in range(V):
        w, s = W[v].copy(), float(step0)
        for k in range(nb_iter):
            r = H.op(w) - Y[v]
            f_w = 0.5 * np.dot(r, r)
            g = H.adj(r)
            for bt in range(max_bt + 1):
                u = w - s * g
                p = soft_threshold(u, lb[v] * s)
                rp = H.op(p) - Y[v]
                f_p = 0.5 * np.dot(rp, rp)
                d = p - w
                q = f_w + np.dot(d, g) + np.dot(d, d) / (2.0 * s)
                margin[v] = min(margin[v], abs(f_p - q) / max(abs(q), 1e-300))
                if f_p <= q or bt >= max_bt:
                    break
                s *= eta
                halv[v] += 1
            w = p + betas[k] * (p - u)
        W[v], steps[v] = w, s
    return W, steps, halv, margin


def fista_outputs(W, hrf):
    """``z = cumsum(w)`` and ``x = h * z`` (bold_signal.py:74-75)."""
    Z = np.cumsum(W, axis=-1)
    return causal_conv(np.asarray(hrf, dtype=np.float64), Z), Z


def gram_lipschitz(hrf, n):
    """``|| A^T A ||_F`` with ``A = toeplitz(h) @ tril(ones)``
    (pybold/bold_signal.py:249-253)."""
    A = toeplitz_from_kernel(hrf, n, n).dot(np.tril(np.ones((n, n)), 0))
    return np.linalg.norm(A.T.dot(A))


def loops_deconv(y, diff_z, H, lbda, nb_iter, early_stopping, wind, tol):
    """``_loops_deconv`` (pybold/bold_signal.py:246-278): dense-Gram form,
    ``s = 1/||A^T A||_F``, stop test evaluated *before* the alias update
    (:267-276) so it compares ``w_{k+1}`` with ``u_k`` (k >= 1) or with 0.
    Does not mutate its arguments (the reference overwrites ``diff_z`` with
    ``u_0`` at :261; callers rebind it).
    """
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    A = np.asarray(H, dtype=np.float64).dot(np.tril(np.ones((n, n)), 0))
    AtA = A.T.dot(A)
    Aty = A.T.dot(y)
    lip = np.linalg.norm(AtA)
    step, th = 1.0 / lip, lbda / lip
    w = np.array(diff_z, dtype=np.float64)
    prev = np.zeros(n)
    t_old = 1.0
    for j in range(nb_iter):
        u = w - step * (AtA.dot(w) - Aty)
        if j > 0:
            prev = u
        p = soft_threshold(u, th)
        t = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t_old ** 2))
        w = p + (t_old - 1.0) / t * (p - prev)
        if early_stopping and j > 2:
            if np.linalg.norm(w - prev) / (np.linalg.norm(w) + 1.0e-10) < tol:
                break
        t_old = t
    return w


def loops_batch(Y, hrf, lbda, step, nb_iter, tol, W0=None):
    """The recurrence and stop rule of :func:`loops_deconv` (pybold/bold_signal.py:259-276) for every row of ``Y``
    with the matrix-free operator and a given ``step`` (any constant step: the rule does not depend on how it was
    chosen).  Returns ``(W, n_done)``: a row that meets ``||w_{k+1} - u_k|| / (||w_{k+1}|| + 1e-10) < tol`` at
    iteration ``j > 2`` keeps its ``w_{k+1}`` and ``n_done = j + 1``; the others run ``nb_iter`` iterations.
    Equal to :func:`loops_deconv` row by row when ``step = 1 / ||A^T A||_F`` (tests/test_oracle_golden.py)."""
    Y = np.atleast_2d(np.asarray(Y, dtype=np.float64))
    V, n = Y.shape
    hrf = np.asarray(hrf, dtype=np.float64)
    W = np.zeros((V, n)) if W0 is None else np.array(W0, dtype=np.float64)
    H = _MatrixFreeH(hrf)
    betas = momentum_sequence(nb_iter)
    th = lbda * step
    active = np.ones(V, dtype=bool)
    n_done = np.full(V, nb_iter, dtype=np.int64)
    out = W.copy()
    for j in range(nb_iter):
        U = W - step * H.adj(H.op(W) - Y)
        P = soft_threshold(U, th)
        W = P + betas[j] * (P - (U if j > 0 else 0.0))
        if j > 2:
            crit = np.linalg.norm(W - U, axis=1) / (np.linalg.norm(W, axis=1) + 1.0e-10)
            fire = active & (crit < tol)
            out[fire] = W[fire]
            n_done[fire] = j + 1
            active &= ~fire
    out[active] = W[active]
    return out, n_done


# --------------------------------------------------------------------------
# HRF model and the blind loop
# --------------------------------------------------------------------------
def spm_hrf(delta, t_r=1.0, dur=60.0, normalized_hrf=True, dt=0.001,
            p_delay=6, undershoot=16.0, p_disp=1.0, u_disp=1.0,
            p_u_ratio=0.167, onset=0.0):
    """Two-gamma SPM HRF with time dilation ``delta``
    (pybold/hrf_model.py:12-39): fine grid ``linspace(0, dur, int(dur/dt))``
    (end point included, :25), gamma pdfs with ``loc = dt/disp`` (:28-30),
    optional max-normalisation (:33-34), decimation by ``int(t_r/dt)`` (:36).
    """
    if delta < MIN_DELTA or delta > MAX_DELTA:
        raise ValueError("delta should belong in [{0}, {1}], got {2}".format(
            MIN_DELTA, MAX_DELTA, delta))
    t = np.linspace(0, dur, int(float(dur) / dt)) - float(onset) / dt
    ts = delta * t
    peak = gamma.pdf(ts, p_delay / p_disp, loc=dt / p_disp)
    under = gamma.pdf(ts, undershoot / u_disp, loc=dt / u_disp)
    hrf = peak - p_u_ratio * under
    if normalized_hrf:
        hrf = hrf / np.max(hrf + 1.0e-30)
    dec = int(t_r / dt)
    return hrf[::dec], t[::dec]


def hrf_fit_err(theta, z, y, t_r, hrf_dur):
    """``0.5 || y - h(theta) * z ||^2`` (pybold/bold_signal.py:217-222)."""
    h, _ = spm_hrf(float(np.ravel(theta)[0]), t_r, hrf_dur, False)
    return 0.5 * np.sum(np.square(y - causal_conv(h, z)))


def hrf_estim(z, y, t_r, dur):
    """pybold/bold_signal.py:225-239 (L-BFGS-B from ``MAX_DELTA``, bounds
    ``[0.6, 1.9]``, finite-difference gradient)."""
    J = []
    args = (z, y, t_r, dur)
    bounds = [(MIN_DELTA + 1.0e-1, MAX_DELTA - 1.0e-1)]
    theta, _, _ = fmin_l_bfgs_b(
        func=hrf_fit_err, x0=MAX_DELTA, args=args, bounds=bounds,
        approx_grad=True, callback=lambda x: J.append(hrf_fit_err(x, *args)),
        maxiter=99999, pgtol=1.0e-12)
    h, _ = spm_hrf(float(theta[0]), t_r, dur, False)
    return h, J


def bd(y, t_r, lbda=1.0, theta_0=None, z_0=None, hrf_dur=20.0, bounds=None,
       nb_iter=100, early_stopping=False, wind=4, tol=1.0e-12):
    """Blind deconvolution loop (pybold/bold_signal.py:281-382): z-step =
    :func:`loops_deconv` with ``nb_iter`` inner iterations (the reference
    passes the outer count, :324), theta-step = bounded L-BFGS-B on
    :func:`hrf_fit_err` (:330-333), normalised cost bookkeeping (:337-342,
    :371-376).  ``nb_sub_iter`` / ``nb_last_iter`` are unused by the reference.
    """
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    theta = MAX_DELTA if theta_0 is None else theta_0
    h, _ = spm_hrf(theta, t_r, hrf_dur, False)
    if z_0 is None:
        w, z, x = np.zeros(n), np.zeros(n), np.zeros(n)
    else:
        z = np.asarray(z_0, dtype=np.float64)
        w = np.append(0, z[1:] - z[:-1])
        x = causal_conv(h, z)
    if bounds is None:
        bounds = [(MIN_DELTA + 1.0e-1, MAX_DELTA - 1.0e-1)]
    r_0 = np.sum(np.square(x - y))
    g_0 = np.sum(np.abs(w))
    j_0 = r_0 + lbda * g_0
    d = {'r': [1.0], 'g': [g_0], 'J': [1.0], 'l_alpha': [], 'theta': []}
    for idx in range(nb_iter):
        H = toeplitz_from_kernel(h, n, n)
        w = loops_deconv(y, w, H, lbda, nb_iter, early_stopping, wind, tol)
        z = np.cumsum(w)
        theta, _, _ = fmin_l_bfgs_b(
            func=hrf_fit_err, x0=theta, args=(z, y, t_r, hrf_dur),
            bounds=bounds, approx_grad=True, maxiter=999, pgtol=1.0e-12)
        d['theta'].append(float(np.ravel(theta)[0]))     # not returned by the reference; for tests
        h, _ = spm_hrf(float(np.ravel(theta)[0]), t_r, hrf_dur, False)
        x = causal_conv(h, z)
        r = np.sum(np.square(x - y))
        g = np.sum(np.abs(w))
        d['J'].append((r + lbda * g) / j_0 + 1.0e-30)
        d['r'].append(r / r_0 + 1.0e-30)
        d['g'].append(g)
        if early_stopping and idx > wind:
            half = int(wind / 2)
            old_j = np.mean(d['J'][:-half])
            new_j = np.mean(d['J'][-half:])
            if (new_j - old_j) / new_j < tol:
                break
    H = toeplitz_from_kernel(h, n, n)
    w = loops_deconv(y, w, H, lbda, nb_iter, early_stopping, wind, tol)
    z = np.cumsum(w)
    x = causal_conv(h, z)
    r = np.sum(np.square(x - y))
    g = np.sum(np.abs(w))
    d['J'].append((r + lbda * g) / j_0)
    d['r'].append(r / r_0)
    d['g'].append(g)
    for key in ('J', 'r', 'g'):
        d[key] = np.array(d[key])
    return x, z, w, h, d


# --------------------------------------------------------------------------
# noise-driven lambda search (deconv with lbda=None)  -- pinned with sigma injected; the db3 estimate itself UNPINNED
# --------------------------------------------------------------------------
# db3 decomposition high-pass filter (PyWavelets' Wavelet('db3').dec_hi)
DB3_DEC_HI = np.array([-0.3326705529509569, 0.8068915093133388, -0.4598775021193313,
                       -0.13501102001039084, 0.08544127388224149, 0.035226291882100656])


def db3_detail_level1(x):
    """Level-1 detail coefficients of the db3 DWT with half-sample symmetric
    extension (PyWavelets' default mode), as ``pywt.wavedec(x, 'db3', level=1)[1]``
    computes them: ``cD[k] = sum_j g[j] xe[2k + 1 - j]``, ``len = (N + 5) // 2``.
    UNPINNED: pywt is absent from the build image (pybold/utils.py:16-25 calls it),
    so this follows the published filter/extension definition only."""
    x = np.asarray(x, dtype=np.float64)
    n, F = len(x), len(DB3_DEC_HI)
    xe = np.concatenate([x[:F - 1][::-1], x, x[::-1][:F - 1]])   # xe[i + F-1] = x_ext[i]
    out = np.empty((n + F - 1) // 2)
    for k in range(len(out)):
        i = 2 * k + 1
        out[k] = sum(DB3_DEC_HI[j] * xe[i - j + F - 1] for j in range(F))
    return out


def mad_daub_noise_est(x, c=0.6744):
    """pybold/utils.py:10-25: MAD of the db3 level-1 detail coefficients / c."""
    cD = db3_detail_level1(x)
    return np.median(np.abs(cD - np.median(cD))) / c


def _inner_fista(w, H, H_adj_y, step, th, nb_sub_iter, early_stopping, wind, tol):
    """Inner loop shared by the lambda search (pybold/bold_signal.py:114-138 and
    :185-209): the fixed-lambda recurrence, momentum restarted, window rule."""
    hist = []
    t_old = 1.0
    for j in range(nb_sub_iter):
        u = w - step * (H.adj(H.op(w)) - H_adj_y)
        if j > 0 and hist:
            hist[-1] = u
        prev = u if j > 0 else 0.0
        p = soft_threshold(u, th)
        t = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t_old ** 2))
        w = p + (t_old - 1.0) / t * (p - prev)
        t_old = t
        hist.append(w)
        if len(hist) > wind:
            hist = hist[1:]
        if early_stopping and j > wind and _window_stop(hist, wind, tol):
            break
    return w


def deconv_auto_lbda(y, hrf, sigma, lipschitz, early_stopping=True, tol=1.0e-6, wind=6,
                     nb_iter=1000, nb_sub_iter=1000):
    """lbda=None branch of ``deconv`` (pybold/bold_signal.py:99-214) for a given
    noise level ``sigma`` (:103) and Lipschitz constant (:52): outer loop
    ``alpha += mu (||x - y||^2 - N sigma^2)``, ``lbda = 1 / (2 alpha)`` (:141-145)
    around warm-started inner solves, windowed stop on ``alpha`` (:164-178), final
    inner solve (:181-209).  Returns ``(x, z, w, J, R, G)`` as lists like the
    reference.  Pinned: 108 runs of the reference's own branch with ``sigma`` injected
    (tests/golden/auto_lbda.npz), <= 1e-10 (one default run that drives alpha through 7e-4:
    over the outer iterations before that)."""
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    hrf = np.asarray(hrf, dtype=np.float64)
    H = _MatrixFreeH(hrf)
    H_adj_y = H.adj(y)
    step = 1.0 / lipschitz
    w = np.zeros(n)
    alpha, mu = 1.0, 1.0e-4
    lbda = 1.0 / (2.0 * alpha)
    l_alpha, J, R, G = [], [], [], []
    for i in range(nb_iter):
        w = _inner_fista(w, H, H_adj_y, step, lbda / lipschitz, nb_sub_iter, early_stopping,
                         wind, tol)
        z = np.cumsum(w)
        x = causal_conv(hrf, z)
        grad = np.sum(np.square(x - y)) - n * sigma ** 2
        alpha += mu * grad
        lbda = 1.0 / (2.0 * alpha)
        l_alpha.append(alpha)
        if len(l_alpha) > wind:
            l_alpha = l_alpha[1:]
        r = np.sum(np.square(x - y))
        g = np.sum(np.abs(w))
        R.append(r)
        G.append(g)
        J.append(0.5 * r + lbda * g)
        if early_stopping and i > wind:
            half = int(wind / 2)
            old = np.mean(l_alpha[:-half])
            new = np.mean(l_alpha[-half:])
            if np.abs(new - old) / np.abs(new) < tol:
                break
    w = _inner_fista(w, H, H_adj_y, step, lbda / lipschitz, nb_sub_iter, early_stopping, wind, tol)
    z = np.cumsum(w)
    x = causal_conv(hrf, z)
    return x, z, w, J, R, G


# --------------------------------------------------------------------------
# Helpers around the path: regularisation path top, inf-norm, shared-HRF step
# --------------------------------------------------------------------------
def lambda_max(Y, hrf):
    """``|| H^T y ||_inf`` per row, ``H = toeplitz(hrf) . cumsum`` (pybold/linear.py:95-113):
    for ``lbda >= lambda_max`` the minimiser of ``0.5||H w - y||^2 + lbda ||w||_1`` is 0."""
    Y = np.atleast_2d(np.asarray(Y, dtype=np.float64))
    return np.array([np.max(np.abs(integ_adj(causal_corr(hrf, y)))) for y in Y])


def inf_norm(arrays, axis=1):
    """pybold/utils.py:112-138: ``x / (max|x| + 1e-12)``; 2-D arrays along ``axis``, 1-D and
    3-D arrays as a whole, lists element-wise."""
    def one(x):
        return x / (np.max(np.abs(x)) + 1.0e-12)

    def arr(a):
        a = np.asarray(a, dtype=np.float64)
        if a.ndim == 2:
            return np.vstack(np.apply_along_axis(one, axis, a))
        if a.ndim in (1, 3):
            return one(a)
        raise ValueError("inf-norm normalization only handle 1D, 2D or 3D arrays")
    if isinstance(arrays, list):
        return [arr(a) for a in arrays]
    return arr(arrays)


def hrf_normal_eq(Z, Y, K):
    """Normal equations of ``hrf_fit_err`` in the K taps, summed over the rows of
    ``(Z, Y)``: with the ``(N, K)`` Toeplitz matrix ``T_z[i, m] = z[i - m]`` (so that
    ``h * z = T_z h``, pybold/convolution.py:105-132 with the roles of signal and kernel
    swapped, as pybold/tests/test_convolution.py:169-176 does): ``G = sum T_z^T T_z``,
    ``b = sum T_z^T y``, ``yy = sum ||y||^2``."""
    Z = np.atleast_2d(np.asarray(Z, dtype=np.float64))
    Y = np.atleast_2d(np.asarray(Y, dtype=np.float64))
    n = Z.shape[1]
    G, b, yy = np.zeros((K, K)), np.zeros(K), 0.0
    for z, y in zip(Z, Y):
        T = toeplitz_from_kernel(z, K, n)          # (n, K): T[i, m] = z[i - m]
        G += T.T.dot(T)
        b += T.T.dot(y)
        yy += float(np.dot(y, y))
    return G, b, yy


def shared_hrf_cost(theta, Z, Y, t_r, hrf_dur):
    """``sum_v 0.5 || y_v - h(theta) * z_v ||^2``: the shared-HRF objective of BASELINE
    config 4; for one voxel it is exactly :func:`hrf_fit_err`."""
    h, _ = spm_hrf(float(np.ravel(theta)[0]), t_r, hrf_dur, False)
    Z, Y = np.atleast_2d(Z), np.atleast_2d(Y)
    return float(sum(0.5 * np.sum(np.square(y - causal_conv(h, z))) for z, y in zip(Z, Y)))


def shared_theta_argmin(Z, Y, t_r, hrf_dur, bounds=(MIN_DELTA + 0.1, MAX_DELTA - 0.1), n_grid=64,
                        n_refine=6):
    """Global minimiser of :func:`shared_hrf_cost` over ``bounds`` by plain section search
    on the DIRECT cost (no normal equations): the checker of the device theta-step."""
    a, b = bounds
    best = a
    for _ in range(n_refine):
        grid = np.linspace(a, b, n_grid)
        f = [shared_hrf_cost(t, Z, Y, t_r, hrf_dur) for t in grid]
        m = int(np.argmin(f))
        best = grid[m]
        a, b = grid[max(m - 1, 0)], grid[min(m + 1, n_grid - 1)]
    return best, shared_hrf_cost(best, Z, Y, t_r, hrf_dur)


def gen_regular_bloc_bold(dur=10, tr=1.0, dur_bloc=30.0, hrf=None, snr=1.0, noise=None):
    """Regular block design of pybold/data.py:10-41: innovation +1/-1 alternating every
    ``int(dur_bloc / tr)`` samples starting at 0 (:14-16), block signal = its cumsum, both
    centred (:20-22), BOLD = ``hrf * blocks`` (:34); ``noise`` (a unit-variance draw) is
    rescaled to the exact SNR in dB as ``add_gaussian_noise`` does (:436-444).
    Returns ``(noisy, clean, ai_s, i_s, scaled noise)``."""
    n = int(dur * 60 / tr)
    i_s = np.zeros(n)
    marks = np.arange(0, n, int(dur_bloc / tr))
    i_s[marks] = -1.0
    i_s[marks[::2]] = 1.0
    ai_s = np.cumsum(i_s)
    ai_s = ai_s - ai_s.mean()
    i_s = i_s - i_s.mean()
    clean = causal_conv(np.asarray(hrf, dtype=np.float64), ai_s)
    if noise is None:
        return clean, clean, ai_s, i_s, np.zeros(n)
    noise = np.asarray(noise, dtype=np.float64)
    ratio = np.linalg.norm(clean) / (np.linalg.norm(noise) + np.finfo(np.float64).eps)
    noise = (1.0 / np.sqrt(10 ** (snr / 10.0))) * ratio * noise
    return clean + noise, clean, ai_s, i_s, noise


def theta_fit_normal_eq(G, b, yy, t_r, hrf_dur, bounds, n_refine=3, n_grid=64):
    """The device theta-step restated on the CPU: minimise the quadratic form
    ``0.5 yy - h^T b + 0.5 h^T G h`` (= :func:`shared_hrf_cost` written with the normal
    equations of :func:`hrf_normal_eq`) over ``bounds`` by section search -- ``n_grid``
    equispaced candidates, bracket = the two cells around the best one -- closed by the
    vertex of the parabola through the best candidate and its neighbours.
    Returns ``(theta, F(theta), h(theta))``."""
    def price(th):
        h = spm_hrf(th, t_r, hrf_dur, False)[0]
        return 0.5 * yy - h.dot(b) + 0.5 * h.dot(G.dot(h))
    a, c = bounds
    best = a
    for r in range(n_refine):
        grid = a + (c - a) * (np.arange(n_grid) / float(n_grid - 1))
        f = np.array([price(t) for t in grid])
        m = int(np.argmin(f))
        il, ir = max(m - 1, 0), min(m + 1, n_grid - 1)
        best = grid[m]
        if r == n_refine - 1 and 0 < m < n_grid - 1:
            den = f[il] - 2.0 * f[m] + f[ir]
            if den > 0.0:
                tv = grid[m] + 0.5 * (grid[m] - grid[il]) * (f[il] - f[ir]) / den
                best = min(max(tv, grid[il]), grid[ir])
        a, c = grid[il], grid[ir]
    return best, price(best), spm_hrf(best, t_r, hrf_dur, False)[0]
