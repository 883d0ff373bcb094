"""Device-level batched API over the C ABI (``include/pybold_hip.h``).

Everything here takes and returns ``torch`` CUDA tensors (PyTorch is used for
device memory and streams only); the arithmetic happens in the hand-written
HIP kernels of ``libpybold_hip.so``.  Rows are problems ("voxels"), the last
dimension is time.
"""
import numpy as np
import torch

from . import _lib
from ._lib import (PB_FLAG_COLD_START, PB_FLAG_NO_CERT, PB_FLAG_FORCE_CERT, PB_FLAG_DIRECT_FIR, PB_FLAG_ONE_STREAM, PB_FLAG_FORCE_FAST, PB_FLAG_FORCE_GENERIC, PB_FLAG_FORCE_PAIR, PB_FLAG_FORCE_WIDE,
                   PB_FLAG_NO_PAIR, PB_FLAG_ONE_LAUNCH, PB_STOP_LOOPS, PB_STOP_NONE, PB_STOP_WINDOW)

_STOP = {None: PB_STOP_NONE, "none": PB_STOP_NONE, "loops": PB_STOP_LOOPS,
         "window": PB_STOP_WINDOW}
# "fast" = a register-resident kernel (library picks single-row or pair form);
# "fast1" / "fast2" pin the single-row / two-problems-per-row form
# "wide" pins the one-problem-per-wave form; "one" = library dispatch restricted to a single launch;
# "fast2d" / "direct": pair form with direct FIRs instead of the 2-parallel fast FIRs;
# "seq": library dispatch without the internal side stream
_FORCE = {None: 0, "generic": PB_FLAG_FORCE_GENERIC, "fast": PB_FLAG_FORCE_FAST,
          "fast1": PB_FLAG_FORCE_FAST | PB_FLAG_NO_PAIR | PB_FLAG_ONE_LAUNCH,
          "fast2": PB_FLAG_FORCE_FAST | PB_FLAG_FORCE_PAIR | PB_FLAG_ONE_LAUNCH,
          "fast2d": PB_FLAG_FORCE_FAST | PB_FLAG_FORCE_PAIR | PB_FLAG_ONE_LAUNCH | PB_FLAG_DIRECT_FIR,
          "direct": PB_FLAG_DIRECT_FIR,
          "wide": PB_FLAG_FORCE_FAST | PB_FLAG_FORCE_WIDE | PB_FLAG_ONE_LAUNCH,
          "one": PB_FLAG_ONE_LAUNCH, "seq": PB_FLAG_ONE_STREAM,
          # window rule: "cert" = no-fire certificate on the pair form + exact re-solve whatever
          # tol * n_iter is; "nocert" = always the full rule (single-row form)
          "cert": PB_FLAG_FORCE_CERT, "cert2": PB_FLAG_FORCE_CERT | PB_FLAG_FORCE_PAIR | PB_FLAG_ONE_LAUNCH,
          "certonly": PB_FLAG_FORCE_CERT | PB_FLAG_FORCE_PAIR | PB_FLAG_ONE_LAUNCH | _lib.PB_FLAG_CERT_NO_RESOLVE,
          "nocert": PB_FLAG_NO_CERT,
          # "valu": library dispatch without the matrix-pipe form; "mfma": everything on it, one launch
          "valu": _lib.PB_FLAG_NO_MFMA, "valuseq": _lib.PB_FLAG_NO_MFMA | PB_FLAG_ONE_STREAM,
          "mfma": PB_FLAG_ONE_LAUNCH | _lib.PB_FLAG_FORCE_MFMA,
          # "mfma2": everything on the matrix-pipe form with each series split over two waves, one launch
          # a partitioned call, measurement aids: only the dense class is solved / only the sparse class
          "path_dense": _lib.PB_FLAG_ONLY_DENSE, "path_sparse": _lib.PB_FLAG_ONLY_SPARSE,
          # the host-side plan of round 4 (no partition on the device)
          "noill": _lib.PB_FLAG_NO_ILL_GUARD,     # partitioned, but ill-conditioned series stay with their class (measurement aid)
          "nopart": _lib.PB_FLAG_NO_PARTITION, "nopartseq": _lib.PB_FLAG_NO_PARTITION | PB_FLAG_ONE_STREAM,
          "mfma2": _lib.PB_FLAG_FORCE_MFMA2, "mfma2only": _lib.PB_FLAG_FORCE_MFMA2 | _lib.PB_FLAG_CERT_NO_RESOLVE,
          "mfma2cert": _lib.PB_FLAG_FORCE_MFMA2 | PB_FLAG_FORCE_CERT,
          "mfma2certonly": _lib.PB_FLAG_FORCE_MFMA2 | PB_FLAG_FORCE_CERT | _lib.PB_FLAG_CERT_NO_RESOLVE,
          # intermediate solve of an outer loop: the matrix-pipe form keeps sparse iterates (no accuracy guard)
          "intermediate": _lib.PB_FLAG_NO_RHO_GUARD,
          "intermediate_noresolve": _lib.PB_FLAG_NO_RHO_GUARD | _lib.PB_FLAG_CERT_NO_RESOLVE,   # measurement aid
          # diagnostic: no re-solve of the problems a guard handed back (they keep n_done = -1)
          "noresolve": _lib.PB_FLAG_CERT_NO_RESOLVE,          # ... through the default dispatch
          "mfmaonly": PB_FLAG_ONE_LAUNCH | _lib.PB_FLAG_FORCE_MFMA | _lib.PB_FLAG_CERT_NO_RESOLVE,
          "mfmacert": PB_FLAG_ONE_LAUNCH | _lib.PB_FLAG_FORCE_MFMA | PB_FLAG_FORCE_CERT,
          "mfmacertonly": PB_FLAG_ONE_LAUNCH | _lib.PB_FLAG_FORCE_MFMA | PB_FLAG_FORCE_CERT | _lib.PB_FLAG_CERT_NO_RESOLVE}


_warned = set()


def warn_once(key, message):
    """One ``RuntimeWarning`` per process and ``key``: the silent performance cliffs of the
    dispatch (LDS kernel, all-float64 kernel) are made loud here, not in a header."""
    if key not in _warned:
        _warned.add(key)
        import warnings
        warnings.warn(message, RuntimeWarning, stacklevel=3)


def _warn_if_slow_kernel(lib, N, K, P, want_J, stop, wind, flags):
    """The any-size LDS kernel is 10-20x slower per voxel-iteration than the register-resident
    forms; say so once when a batch of some size lands on it by dispatch (not by `force`)."""
    if P < 256 or (flags & (PB_FLAG_FORCE_GENERIC | PB_FLAG_FORCE_FAST)):
        return
    form = lib.pb_fista_which_kernel(int(N), int(K), int(P), int(bool(want_J)), _STOP[stop], int(wind))
    if form in (1, 2, 3) and P >= 16384 and not (flags & (_lib.PB_FLAG_NO_MFMA | PB_FLAG_NO_PAIR | PB_FLAG_FORCE_PAIR | PB_FLAG_FORCE_WIDE)):
        # a machine-filling batch on a vector form although a matrix-pipe form exists for neighbouring shapes: say which
        # limit the call ran into (rates at 300 / 600 scans: matrix pipe 4.8 / 2.0e9 voxel-iterations/s, vector forms
        # 1.4-3.0e9, one problem per wave 0.5-0.6e9 -- DESIGN 6)
        why = []
        if N <= 128:
            why = []                                  # (short series: the pair form is the fast one there)
        elif N > 1280:
            why.append("series of %d scans (matrix-pipe forms: 129..1280)" % N)
        elif K > 48 or (K > 33 and _STOP[stop] != PB_STOP_NONE and N <= 224):
            why.append("HRF of %d taps (matrix-pipe forms: <= 48 taps; <= 33 with a stop rule up to 224 scans)" % K)
        elif _STOP[stop] == PB_STOP_LOOPS and want_J:
            why.append("the _loops_deconv rule with a cost trace")
        elif _STOP[stop] == PB_STOP_WINDOW:
            why.append("the window rule with wind=%d / this tolerance (matrix pipe: wind = 6 and tol * n_iter < 0.02)" % wind)
        if why:
            warn_once(("vector", N, K, _STOP[stop], wind, bool(want_J)),
                      "pybold_amd: %d problems (N=%d, K=%d) run on the float32 vector forms (%s), 1.5-4x below the "
                      "matrix-pipe rate: %s" % (P, N, K, KERNEL_NAMES[form].split(" (")[0], "; ".join(why)))
    if form != 0:
        return
    why = []
    if _STOP[stop] == PB_STOP_WINDOW and wind not in (4, 6, 8):
        why.append("wind=%d (register-resident window rule: wind in {4, 6, 8})" % wind)
    if _STOP[stop] == PB_STOP_WINDOW and N > 1216:
        why.append("window rule beyond 1216 scans")
    if K > 48:
        why.append("HRF of %d taps (> 48)" % K)
    if N > 2432:
        why.append("series of %d scans (> 2432)" % N)
    warn_once(("lds", N, K, _STOP[stop], wind),
              "pybold_amd: %d problems (N=%d, K=%d) run on the any-size LDS kernel, 10-20x slower per "
              "voxel-iteration than the register-resident kernels: %s" % (P, N, K, "; ".join(why) or "shape outside the tables"))


def device(dev=None):
    """The HIP device to run on; raises when no GPU is visible (no CPU path)."""
    if not torch.cuda.is_available():
        raise RuntimeError("pybold_amd needs a ROCm GPU (MI355X); none is visible "
                           "and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device()) if dev is None else torch.device(dev)


def momentum_betas(n_iter, t0=1.0):
    """``beta_k = (t_k - 1) / t_{k+1}`` with ``t_{k+1} = (1 + sqrt(1 + 4 t_k^2)) / 2``
    in float64, exactly the scalar sequence of pybold/bold_signal.py:60,68-71."""
    betas = np.empty(n_iter, dtype=np.float64)
    t_old = float(t0)
    for k in range(n_iter):
        t = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t_old ** 2))
        betas[k] = (t_old - 1.0) / t
        t_old = t
    return betas


_beta_cache = {}


def _betas_on(dev, n_iter):
    key = (dev.type, dev.index)
    cur = _beta_cache.get(key)
    if cur is None or cur.numel() < n_iter:
        n = max(n_iter, 1024, 2 * (cur.numel() if cur is not None else 0))
        cur = torch.from_numpy(momentum_betas(n)).to(dev)
        _beta_cache[key] = cur
    return cur


_inited = set()


def _ensure_init(dev):
    """The first use of a device asks the library for its side stream (``pb_init``): created
    before the application's own streams it overlaps best with the caller's
    (include/pybold_hip.h)."""
    if dev.index not in _inited:
        _inited.add(dev.index)
        with torch.cuda.device(dev):
            _lib.load().pb_init()           # optional: a failure only postpones the creation


def _stream_ptr(dev):
    _ensure_init(dev)
    return torch.cuda.current_stream(dev).cuda_stream


def _as_taps(hrf):
    taps = np.ascontiguousarray(np.asarray(hrf, dtype=np.float64).ravel())
    if taps.size < 1:
        raise ValueError("empty HRF")
    return taps


def _rows(t, dtype, name):
    if t.dim() != 2:
        raise ValueError("%s must be 2-D (problems, time), got %s" % (name, tuple(t.shape)))
    if t.dtype != dtype or not t.is_cuda:
        raise TypeError("%s must be a CUDA tensor of %s" % (name, dtype))
    if t.stride(1) != 1:
        t = t.contiguous()
    return t


def _y_rows(Y, fn_f32, fn_f64):
    """Observed series are float32 (batch layout) or float64 (reference arithmetic):
    returns the checked tensor and the matching entry point."""
    if torch.is_tensor(Y) and Y.dtype == torch.float64:
        return _rows(Y, torch.float64, "Y"), fn_f64
    return _rows(Y, torch.float32, "Y"), fn_f32


def _ld(t):
    """Leading dimension in elements (a single-row view may report stride 0)."""
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def has_fast_path(n_scans, n_taps):
    return bool(_lib.load().pb_fista_has_fast_path(int(n_scans), int(n_taps)))


KERNEL_NAMES = {0: "fista_generic_kernel (LDS)", 1: "fista_fast_kernel (register-resident)",
                2: "fista_pair_ffa_kernel (register-resident, two problems per row, fast FIRs)",
                3: "fista_fast_kernel (register-resident, one problem per wave)",
                4: "fista_mfma_kernel (register-resident, 16 problems per wave, both operators on the matrix pipe)",
                5: "fista_mfma2_kernel (register-resident, 16 problems per two waves -- every series split over two "
                   "SIMDs --, both operators on the matrix pipe)",
                6: "fista_mfma4_kernel (register-resident, 16 problems per workgroup of four waves -- every series split over "
                   "the four SIMDs of a compute unit --, both operators on the matrix pipe)"}


def which_kernel(n_scans, n_taps, n_problems, want_J=False, stop=None, wind=6):
    """Name of the kernel :func:`fista_solve` dispatches to for this call shape."""
    return KERNEL_NAMES[_lib.load().pb_fista_which_kernel(
        int(n_scans), int(n_taps), int(n_problems), int(bool(want_J)), _STOP[stop], int(wind))]


def launch_plan(n_scans, n_taps, n_problems, stop=None, wind=6, force=None):
    """``(n_main, main kernel, tail kernel)``: how :func:`fista_solve` lays the problems out --
    the first ``n_main`` (whole rounds of waves) in one launch, the rest in a second one
    (``force="valu"``: the plan without the matrix-pipe form)."""
    import ctypes
    nm, mf, tf = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    _lib.load().pb_fista_plan_ex(int(n_scans), int(n_taps), int(n_problems), _STOP[stop], int(wind),
                                 _FORCE[force], ctypes.byref(nm), ctypes.byref(mf), ctypes.byref(tf))
    return nm.value, (KERNEL_NAMES[mf.value] if nm.value else None), KERNEL_NAMES[tf.value]


def fista_solve(Y, hrf, lbda, step, n_iter, W0=None, want_J=False, stop=None,
                tol=0.0, wind=6, y_rep=1, force=None, lmax=None, dense_ratio=0.0):
    """Run ``n_iter`` iterations of the reference recurrence
    (pybold/bold_signal.py:62-72, :259-276) for every row of ``Y`` in one launch.

    Y     float32 CUDA ``(V, N)`` (the batch layout; register-resident kernels), or
          float64 CUDA ``(V, N)``: float64 end to end (``pb_fista_solve_d``: register-resident
          one-problem-per-wave kernel for N <= 640, K <= 32, else the LDS kernel), the
          reference's arithmetic -- what the 1-D calls of the API use
    hrf   1-D array of K taps (host)
    lbda  scalar, or array/tensor of ``V * y_rep`` per-problem values
    step  ``1 / L``
    W0    optional float64 CUDA ``(P, N)`` warm start (not modified)
    stop  None | "loops" (_loops_deconv rule) | "window" (deconv rule)
    lmax  optional float64 CUDA ``(V,)``: :func:`lambda_max` of every series, if the caller has it (a regularisation
          path, BASELINE config 5, does); otherwise the library computes it when it partitions the call: the problems
          with ``lbda < dense_ratio * lmax`` (default 0.13) run on the matrix-pipe form, the sparse rest on the float32
          vector forms (``pb_fista_solve_ex``; any call shape since round 5)
    Returns ``(W float64 (P, N), J float32 (P, n_iter) or None, n_done int32 (P,))``.
    """
    lib = _lib.load()
    f64 = torch.is_tensor(Y) and Y.dtype == torch.float64
    Y = _rows(Y, torch.float64 if f64 else torch.float32, "Y")
    dev = Y.device
    V, N = Y.shape
    P = V * int(y_rep)
    taps = _as_taps(hrf)
    taps_dev = torch.from_numpy(taps).to(dev)
    cold = 0
    if W0 is None:                     # the kernels start from 0 themselves: no memset, no read
        W = torch.empty((P, N), dtype=torch.float64, device=dev)
        cold = PB_FLAG_COLD_START
    else:
        W = _rows(W0, torch.float64, "W0").clone()
        if W.shape != (P, N):
            raise ValueError("W0 must be %s, got %s" % ((P, N), tuple(W.shape)))
    lbda_dev = None
    lbda_scalar = 0.0
    if not f64 and not (torch.is_tensor(lbda) and lbda.is_cuda) and (np.asarray(lbda) < 0).any():
        # pybold/bold_signal.py:66 with a negative threshold GROWS every entry (its lambda search gets there,
        # :141-145); only the float64 kernels restate that expression, the float32-FIR kernels clamp
        raise ValueError("negative lbda: only the float64 path (float64 Y) reproduces the reference's prox for it")
    if np.ndim(lbda) == 0 and not torch.is_tensor(lbda):
        lbda_scalar = float(lbda)
    else:
        lbda_dev = torch.as_tensor(lbda, dtype=torch.float64).to(dev).contiguous().ravel()
        if lbda_dev.numel() != P:
            raise ValueError("per-problem lbda must have %d entries" % P)
    betas = _betas_on(dev, n_iter)
    J = torch.empty((P, max(n_iter, 1)), dtype=torch.float64 if f64 else torch.float32,
                    device=dev) if want_J else None
    if J is not None:
        J.fill_(float("nan"))
    n_done = torch.empty((P,), dtype=torch.int32, device=dev)
    if f64:
        if force not in (None, "generic", "fast"):
            raise ValueError("float64 y: force must be None, 'fast' (register-resident float64 "
                             "kernel) or 'generic' (LDS kernel)")
        with torch.cuda.device(dev):
            rc = lib.pb_fista_solve_d(
                Y.data_ptr(), _ld(Y), int(y_rep), W.data_ptr(), _ld(W), P, N,
                taps.ctypes.data, taps_dev.data_ptr(), taps.size, float(step), lbda_scalar,
                lbda_dev.data_ptr() if lbda_dev is not None else None,
                betas.data_ptr(), int(n_iter),
                J.data_ptr() if J is not None else None, _ld(J) if J is not None else 0,
                _STOP[stop], float(tol), int(wind), n_done.data_ptr(), _FORCE[force] | cold,
                _stream_ptr(dev))
        _lib.check(rc, "pb_fista_solve_d")
        return W, J, n_done
    # Every float32 call goes through pb_fista_solve_ex with a workspace: where the matrix-pipe form can carry the call
    # (129..310 scans, >= 4 096 problems) the problems are partitioned on the device -- dense class on the matrix
    # pipe, sparse class (lambda near lambda_max) on the float32 vector forms -- so that nothing is solved twice.
    if lmax is not None:
        lmax = lmax.to(device=dev, dtype=torch.float64).contiguous().ravel()
        if lmax.numel() != V:
            raise ValueError("lmax must have one entry per series (%d)" % V)
    work = torch.empty((int(lib.pb_fista_work_len(P, int(y_rep))),), dtype=torch.int32, device=dev)
    _warn_if_slow_kernel(lib, N, taps.size, P, want_J, stop, wind, _FORCE[force])
    with torch.cuda.device(dev):
        rc = lib.pb_fista_solve_ex(
            Y.data_ptr(), _ld(Y), int(y_rep), W.data_ptr(), _ld(W), P, N,
            taps.ctypes.data, taps_dev.data_ptr(), taps.size, float(step), lbda_scalar,
            lbda_dev.data_ptr() if lbda_dev is not None else None,
            betas.data_ptr(), int(n_iter),
            J.data_ptr() if J is not None else None, _ld(J) if J is not None else 0,
            _STOP[stop], float(tol), int(wind), n_done.data_ptr(), _FORCE[force] | cold,
            _stream_ptr(dev), lmax.data_ptr() if lmax is not None else None, float(dense_ratio),
            work.data_ptr(), work.numel())
    _lib.check(rc, "pb_fista_solve_ex")
    return W, J, n_done


def fista_solve_backtrack(Y, hrf, lbda, step0, n_iter, eta=0.5, max_halvings_per_iter=40, W0=None, y_rep=1):
    """OPT-IN EXTRA (never part of a parity run; the reference has a constant step only): the recurrence of
    :func:`fista_solve` with a backtracked step -- ``pb_fista_solve_backtrack_d``.  ``Y`` float64 CUDA ``(V, N)``; ``step0``
    any positive start step (a caller without ``spectral_radius_est``).  Returns ``(W, step, halvings)``: the iterate, the
    final step and the number of step reductions of every problem."""
    lib = _lib.load()
    Y = _rows(Y, torch.float64, "Y")
    dev = Y.device
    V, N = Y.shape
    P = V * int(y_rep)
    taps_dev = torch.from_numpy(_as_taps(hrf)).to(dev)
    cold = 0
    if W0 is None:
        W = torch.empty((P, N), dtype=torch.float64, device=dev)
        cold = PB_FLAG_COLD_START
    else:
        W = _rows(W0, torch.float64, "W0").clone()
    lbda_dev, lbda_scalar = None, 0.0
    if np.ndim(lbda) == 0 and not torch.is_tensor(lbda):
        lbda_scalar = float(lbda)
    else:
        lbda_dev = torch.as_tensor(lbda, dtype=torch.float64).to(dev).contiguous().ravel()
    betas = _betas_on(dev, n_iter)
    step = torch.empty((P,), dtype=torch.float64, device=dev)
    halv = torch.empty((P,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pb_fista_solve_backtrack_d(
            Y.data_ptr(), _ld(Y), int(y_rep), W.data_ptr(), _ld(W), P, N, taps_dev.data_ptr(), taps_dev.numel(),
            float(step0), float(eta), int(max_halvings_per_iter), lbda_scalar,
            lbda_dev.data_ptr() if lbda_dev is not None else None, betas.data_ptr(), int(n_iter), None,
            step.data_ptr(), halv.data_ptr(), cold, _stream_ptr(dev))
    _lib.check(rc, "pb_fista_solve_backtrack_d")
    return W, step, halv


class FistaPlan:
    """Pre-allocated, launch-only form of :func:`fista_solve` for hot loops and
    graph capture: ``run()`` makes ONE ``pb_fista_solve`` call on the current stream of the
    plan's device (a cold start: no memset, the kernels start from 0) and allocates nothing.
    That call enqueues up to four kernels -- whole rounds of waves on the densest form, the
    remainder on others, part of it on the library's side stream (forked from and joined back
    into the current stream; see ``include/pybold_hip.h``)."""

    def __init__(self, Y, hrf, lbda, step, n_iter, y_rep=1, force="fast", W=None, lmax=None, dense_ratio=0.0):
        self.lib = _lib.load()
        self.Y = _rows(Y, torch.float32, "Y")
        self.dev = self.Y.device
        V, self.N = self.Y.shape
        self.P = V * int(y_rep)
        self.y_rep = int(y_rep)
        self.taps = _as_taps(hrf)
        self.taps_dev = torch.from_numpy(self.taps).to(self.dev)
        if W is None:
            W = torch.zeros((self.P, self.N), dtype=torch.float64, device=self.dev)
        elif tuple(W.shape) != (self.P, self.N) or W.stride(1) != 1:
            raise ValueError("W must be (%d, %d) with unit stride along time" % (self.P, self.N))
        self.W = _rows(W, torch.float64, "W")
        self.n_done = torch.empty((self.P,), dtype=torch.int32, device=self.dev)
        self.lbda_dev = None
        self.lbda = 0.0
        if np.ndim(lbda) == 0 and not torch.is_tensor(lbda):
            self.lbda = float(lbda)
        else:
            self.lbda_dev = torch.as_tensor(lbda, dtype=torch.float64).to(self.dev).contiguous().ravel()
        self.step = float(step)
        self.n_iter = int(n_iter)
        self.betas = _betas_on(self.dev, self.n_iter)
        self.flags = _FORCE[force]
        # workspace of the device-side partition (pb_fista_solve_ex); lambda_max of every series if the caller has it
        self.lmax, self.dense_ratio = None, float(dense_ratio)
        if lmax is not None:
            self.lmax = lmax.to(device=self.dev, dtype=torch.float64).contiguous().ravel()
            if self.lmax.numel() != V:
                raise ValueError("lmax must have one entry per series (%d)" % V)
        self.work = torch.empty((int(self.lib.pb_fista_work_len(self.P, self.y_rep)),), dtype=torch.int32, device=self.dev)

    def launch(self, cold=False):
        """Only the solver launch: the iterate continues from its current value, or
        (``cold``) starts from 0 without being read."""
        # the library finds its per-device side stream and wave count through the CURRENT device
        with torch.cuda.device(self.dev):
            rc = self.lib.pb_fista_solve_ex(
                self.Y.data_ptr(), _ld(self.Y), self.y_rep, self.W.data_ptr(), _ld(self.W), self.P,
                self.N, self.taps.ctypes.data, self.taps_dev.data_ptr(), self.taps.size, self.step,
                self.lbda, self.lbda_dev.data_ptr() if self.lbda_dev is not None else None,
                self.betas.data_ptr(), self.n_iter, None, 0, PB_STOP_NONE, 0.0, 0,
                self.n_done.data_ptr(), self.flags | (PB_FLAG_COLD_START if cold else 0), _stream_ptr(self.dev),
                self.lmax.data_ptr() if self.lmax is not None else None, self.dense_ratio,
                self.work.data_ptr(), self.work.numel())
        _lib.check(rc, "pb_fista_solve_ex")

    def run(self):
        """Cold start (what ``deconv`` does, pybold/bold_signal.py:57): solve from w = 0."""
        self.launch(cold=True)
        return self.W


def round_size(n_scans, n_taps, dev=None):
    """Problems in one full round of waves of the densest plain kernel form for this shape
    (two waves on every SIMD of the device), or None when only the LDS kernel applies."""
    n_main, main, tail = launch_plan(n_scans, n_taps, 1 << 22)
    # problems per wave x waves per SIMD of each form
    per_simd = {KERNEL_NAMES[4]: 16, KERNEL_NAMES[5]: 8, KERNEL_NAMES[2]: 16, KERNEL_NAMES[1]: 8,
                KERNEL_NAMES[3]: 2}.get(main if n_main else tail)
    if per_simd is None:
        return None
    return torch.cuda.get_device_properties(device(dev)).multi_processor_count * 4 * per_simd


class HostPipeline:
    """Host buffers in, (optionally) host buffers out: ``y`` (float32, pinned host memory)
    -> ``diff_z``, in chunks of ``chunk`` voxels on separate streams so that the H2D copy of
    chunk c+1 and the D2H copy of chunk c-1 overlap the solve of chunk c.

    ``out_dtype=None``: the result stays in HBM (float64 ``(V, N)``, ``self.W``);
    ``torch.float32`` / ``torch.float64``: it is copied back into pinned host memory
    (``self.out``) through two device slots of ``chunk`` voxels.  ``chunk`` defaults to one
    full round of waves (:func:`round_size`: 16 384 voxels at N <= 304), which leaves the
    solver's own cost unchanged but for the launch boundaries (measured at config-3 size:
    17.25 ms in 7 chunks against 16.76 ms in one call) and one chunk copy exposed at each end.
    Everything is allocated here; ``run`` allocates nothing."""

    def __init__(self, n_voxels, n_scans, hrf, lbda, step, n_iter, chunk=None, out_dtype=torch.float32,
                 dev=None, force=None):
        self.dev = device(dev)
        _ensure_init(self.dev)                      # the library's side stream before ours
        self.V, self.N = int(n_voxels), int(n_scans)
        if chunk is None:
            chunk = round_size(self.N, len(_as_taps(hrf)), self.dev) or 8192
            if self.V < 3 * chunk:                  # too few chunks to hide anything: measured slower
                chunk = self.V                      # than copy, solve, copy on one stream
        self.chunk = max(1, min(int(chunk), max(self.V, 1)))
        self.out_dtype = out_dtype
        self.bounds = [(lo, min(lo + self.chunk, self.V)) for lo in range(0, self.V, self.chunk)]
        self.s_in, self.s_cmp, self.s_out = (torch.cuda.Stream(self.dev) for _ in range(3))
        lam = None if (np.ndim(lbda) == 0 and not torch.is_tensor(lbda)) else \
            torch.as_tensor(lbda, dtype=torch.float64).ravel()
        self.Yd = [torch.empty((self.chunk, self.N), dtype=torch.float32, device=self.dev) for _ in range(2)]
        if out_dtype is None:
            self.W = torch.empty((self.V, self.N), dtype=torch.float64, device=self.dev)
            self.out = None
        else:
            self.W = None
            self.Wd = [torch.empty((self.chunk, self.N), dtype=torch.float64, device=self.dev) for _ in range(2)]
            self.Od = self.Wd if out_dtype == torch.float64 else [
                torch.empty((self.chunk, self.N), dtype=out_dtype, device=self.dev) for _ in range(2)]
            self.out = torch.empty((self.V, self.N), dtype=out_dtype).pin_memory()
        self.plans = []
        for c, (lo, hi) in enumerate(self.bounds):
            b = c % 2
            self.plans.append(FistaPlan(self.Yd[b][:hi - lo], hrf, lbda if lam is None else lam[lo:hi], step, n_iter,
                                        force=force, W=self.W[lo:hi] if self.W is not None else self.Wd[b][:hi - lo]))
        self.ev_in = [torch.cuda.Event() for _ in range(2)]
        self.ev_cmp = [torch.cuda.Event() for _ in range(2)]
        self.ev_out = [torch.cuda.Event() for _ in range(2)]

    def run(self, Yh, sync=True):
        """``Yh``: float32 host tensor ``(V, N)`` (pinned, or the copies are synchronous).
        Returns ``self.out`` (pinned host) or ``self.W`` (HBM); with ``sync=False`` the work is
        only enqueued and ordered before later work on the current stream."""
        if tuple(Yh.shape) != (self.V, self.N) or Yh.dtype != torch.float32 or Yh.is_cuda:
            raise ValueError("Yh must be a float32 host tensor of shape (%d, %d)" % (self.V, self.N))
        cur = torch.cuda.current_stream(self.dev)
        if len(self.bounds) == 1:                   # one chunk: everything on the current stream
            self.Yd[0][:self.V].copy_(Yh, non_blocking=True)
            self.plans[0].run()
            if self.out is not None:
                if self.Od is not self.Wd:
                    self.Od[0][:self.V].copy_(self.Wd[0][:self.V])
                self.out.copy_(self.Od[0][:self.V], non_blocking=True)
            if sync:
                cur.synchronize()
            return self.out if self.out is not None else self.W
        for s in (self.s_in, self.s_cmp, self.s_out):
            s.wait_stream(cur)
        for c, (lo, hi) in enumerate(self.bounds):
            b, n = c % 2, hi - lo
            with torch.cuda.stream(self.s_in):
                if c >= 2:
                    self.s_in.wait_event(self.ev_cmp[b])        # chunk c-2 no longer reads this slot
                self.Yd[b][:n].copy_(Yh[lo:hi], non_blocking=True)
                self.ev_in[b].record(self.s_in)
            with torch.cuda.stream(self.s_cmp):
                self.s_cmp.wait_event(self.ev_in[b])
                if c >= 2 and self.out is not None:
                    self.s_cmp.wait_event(self.ev_out[b])       # chunk c-2 has left this slot
                self.plans[c].run()
                if self.out is not None and self.Od is not self.Wd:
                    self.Od[b][:n].copy_(self.Wd[b][:n])
                self.ev_cmp[b].record(self.s_cmp)
            if self.out is not None:
                with torch.cuda.stream(self.s_out):
                    self.s_out.wait_event(self.ev_cmp[b])
                    self.out[lo:hi].copy_(self.Od[b][:n], non_blocking=True)
                    self.ev_out[b].record(self.s_out)
        for s in (self.s_in, self.s_cmp, self.s_out):
            cur.wait_stream(s)
        if sync:
            cur.synchronize()
        return self.out if self.out is not None else self.W


def fista_outputs(W, hrf):
    """``z = cumsum(w)`` and ``x = hrf * z`` (pybold/bold_signal.py:74-75), float64."""
    lib = _lib.load()
    W = _rows(W, torch.float64, "W")
    dev = W.device
    P, N = W.shape
    taps_dev = torch.from_numpy(_as_taps(hrf)).to(dev)
    Z = torch.empty_like(W)
    X = torch.empty_like(W)
    with torch.cuda.device(dev):
        rc = lib.pb_fista_outputs(W.data_ptr(), _ld(W), P, N, taps_dev.data_ptr(),
                                  taps_dev.numel(), Z.data_ptr(), _ld(Z),
                                  X.data_ptr(), _ld(X), _stream_ptr(dev))
    _lib.check(rc, "pb_fista_outputs")
    return X, Z


def fista_outputs_into(W, taps_dev, Z, X):
    """:func:`fista_outputs` into caller-owned buffers, taps already on the device: one launch, nothing allocated (the
    form a timed loop wants)."""
    lib = _lib.load()
    P, N = W.shape
    with torch.cuda.device(W.device):
        rc = lib.pb_fista_outputs(W.data_ptr(), _ld(W), P, N, taps_dev.data_ptr(), taps_dev.numel(), Z.data_ptr(), _ld(Z),
                                  X.data_ptr(), _ld(X), _stream_ptr(W.device))
    _lib.check(rc, "pb_fista_outputs")


def fista_stats(W, Y, hrf, y_rep=1):
    """Per-problem ``||hrf * cumsum(w) - y||^2`` and ``||w||_1`` (float64 ``(P,)``
    each): the R / G terms of pybold/bold_signal.py:141-157."""
    lib = _lib.load()
    W = _rows(W, torch.float64, "W")
    Y, fn = _y_rows(Y, lib.pb_fista_stats, lib.pb_fista_stats_d)
    dev = W.device
    P, N = W.shape
    taps_dev = torch.from_numpy(_as_taps(hrf)).to(dev)
    r2 = torch.empty((P,), dtype=torch.float64, device=dev)
    l1 = torch.empty((P,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = fn(W.data_ptr(), _ld(W), Y.data_ptr(), _ld(Y), int(y_rep), P, N,
                                taps_dev.data_ptr(), taps_dev.numel(), r2.data_ptr(),
                                l1.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "pb_fista_stats")
    return r2, l1


def spectral_radius(x0, hrf, nb_iter=30, tol=1.0e-6):
    """Power iteration of pybold/utils.py:94-109 for ``H = toeplitz(hrf) . cumsum``
    from the start vector ``x0`` (1-D float64 array), in one kernel launch.
    Returns ``(rho, iterations done)``."""
    lib = _lib.load()
    dev = device()
    x = torch.from_numpy(np.ascontiguousarray(x0, dtype=np.float64).ravel()).to(dev)
    t = torch.from_numpy(_as_taps(hrf)).to(dev)
    out = torch.empty((2,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pb_spectral_radius(x.data_ptr(), x.numel(), t.data_ptr(), t.numel(), int(nb_iter),
                                    float(tol), out.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "pb_spectral_radius")
    rho, n_it = out.cpu().numpy()
    return float(rho), int(n_it)


def _apply(fn_name, X, n_out, taps=None, n_in=None):
    lib = _lib.load()
    X = _rows(X, torch.float64, "x")
    dev = X.device
    V, n_src = X.shape
    out = torch.empty((V, n_out), dtype=torch.float64, device=dev)
    fn = getattr(lib, fn_name)
    with torch.cuda.device(dev):
        if taps is None:
            rc = fn(X.data_ptr(), _ld(X), out.data_ptr(), _ld(out), V, n_src,
                    _stream_ptr(dev))
        else:
            t = torch.from_numpy(_as_taps(taps)).to(dev)
            if fn_name in ("pb_conv", "pb_op_forward"):
                dim_in, dim_out = n_src, n_out
            else:                       # adjoint forms: input lives in the range
                dim_in, dim_out = n_out, n_src
            rc = fn(X.data_ptr(), _ld(X), out.data_ptr(), _ld(out), V, dim_in,
                    dim_out, t.data_ptr(), t.numel(), _stream_ptr(dev))
    _lib.check(rc, fn_name)
    return out


def integ_op(X):
    """Row-wise cumulative sum (DiscretInteg.op, pybold/linear.py:15-28)."""
    return _apply("pb_integ_op", X, X.shape[1])


def integ_adj(X):
    """Row-wise reverse cumulative sum (DiscretInteg.adj, pybold/linear.py:30-43)."""
    return _apply("pb_integ_adj", X, X.shape[1])


def conv(X, taps, dim_out=None):
    """``toeplitz_from_kernel(taps, n_in, dim_out) @ x`` per row (convolution.py:105-132)."""
    return _apply("pb_conv", X, X.shape[1] if dim_out is None else dim_out, taps)


def corr(R, taps, dim_in=None):
    """``toeplitz_from_kernel(taps, dim_in, n_out).T @ r`` per row."""
    return _apply("pb_corr", R, R.shape[1] if dim_in is None else dim_in, taps)


def op_forward(X, taps, dim_out=None):
    """``ConvAndLinear(DiscretInteg(), taps, n_in, dim_out).op`` (linear.py:73-93)."""
    return _apply("pb_op_forward", X, X.shape[1] if dim_out is None else dim_out, taps)


def op_adjoint(R, taps, dim_in=None):
    """``ConvAndLinear(DiscretInteg(), taps, dim_in, n_out).adj`` (linear.py:95-113)."""
    return _apply("pb_op_adjoint", R, R.shape[1] if dim_in is None else dim_in, taps)


def hrf_cost(Z, Y, taps):
    """``0.5 ||y_v - taps_c * z_v||^2`` for every voxel v and candidate HRF c
    (hrf_fit_err, pybold/bold_signal.py:217-222).  ``taps`` is ``(K,)`` or
    ``(C, K)``; returns float64 ``(C, V)``."""
    lib = _lib.load()
    Z = _rows(Z, torch.float64, "Z")
    Y, fn = _y_rows(Y, lib.pb_hrf_cost, lib.pb_hrf_cost_d)
    dev = Z.device
    V, N = Z.shape
    if torch.is_tensor(taps):
        t_dev = taps.to(device=dev, dtype=torch.float64).reshape(-1, taps.shape[-1]).contiguous()
        C, K = t_dev.shape
    else:
        t = np.atleast_2d(np.asarray(taps, dtype=np.float64))
        C, K = t.shape
        t_dev = torch.from_numpy(np.ascontiguousarray(t)).to(dev)
    cost = torch.empty((C, V), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = fn(Z.data_ptr(), _ld(Z), Y.data_ptr(), _ld(Y), V, N,
                             t_dev.data_ptr(), K, C, cost.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "pb_hrf_cost")
    return cost


# ---- per-voxel HRFs (batched blind deconvolution) ---------------------------
def fista_solve_pp(Y, taps, steps, lbda, n_iter, W0=None, stop=None, tol=0.0, force=None,
                   inplace=False):
    """:func:`fista_solve` with one HRF and one step per problem: ``taps`` float64
    CUDA ``(V, K)``, ``steps`` float64 CUDA ``(V,)`` -- or ONE HRF ``(K,)`` and ONE
    step ``(1,)`` living in device memory and shared by every problem (the shared-HRF
    blind step: taps come straight from :func:`theta_fit`).  ``inplace=True`` iterates
    in ``W0`` itself instead of a copy.  Returns ``(W, n_done)``."""
    lib = _lib.load()
    Y = _rows(Y, torch.float32, "Y")
    dev = Y.device
    V, N = Y.shape
    shared = taps.dim() == 1          # ONE HRF / step in device memory for every problem
    if shared:
        if taps.dtype != torch.float64 or not taps.is_cuda:
            raise TypeError("taps must be a float64 CUDA tensor")
        taps = taps.contiguous().reshape(1, -1)
    else:
        taps = _rows(taps, torch.float64, "taps")
        if taps.shape[0] != V:
            raise ValueError("taps must have one row per voxel")
    steps = steps.to(device=dev, dtype=torch.float64).contiguous().ravel()
    if steps.numel() != (1 if shared else V):
        raise ValueError("steps must have one entry per voxel (one in all for a shared HRF)")
    cold = 0
    if W0 is None:
        W = torch.empty((V, N), dtype=torch.float64, device=dev)
        cold = PB_FLAG_COLD_START
    else:
        W = _rows(W0, torch.float64, "W0")
        if not inplace or W.data_ptr() != W0.data_ptr():
            W = W.clone()
        if W.shape != (V, N):
            raise ValueError("W0 must be %s, got %s" % ((V, N), tuple(W.shape)))
    lbda_dev, lbda_scalar = None, 0.0
    if np.ndim(lbda) == 0 and not torch.is_tensor(lbda):
        lbda_scalar = float(lbda)
    else:
        lbda_dev = torch.as_tensor(lbda, dtype=torch.float64).to(dev).contiguous().ravel()
    betas = _betas_on(dev, n_iter)
    n_done = torch.empty((V,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pb_fista_solve_pp(
            Y.data_ptr(), _ld(Y), W.data_ptr(), _ld(W), V, N, taps.data_ptr(),
            0 if shared else _ld(taps), taps.shape[1], steps.data_ptr(), lbda_scalar,
            lbda_dev.data_ptr() if lbda_dev is not None else None, betas.data_ptr(), int(n_iter),
            _STOP[stop], float(tol), n_done.data_ptr(), _FORCE[force] | cold, _stream_ptr(dev))
    _lib.check(rc, "pb_fista_solve_pp")
    return W, n_done


def fista_outputs_pp(W, taps):
    """``z = cumsum(w)``, ``x = taps_v * z`` with one HRF per row."""
    lib = _lib.load()
    W = _rows(W, torch.float64, "W")
    taps = _rows(taps, torch.float64, "taps")
    dev = W.device
    P, N = W.shape
    Z = torch.empty_like(W)
    X = torch.empty_like(W)
    with torch.cuda.device(dev):
        rc = lib.pb_fista_outputs_pp(W.data_ptr(), _ld(W), P, N, taps.data_ptr(), _ld(taps),
                                     taps.shape[1], Z.data_ptr(), _ld(Z), X.data_ptr(), _ld(X),
                                     _stream_ptr(dev))
    _lib.check(rc, "pb_fista_outputs_pp")
    return X, Z


def hrf_cost_pv(Z, Y, taps):
    """``0.5 ||y_v - taps[c, v] * z_v||^2``: ``taps`` float64 CUDA ``(C, V, K)`` ->
    float64 ``(C, V)``."""
    lib = _lib.load()
    Z = _rows(Z, torch.float64, "Z")
    Y, fn = _y_rows(Y, lib.pb_hrf_cost_pv, lib.pb_hrf_cost_pv_d)
    dev = Z.device
    V, N = Z.shape
    if taps.dim() != 3 or taps.shape[1] != V or taps.dtype != torch.float64:
        raise ValueError("taps must be float64 (C, V, K)")
    taps = taps.contiguous()
    C, _, K = taps.shape
    cost = torch.empty((C, V), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = fn(Z.data_ptr(), _ld(Z), Y.data_ptr(), _ld(Y), V, N, taps.data_ptr(),
                                K, C, cost.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "pb_hrf_cost_pv")
    return cost


def gram_frobenius_batch(taps, n):
    """``||A^T A||_F`` per row of ``taps`` (float64 CUDA ``(P, K)``), ``A =
    toeplitz(taps_p, n, n) tril(1)`` (pybold/bold_signal.py:249-253)."""
    lib = _lib.load()
    taps = _rows(taps, torch.float64, "taps")
    dev = taps.device
    P, K = taps.shape
    out = torch.empty((P,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pb_gram_frobenius(taps.data_ptr(), _ld(taps), P, K, int(n), out.data_ptr(),
                                   _stream_ptr(dev))
    _lib.check(rc, "pb_gram_frobenius")
    return out


def spm_hrf_batch(deltas, t_r, dur, dt=0.001, p_delay=6, undershoot=16.0, p_disp=1.0, u_disp=1.0,
                  p_u_ratio=0.167):
    """Un-normalised SPM HRFs for a CUDA vector of dilations (any shape), sampled
    like ``spm_hrf(delta, t_r, dur, normalized_hrf=False)`` (pybold/hrf_model.py:12-39).
    Returns float64 CUDA ``deltas.shape + (K,)``."""
    lib = _lib.load()
    dev = deltas.device
    d = deltas.to(torch.float64).contiguous()
    t_dev = _sample_times_on(dev, t_r, dur, dt)
    M, K = d.numel(), t_dev.numel()
    out = torch.empty((M, K), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pb_spm_hrf(d.data_ptr(), M, t_dev.data_ptr(), K, p_delay / p_disp, dt / p_disp,
                            undershoot / u_disp, dt / u_disp, p_u_ratio, out.data_ptr(),
                            _stream_ptr(dev))
    _lib.check(rc, "pb_spm_hrf")
    return out.reshape(tuple(deltas.shape) + (K,))


# ---- regularisation path, post-processing, theta-step on the device ------------
def lambda_max(Y, hrf):
    """``|| H^T y_v ||_inf`` per voxel (float64 ``(V,)``), ``H = toeplitz(hrf) . cumsum``:
    the smallest ``lbda`` whose ``deconv`` solution is ``diff_z = 0``; a
    regularisation path is ``lbda = c * lambda_max`` with ``c`` in ``(0, 1]``."""
    lib = _lib.load()
    Y, fn = _y_rows(Y, lib.pb_lambda_max, lib.pb_lambda_max_d)
    dev = Y.device
    V, N = Y.shape
    taps_dev = torch.from_numpy(_as_taps(hrf)).to(dev)
    out = torch.empty((V,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = fn(Y.data_ptr(), _ld(Y), V, N, taps_dev.data_ptr(), taps_dev.numel(), out.data_ptr(),
                _stream_ptr(dev))
    _lib.check(rc, "pb_lambda_max")
    return out


def inf_norm_rows(X):
    """``x / (max|x| + 1e-12)`` for every row of a float64 CUDA ``(V, n)`` tensor
    (pybold/utils.py:112-115 applied row-wise)."""
    lib = _lib.load()
    X = _rows(X, torch.float64, "x")
    dev = X.device
    V, n = X.shape
    out = torch.empty_like(X)
    if n == 0:
        return out
    with torch.cuda.device(dev):
        rc = lib.pb_inf_norm(X.data_ptr(), _ld(X), out.data_ptr(), _ld(out), V, n, _stream_ptr(dev))
    _lib.check(rc, "pb_inf_norm")
    return out


def hrf_normal_eq(Z, Y, n_taps, per_voxel=False, work=None):
    """One pass over ``(Z, Y)`` -> the normal equations of ``hrf_fit_err``
    (pybold/bold_signal.py:217-222) in the ``K`` taps: ``G (K, K)``, ``b (K,)``, ``yy``
    packed as ``K*K + K + 1`` float64 -- summed over voxels (``per_voxel=False``, shape
    ``(len,)``; zeros for an empty shard) or one set per voxel (``(V, len)``)."""
    lib = _lib.load()
    Z = _rows(Z, torch.float64, "Z")
    Y, fn = _y_rows(Y, lib.pb_hrf_normal_eq, lib.pb_hrf_normal_eq_d)
    dev = Z.device
    V, N = Z.shape
    K = int(n_taps)
    ne = int(lib.pb_hrf_normal_eq_len(K))
    if per_voxel:
        out = torch.empty((V, ne), dtype=torch.float64, device=dev)
        work_ptr, work_len = None, 0
    else:
        out = torch.empty((ne,), dtype=torch.float64, device=dev)
        if work is None or work.numel() < ne:
            work = torch.empty((2048 * ne,), dtype=torch.float64, device=dev)
        work_ptr, work_len = work.data_ptr(), work.numel()
    with torch.cuda.device(dev):
        rc = fn(Z.data_ptr(), _ld(Z) if V else N, Y.data_ptr(), _ld(Y) if V else N, V, N, K,
                int(bool(per_voxel)), work_ptr, work_len, out.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "pb_hrf_normal_eq")
    return out


def hrf_sample_times(t_r, dur, dt=0.001):
    """Sample times of ``spm_hrf(., t_r, dur)`` (pybold/hrf_model.py:25: the fine
    ``linspace`` grid includes its end point, then decimation by ``int(t_r / dt)``)."""
    return np.ascontiguousarray(np.linspace(0, dur, int(float(dur) / dt))[::int(t_r / dt)])


_times_cache = {}


def _sample_times_on(dev, t_r, dur, dt=0.001):
    """The HRF sample times as a device tensor, uploaded once per (device, t_r, dur): a
    pageable host-to-device copy would otherwise synchronise every call."""
    key = (dev.type, dev.index, float(t_r), float(dur), float(dt))
    t = _times_cache.get(key)
    if t is None:
        t = torch.from_numpy(hrf_sample_times(t_r, dur, dt)).to(dev)
        _times_cache[key] = t
    return t


def theta_fit(ne, t_r, dur, bounds, n_refine=3, dt=0.001, p_delay=6, undershoot=16.0, p_disp=1.0,
              u_disp=1.0, p_u_ratio=0.167):
    """``argmin_theta 0.5 ||y - h(theta) * z||^2`` over ``bounds = (lo, hi)`` from normal
    equations (:func:`hrf_normal_eq`), entirely on the device: ``ne`` is ``(len,)`` (one
    shared set) or ``(M, len)``.  Returns ``(theta (M,), cost (M,), taps (M, K))`` float64
    CUDA tensors; nothing is copied to the host."""
    lib = _lib.load()
    one = ne.dim() == 1
    ne2 = _rows(ne.reshape(1, -1) if one else ne, torch.float64, "ne")
    dev = ne2.device
    t_dev = _sample_times_on(dev, t_r, dur, dt)
    K = t_dev.numel()
    if ne2.shape[1] != int(lib.pb_hrf_normal_eq_len(K)):
        raise ValueError("normal equations do not match an HRF of %d taps" % K)
    M = ne2.shape[0]
    theta = torch.empty((M,), dtype=torch.float64, device=dev)
    cost = torch.empty((M,), dtype=torch.float64, device=dev)
    taps = torch.empty((M, K), dtype=torch.float64, device=dev)
    lo, hi = float(bounds[0]), float(bounds[1])
    with torch.cuda.device(dev):
        rc = lib.pb_theta_fit(ne2.data_ptr(), _ld(ne2), M, K, t_dev.data_ptr(), p_delay / p_disp,
                              dt / p_disp, undershoot / u_disp, dt / u_disp, p_u_ratio, lo, hi,
                              int(n_refine), theta.data_ptr(), cost.data_ptr(), taps.data_ptr(),
                              _ld(taps), _stream_ptr(dev))
    _lib.check(rc, "pb_theta_fit")
    return theta, cost, taps


# ---- the shared-HRF outer iteration in three launches (z-step, normal equations, theta fit) -------
def hrf_normal_eq_w(W, Y, n_taps, work=None, out=None):
    """:func:`hrf_normal_eq` (summed over voxels) straight from the innovation ``W = diff_z``: the
    cumulative sum (pybold/bold_signal.py:326) and ``sum_v ||w_v||_1`` are taken in the same pass.
    Returns float64 ``(K*K + K + 2,)``: the normal equations followed by the L1 sum -- the message of
    the outer iteration's one all-reduce (``pb_hrf_normal_eq_w``)."""
    lib = _lib.load()
    W = _rows(W, torch.float64, "W")
    Y = _rows(Y, torch.float32, "Y")
    dev = W.device
    V, N = W.shape
    K = int(n_taps)
    ne = int(lib.pb_hrf_normal_eq_len(K)) + 1
    if out is None:
        out = torch.empty((ne,), dtype=torch.float64, device=dev)
    if work is None or work.numel() < ne:
        work = torch.empty((2048 * ne,), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.pb_hrf_normal_eq_w(W.data_ptr(), _ld(W) if V else N, Y.data_ptr(), _ld(Y) if V else N, V, N, K,
                                    work.data_ptr(), work.numel(), out.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "pb_hrf_normal_eq_w")
    return out


def theta_fit_step(msg, t_r, dur, bounds, n_scans, lbda, n_refine=3, dt=0.001, p_delay=6, undershoot=16.0,
                   p_disp=1.0, u_disp=1.0, p_u_ratio=0.167):
    """:func:`theta_fit` on ONE message of :func:`hrf_normal_eq_w` that also leaves on the device what the
    next z-step and the cost trace need: returns ``(theta (1,), cost (1,), taps (K,), step (1,),
    jcost (1,))`` with ``step = 1 / ||A^T A||_F`` for the new HRF on ``n_scans`` scans and ``jcost =
    (2 F(theta*) + lbda ||w||_1) / ||y||^2`` (pybold/bold_signal.py:249-254, :337-342)."""
    lib = _lib.load()
    dev = msg.device
    t_dev = _sample_times_on(dev, t_r, dur, dt)
    K = t_dev.numel()
    if msg.dtype != torch.float64 or msg.numel() != int(lib.pb_hrf_normal_eq_len(K)) + 1 or not msg.is_contiguous():
        raise ValueError("msg must be the contiguous float64 output of hrf_normal_eq_w for %d taps" % K)
    out = torch.empty((4 + K,), dtype=torch.float64, device=dev)
    theta, cost, step, jc, taps = out[0:1], out[1:2], out[2:3], out[3:4], out[4:]
    with torch.cuda.device(dev):
        rc = lib.pb_theta_fit_step(msg.data_ptr(), K, t_dev.data_ptr(), p_delay / p_disp, dt / p_disp,
                                   undershoot / u_disp, dt / u_disp, p_u_ratio, float(bounds[0]), float(bounds[1]),
                                   int(n_refine), int(n_scans), float(lbda), theta.data_ptr(), cost.data_ptr(),
                                   taps.data_ptr(), step.data_ptr(), jc.data_ptr(), _stream_ptr(dev))
    _lib.check(rc, "pb_theta_fit_step")
    return theta, cost, taps, step, jc
