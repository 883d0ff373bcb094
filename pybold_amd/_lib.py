"""ctypes binding of ``libpybold_hip.so`` (C ABI in ``include/pybold_hip.h``).

There is no CPU fallback: if the shared library is missing or a call fails the
caller gets an exception.  Build the library with ``python -c "import
__graft_entry__ as g; g.build()"`` or ``make -C pybold_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYBOLD_HIP_LIB points at an alternative build of the same library (A/B kernel experiments)
LIB_PATH = os.environ.get("PYBOLD_HIP_LIB") or os.path.join(_HERE, "libpybold_hip.so")

PB_FLAG_FORCE_GENERIC = 1
PB_FLAG_FORCE_FAST = 2
PB_FLAG_NO_PAIR = 4
PB_FLAG_FORCE_PAIR = 8
PB_FLAG_FORCE_WIDE = 16
PB_FLAG_ONE_LAUNCH = 32
PB_FLAG_DIRECT_FIR = 64
PB_FLAG_ONE_STREAM = 128
PB_FLAG_COLD_START = 256
PB_FLAG_NO_CERT = 512
PB_FLAG_FORCE_CERT = 1024
PB_FLAG_CERT_NO_RESOLVE = 2048
PB_FLAG_NO_PARTITION = 4096
PB_FLAG_NO_MFMA = 8192
PB_FLAG_FORCE_MFMA = 16384
PB_FLAG_NO_RHO_GUARD = 32768
PB_FLAG_FORCE_MFMA2 = 65536
PB_FLAG_ONLY_DENSE = 131072
PB_FLAG_ONLY_SPARSE = 262144
PB_FLAG_NO_ILL_GUARD = 524288
PB_PATH_DENSE_RATIO = 0.19          # include/pybold_hip.h (series of up to 310 scans; 0.22 beyond)
PB_STOP_NONE = 0
PB_STOP_LOOPS = 1
PB_STOP_WINDOW = 2

_c_int = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_dbl = ctypes.c_double
_ptr = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/pybold_hip.h one to one
SIGNATURES = {
    "pb_version": (_c_int, []),
    "pb_init": (_c_int, []),
    "pb_last_error": (ctypes.c_char_p, []),
    "pb_fista_has_fast_path": (_c_int, [_c_int, _c_int]),
    "pb_fista_which_kernel": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int]),
    "pb_fista_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr, _ptr]),
    "pb_fista_plan_ex": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, ctypes.c_uint, _ptr, _ptr, _ptr]),
    "pb_fista_solve": (_c_int, [
        _ptr, _c_i64, _c_int,            # y_dev, ldy, y_rep
        _ptr, _c_i64, _c_int, _c_int,    # w_dev, ldw, P, N
        _ptr, _ptr, _c_int,              # taps_host, taps_dev, K
        _c_dbl, _c_dbl, _ptr,            # step, lbda, lbda_dev
        _ptr, _c_int,                    # betas_dev, n_iter
        _ptr, _c_i64,                    # J_dev, ldj
        _c_int, _c_dbl, _c_int, _ptr,    # stop_mode, tol, wind, n_done_dev
        ctypes.c_uint, _ptr]),           # flags, stream
    "pb_fista_work_len": (_c_i64, [_c_int, _c_int]),
    "pb_fista_list_plan": (_c_int, [_c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "pb_fista_solve_ex": (_c_int, [
        _ptr, _c_i64, _c_int,            # y_dev, ldy, y_rep
        _ptr, _c_i64, _c_int, _c_int,    # w_dev, ldw, P, N
        _ptr, _ptr, _c_int,              # taps_host, taps_dev, K
        _c_dbl, _c_dbl, _ptr,            # step, lbda, lbda_dev
        _ptr, _c_int,                    # betas_dev, n_iter
        _ptr, _c_i64,                    # J_dev, ldj
        _c_int, _c_dbl, _c_int, _ptr,    # stop_mode, tol, wind, n_done_dev
        ctypes.c_uint, _ptr,             # flags, stream
        _ptr, _c_dbl, _ptr, _c_i64]),    # lmax_dev, dense_ratio, work_dev, work_len
    "pb_fista_solve_backtrack_d": (_c_int, [
        _ptr, _c_i64, _c_int, _ptr, _c_i64, _c_int, _c_int,      # y_dev, ldy, y_rep, w_dev, ldw, P, N
        _ptr, _c_int, _c_dbl, _c_dbl, _c_int,                    # taps_dev, K, step0, eta, max_halvings_per_iter
        _c_dbl, _ptr, _ptr, _c_int,                              # lbda, lbda_dev, betas_dev, n_iter
        _ptr, _ptr, _ptr, ctypes.c_uint, _ptr]),                 # n_done_dev, step_out_dev, halvings_out_dev, flags, stream
    "pb_fista_path_work_len": (_c_i64, [_c_int]),
    "pb_fista_solve_path": (_c_int, [
        _ptr, _c_i64, _c_int,            # y_dev, ldy, y_rep
        _ptr, _c_i64, _c_int, _c_int,    # w_dev, ldw, P, N
        _ptr, _ptr, _c_int,              # taps_host, taps_dev, K
        _c_dbl, _ptr, _ptr, _c_dbl,      # step, lbda_dev, lmax_dev, dense_ratio
        _ptr, _c_int,                    # betas_dev, n_iter
        _ptr, _ptr, _c_i64,              # n_done_dev, work_dev, work_len
        ctypes.c_uint, _ptr]),           # flags, stream
    "pb_fista_outputs": (_c_int, [_ptr, _c_i64, _c_int, _c_int, _ptr, _c_int,
                                  _ptr, _c_i64, _ptr, _c_i64, _ptr]),
    "pb_fista_stats": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _c_int, _ptr,
                                _c_int, _ptr, _ptr, _ptr]),
    "pb_integ_op": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _ptr]),
    "pb_integ_adj": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _ptr]),
    "pb_conv": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _c_int,
                         _ptr, _c_int, _ptr]),
    "pb_corr": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _c_int,
                         _ptr, _c_int, _ptr]),
    "pb_op_forward": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int,
                               _c_int, _ptr, _c_int, _ptr]),
    "pb_op_adjoint": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int,
                               _c_int, _ptr, _c_int, _ptr]),
    "pb_hrf_cost": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _ptr,
                             _c_int, _c_int, _ptr, _ptr]),
    "pb_spectral_radius": (_c_int, [_ptr, _c_int, _ptr, _c_int, _c_int, _c_dbl, _ptr, _ptr]),
    "pb_fista_solve_pp": (_c_int, [
        _ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int,     # y, ldy, w, ldw, P, N
        _ptr, _c_i64, _c_int, _ptr,                     # taps_dev, ldt, K, step_dev
        _c_dbl, _ptr, _ptr, _c_int,                     # lbda, lbda_dev, betas_dev, n_iter
        _c_int, _c_dbl, _ptr, ctypes.c_uint, _ptr]),    # stop_mode, tol, n_done, flags, stream
    "pb_hrf_cost_pv": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _ptr,
                                _c_int, _c_int, _ptr, _ptr]),
    "pb_gram_frobenius": (_c_int, [_ptr, _c_i64, _c_int, _c_int, _c_int, _ptr, _ptr]),
    "pb_fista_outputs_pp": (_c_int, [_ptr, _c_i64, _c_int, _c_int, _ptr, _c_i64, _c_int,
                                     _ptr, _c_i64, _ptr, _c_i64, _ptr]),
    "pb_spm_hrf": (_c_int, [_ptr, _c_int, _ptr, _c_int, _c_dbl, _c_dbl, _c_dbl, _c_dbl, _c_dbl,
                            _ptr, _ptr]),
    # float64-y forms
    "pb_fista_solve_d": (_c_int, [
        _ptr, _c_i64, _c_int,            # y_dev (float64), ldy, y_rep
        _ptr, _c_i64, _c_int, _c_int,    # w_dev, ldw, P, N
        _ptr, _ptr, _c_int,              # taps_host, taps_dev, K
        _c_dbl, _c_dbl, _ptr,            # step, lbda, lbda_dev
        _ptr, _c_int,                    # betas_dev, n_iter
        _ptr, _c_i64,                    # J_dev (float64), ldj
        _c_int, _c_dbl, _c_int, _ptr,    # stop_mode, tol, wind, n_done_dev
        ctypes.c_uint, _ptr]),           # flags, stream
    "pb_fista_stats_d": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _c_int, _ptr,
                                  _c_int, _ptr, _ptr, _ptr]),
    "pb_hrf_cost_d": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _ptr,
                               _c_int, _c_int, _ptr, _ptr]),
    "pb_hrf_cost_pv_d": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _ptr,
                                  _c_int, _c_int, _ptr, _ptr]),
    "pb_lambda_max": (_c_int, [_ptr, _c_i64, _c_int, _c_int, _ptr, _c_int, _ptr, _ptr]),
    "pb_lambda_max_d": (_c_int, [_ptr, _c_i64, _c_int, _c_int, _ptr, _c_int, _ptr, _ptr]),
    "pb_inf_norm": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_i64, _ptr]),
    "pb_hrf_normal_eq_len": (_c_i64, [_c_int]),
    "pb_hrf_normal_eq": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _c_int, _c_int,
                                  _ptr, _c_i64, _ptr, _ptr]),
    "pb_hrf_normal_eq_d": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _c_int, _c_int,
                                    _ptr, _c_i64, _ptr, _ptr]),
    "pb_theta_fit": (_c_int, [_ptr, _c_i64, _c_int, _c_int, _ptr, _c_dbl, _c_dbl, _c_dbl, _c_dbl,
                              _c_dbl, _c_dbl, _c_dbl, _c_int, _ptr, _ptr, _ptr, _c_i64, _ptr]),
    "pb_hrf_normal_eq_w": (_c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_int, _c_int, _c_int, _ptr, _c_i64,
                                    _ptr, _ptr]),
    "pb_theta_fit_step": (_c_int, [_ptr, _c_int, _ptr, _c_dbl, _c_dbl, _c_dbl, _c_dbl, _c_dbl, _c_dbl,
                                   _c_dbl, _c_int, _c_int, _c_dbl, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
}

_lib = None


class PyboldHipError(RuntimeError):
    """A libpybold_hip entry point returned an error code."""


def load():
    """Load (once) and return the ctypes handle; raises if the HIP library has
    not been built -- the product never computes on the CPU instead."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "pybold_amd: %s not found; build it with "
                "`make -C pybold_amd/csrc` (hipcc, gfx950). There is no CPU "
                "fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if a symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().pb_last_error().decode("utf-8", "replace")
        raise PyboldHipError("%s failed (%d): %s" % (what or "libpybold_hip", rc, msg))
