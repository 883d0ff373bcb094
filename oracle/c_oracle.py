"""ctypes access to oracle/libfista_oracle.so (the C form of the CPU oracle).
Test infrastructure / CPU baseline only."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _so_path():
    """AVX2+FMA build when the host CPU has both, else the plain x86-64 build."""
    try:
        flags = open("/proc/cpuinfo").read()
        fast = " avx2" in flags and " fma" in flags
    except OSError:
        fast = False
    return os.path.join(_HERE, "libfista_oracle.so" if fast else "libfista_oracle_generic.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is None:
        so = _so_path()
        if not os.path.exists(so):
            build()
        lib = ctypes.CDLL(so)
        lib.oracle_fista_batch.restype = ctypes.c_int
        lib.oracle_fista_batch.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
            ctypes.c_double, ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_int]
        lib.oracle_deconv_auto_lbda_batch.restype = None
        lib.oracle_deconv_auto_lbda_batch.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_double, ctypes.c_int,
            ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        _lib = lib
    return _lib


def fista_batch(Y, hrf, lbda, step, n_iter, W0=None, want_J=False, threads=0):
    """Same contract as pybold_oracle.fista_batch; returns (W, J or None, threads used)."""
    lib = load()
    Y = np.ascontiguousarray(np.atleast_2d(Y), dtype=np.float64)
    V, N = Y.shape
    h = np.ascontiguousarray(hrf, dtype=np.float64)
    W = np.zeros((V, N)) if W0 is None else np.array(W0, dtype=np.float64, order="C")
    J = np.zeros((V, n_iter)) if want_J else None
    lv = None
    if np.ndim(lbda) > 0:
        lv = np.ascontiguousarray(lbda, dtype=np.float64)
        lbda = 0.0
    used = lib.oracle_fista_batch(Y.ctypes.data, V, N, h.ctypes.data, len(h), float(lbda),
                                  lv.ctypes.data if lv is not None else None, float(step),
                                  int(n_iter), W.ctypes.data,
                                  J.ctypes.data if J is not None else None, int(threads))
    return W, J, used


def deconv_auto_lbda_batch(Y, hrf, sigma, lipschitz, early_stopping=True, tol=1.0e-6, wind=6,
                           nb_iter=1000, nb_sub_iter=1000, threads=0):
    """C form of pybold_oracle.deconv_auto_lbda for every row of Y (one noise level per row).
    Returns (W (V, N), J, R, G as (V, nb_iter) arrays padded with NaN, n_outer (V,))."""
    lib = load()
    Y = np.ascontiguousarray(np.atleast_2d(Y), dtype=np.float64)
    V, N = Y.shape
    h = np.ascontiguousarray(hrf, dtype=np.float64)
    sig = np.ascontiguousarray(np.broadcast_to(np.asarray(sigma, dtype=np.float64), (V,)))
    W = np.zeros((V, N))
    J, R, G = (np.full((V, nb_iter), np.nan) for _ in range(3))
    n_outer = np.zeros(V, dtype=np.int32)
    lib.oracle_deconv_auto_lbda_batch(Y.ctypes.data, V, N, h.ctypes.data, len(h), sig.ctypes.data,
                                      float(lipschitz), int(bool(early_stopping)), float(tol),
                                      int(wind), int(nb_iter), int(nb_sub_iter), W.ctypes.data,
                                      J.ctypes.data, R.ctypes.data, G.ctypes.data,
                                      n_outer.ctypes.data, int(threads))
    return W, J, R, G, n_outer
