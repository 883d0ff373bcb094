"""pybold_amd: MI355X-native batched FISTA deconvolution behind pyBOLD's
``deconv`` / ``bd`` signatures and ``.op`` / ``.adj`` operator surface.

Importing the package does not touch the GPU; the HIP library is loaded on
first use and its absence is an error (there is no CPU fallback).
"""
from . import _lib  # noqa: F401
from .bold_signal import _loops_deconv, bd, deconv, hrf_estim, hrf_fit_err  # noqa: F401
from .convolution import (kernel_from_toeplitz, simple_convolve, simple_retro_convolve,  # noqa: F401
                          toeplitz_from_kernel)
from .hrf_model import MAX_DELTA, MIN_DELTA, spm_hrf  # noqa: F401
from .linear import ConvAndLinear, DiscretInteg  # noqa: F401
from .utils import gram_frobenius, spectral_radius_est  # noqa: F401

__version__ = "0.1.0"
