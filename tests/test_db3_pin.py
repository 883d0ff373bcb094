"""CPU-only: an INDEPENDENT pin of the db3 level-1 detail band behind
``mad_daub_noise_est`` (pybold/utils.py:16-25 calls ``pywt.wavedec(x, Wavelet('db3'), level=1)``;
PyWavelets is absent from the image, so neither the product's nor the oracle's restatement
could be run against it).  What CAN be checked without PyWavelets:

  * the six filter taps against (a) Daubechies' closed form for N = 3 and (b) a numerical
    spectral factorisation (roots of the degree-2 polynomial P(y) = 1 + 3y + 6y^2 in
    y = sin^2(w/2), minimum-phase selection, three zeros at z = -1), joined by the
    quadrature-mirror relation dec_hi[k] = (-1)^(k+1) rec_lo[k];
  * orthonormality of the filter pair, three vanishing moments of the high-pass;
  * the analysis convention (half-sample symmetric extension, odd-phase downsampling,
    (N + 5) // 2 coefficients): it is the one for which the standard synthesis
    x[n] = sum_k cA[k] rec_lo[n + F - 2 - 2k] + cD[k] rec_hi[n + F - 2 - 2k] reconstructs x
    EXACTLY for even and odd lengths -- and hand-computed responses to an impulse / a ramp.

The row stays "parity unpinned" in DESIGN.md until reference-side vectors exist."""
from math import comb, sqrt

import numpy as np
import pytest

from oracle import pybold_oracle as orc
from pybold_amd import utils

F = 6


def closed_form_db3():
    s10, s = sqrt(10.0), sqrt(5.0 + 2.0 * sqrt(10.0))
    return np.array([1 + s10 + s, 5 + s10 + 3 * s, 10 - 2 * s10 + 2 * s, 10 - 2 * s10 - 2 * s,
                     5 + s10 - 3 * s, 1 + s10 - s]) / (16.0 * sqrt(2.0))


def spectral_factorisation_db3():
    n = 3
    p = [comb(n - 1 + k, k) for k in range(n)]                  # P(y) = 1 + 3y + 6y^2
    zs = []
    for y in np.roots(p[::-1]):
        r = np.roots([1.0, -(2.0 - 4.0 * y), 1.0])              # y = (2 - z - 1/z) / 4
        zs.append(r[np.argmin(np.abs(r))])                      # minimum phase: |z| < 1
    h = np.convolve(np.poly([-1.0] * n), np.poly(zs).real)      # (1 + z^-1)^3 L(z)
    return h / h.sum() * sqrt(2.0)


def qmf_high(h):
    return np.array([(-1) ** (k + 1) * h[k] for k in range(len(h))])


def test_filter_taps_from_two_independent_derivations():
    h_cf, h_sf = closed_form_db3(), spectral_factorisation_db3()
    np.testing.assert_allclose(h_sf, h_cf, rtol=0, atol=1e-14)           # the two derivations agree
    # the shipped constants are PyWavelets' tabulated db3 taps (what the reference computes
    # with); that table is 3.6e-12 away from the exact values in two taps
    np.testing.assert_allclose(utils._DB3_DEC_HI, qmf_high(h_cf), rtol=0, atol=1e-11)
    np.testing.assert_array_equal(orc.DB3_DEC_HI, utils._DB3_DEC_HI)


def test_orthonormality_and_vanishing_moments():
    h, g = closed_form_db3(), utils._DB3_DEC_HI
    assert h.sum() == pytest.approx(sqrt(2.0), abs=1e-14)
    for m in range(3):
        assert np.dot(h[2 * m:], h[:F - 2 * m]) == pytest.approx(1.0 if m == 0 else 0.0, abs=1e-14)
        assert np.dot(g[2 * m:], g[:F - 2 * m]) == pytest.approx(1.0 if m == 0 else 0.0, abs=1e-11)
    assert g.sum() == pytest.approx(0.0, abs=1e-11)
    k = np.arange(F)
    for p in range(3):                                           # db3: three vanishing moments
        assert np.dot(k ** p, g) == pytest.approx(0.0, abs=1e-9)
    assert abs(np.dot(k ** 3, g)) > 0.1


def analysis(x):
    """Both bands with the product's convention (the product only needs cD)."""
    h, g = closed_form_db3(), utils._DB3_DEC_HI
    n = len(x)
    xe = np.concatenate([x[:F - 1][::-1], x, x[::-1][:F - 1]])   # half-sample symmetric
    n_out = (n + F - 1) // 2
    base = 1 + (F - 1) + 2 * np.arange(n_out)                    # xe index of x_ext[2k + 1]
    dec_lo = h[::-1]
    cA = sum(dec_lo[j] * xe[base - j] for j in range(F))
    cD = sum(g[j] * xe[base - j] for j in range(F))
    return cA, cD


@pytest.mark.parametrize("n", [5, 6, 7, 36, 37, 240, 300])
def test_perfect_reconstruction_pins_phase_and_length(n):
    h, g = closed_form_db3(), utils._DB3_DEC_HI
    x = np.random.RandomState(n).randn(n)
    cA, cD = analysis(x)
    assert len(cD) == (n + 5) // 2
    np.testing.assert_allclose(orc.db3_detail_level1(x), cD, rtol=0, atol=1e-13)
    rec_lo, rec_hi = h, g[::-1]
    out = np.zeros(2 * len(cA) + F)
    for k in range(len(cA)):
        out[2 * k:2 * k + F] += rec_lo * cA[k] + rec_hi * cD[k]
    np.testing.assert_allclose(out[F - 2:F - 2 + n], x, rtol=0, atol=1e-10)
    # the other downsampling phase does NOT reconstruct with this synthesis: the check is sharp
    assert np.abs(out[F - 3:F - 3 + n] - x).max() > 1e-2


def test_hand_computed_responses_and_the_product_estimate():
    g = utils._DB3_DEC_HI
    n, p = 40, 17
    x = np.zeros(n)
    x[p] = 1.0
    cD = orc.db3_detail_level1(x)
    expect = np.zeros((n + 5) // 2)
    for k in range(len(expect)):                                  # cD[k] = g[2k + 1 - p - (F-1) + (F-1)]
        j = 2 * k + 1 - p
        if 0 <= j < F:
            expect[k] = g[j]
    np.testing.assert_allclose(cD, expect, rtol=0, atol=1e-15)
    ramp = 0.5 * np.arange(n) - 3.0
    interior = orc.db3_detail_level1(ramp)[3:-3]
    np.testing.assert_allclose(interior, 0.0, atol=1e-9)          # vanishing moments kill a ramp
    # product == oracle == MAD of that band / 0.6744, 1-D and batched
    rng = np.random.RandomState(1)
    X = rng.randn(5, 300) * np.array([0.5, 1.0, 2.0, 4.0, 8.0])[:, None]
    sig = utils.mad_daub_noise_est(X)
    for v in range(5):
        cDv = orc.db3_detail_level1(X[v])
        ref = np.median(np.abs(cDv - np.median(cDv))) / 0.6744
        assert sig[v] == pytest.approx(ref, rel=1e-13)
        assert utils.mad_daub_noise_est(X[v]) == pytest.approx(ref, rel=1e-13)
        assert orc.mad_daub_noise_est(X[v]) == pytest.approx(ref, rel=1e-13)
    # white noise of standard deviation s: the estimate is s up to sampling error
    assert np.all(np.abs(sig / np.array([0.5, 1.0, 2.0, 4.0, 8.0]) - 1.0) < 0.3)   # 152 coefficients each
