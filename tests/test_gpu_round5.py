"""Round 5.

  * `deconv(lbda=None)` (pybold/bold_signal.py:99-214) against fixtures the REAL reference produced
    with the noise level injected (tests/golden/make_golden_r5.py): small budgets, runs where both
    stop windows fire, runs where the search drives lambda NEGATIVE, and the reference's default call.
"""
import numpy as np
import pytest
import torch

from oracle import pybold_oracle as orc
from test_oracle_golden import AUTO_LBDA_CHAOTIC_AFTER, auto_lbda_runs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    from pybold_amd import solver as s
    return s


def rel(a, b):
    return np.linalg.norm(np.asarray(a) - b) / (np.linalg.norm(b) + 1e-300)


def rel_rows(a, b):
    return np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-300)


def _deconv_with_sigma(monkeypatch, y, hrf, sigma, **kw):
    """pybold_amd.deconv(lbda=None) with the db3 MAD estimate (the one unpinned piece: no PyWavelets here
    or there) replaced by the value the fixture's reference run was given."""
    import pybold_amd
    from pybold_amd import bold_signal
    monkeypatch.setattr(bold_signal, "mad_daub_noise_est", lambda x: sigma)
    np.random.seed(0)                       # spectral_radius_est draws from the global RNG (:52), as the fixture did
    return pybold_amd.deconv(y, 1.0, hrf, lbda=None, **kw)


def test_deconv_auto_lambda_against_the_reference_small_budgets_and_both_stop_windows(golden, monkeypatch):
    """Every non-default run of auto_lbda.npz through the 1-D call (float64 end to end: `fista_exact_kernel`,
    and the LDS kernel for wind = 4): the lists J, R, G entry by entry and x, z, diff_z.  Tolerance 1e-7:
    float64 kernels against the reference's float64 with other summation orders; the runs whose alpha passes
    within 0.02 of zero (lambda up to 25, or negative) amplify a last-digit difference by up to 1e5."""
    g = golden("auto_lbda")
    runs = auto_lbda_runs(g, default=False)
    worst, neg, fired = 0.0, 0, 0
    for tag, case, sigma, kw in runs:
        x, z, dz, J, R, G = _deconv_with_sigma(monkeypatch, g[case + "_y"], g[case + "_hrf"], sigma, **kw)
        assert isinstance(J, list) and len(J) == len(g["J_" + tag]), (tag, len(J), len(g["J_" + tag]))
        errs = [rel(dz, g["dz_" + tag]), rel(z, g["z_" + tag]), rel(x, g["x_" + tag]),
                rel(J, g["J_" + tag]), rel(R, g["R_" + tag]), rel(G, g["G_" + tag])]
        assert max(errs) < 1e-7, (tag, errs)
        worst = max(worst, max(errs))
        neg += bool((g["alpha_" + tag] < 0).any())
        fired += len(J) < kw["nb_iter"]
    print("deconv(lbda=None) vs the reference: %d runs, worst rel. error %.2e; lambda < 0 in %d, alpha window fired in %d"
          % (len(runs), worst, neg, fired))
    assert neg >= 10 and fired >= 4


def test_deconv_auto_lambda_batch_rows_follow_the_reference(golden, monkeypatch):
    """The same fixtures as a BATCH (the three sigma of case 1 as three rows of one call; the third one's lambda goes
    negative): this branch runs on the float64 kernels for batches too (bold_signal._deconv_auto_lbda says why), rows
    leave the outer loop on their own (NaN padding), every row follows the reference's run of that row."""
    g = golden("auto_lbda")
    y, hrf, sig = g["c1_y"], g["c1_hrf"], g["c1_sigma"]
    Y = np.repeat(y[None, :], 3, axis=0)
    for o, i, e, tol_f in ((5, 50, 0, 1e-7), (20, 10, 0, 1e-7), (20, 50, 1, 1e-7)):
        tags = ["c1_s%d_o%d_i%d_e%d" % (s, o, i, e) for s in range(3)]
        X, Z, W, J, R, G = _deconv_with_sigma(monkeypatch, Y, hrf, sig.copy(), nb_iter=o, nb_sub_iter=i, early_stopping=bool(e))
        assert J.shape == (o, 3)
        for s, tag in enumerate(tags):
            errs = [rel(W[s], g["dz_" + tag]), rel(Z[s], g["z_" + tag]), rel(X[s], g["x_" + tag]),
                    rel(J[:, s], g["J_" + tag]), rel(R[:, s], g["R_" + tag]), rel(G[:, s], g["G_" + tag])]
            assert max(errs) < tol_f, (tag, errs)
    # both windows firing, rows stopping at different outer iterations
    for tol, wind in ((1e-2, 6), (1e-2, 4)):
        tags = ["c1_s%d_o60_i300_e1_t%g_w%d" % (s, tol, wind) for s in range(3)]
        X, Z, W, J, R, G = _deconv_with_sigma(monkeypatch, Y, hrf, sig.copy(), nb_iter=60, nb_sub_iter=300, early_stopping=True,
                                              tol=tol, wind=wind)
        n_ref = [len(g["J_" + t]) for t in tags]
        assert J.shape[0] == max(n_ref) and min(n_ref) < max(n_ref)
        for s, tag in enumerate(tags):
            n = n_ref[s]
            assert np.isnan(J[n:, s]).all() and not np.isnan(J[:n, s]).any(), tag
            errs = [rel(W[s], g["dz_" + tag]), rel(Z[s], g["z_" + tag]), rel(X[s], g["x_" + tag]),
                    rel(J[:n, s], g["J_" + tag]), rel(R[:n, s], g["R_" + tag]), rel(G[:n, s], g["G_" + tag])]
            assert max(errs) < 1e-7, (tag, errs)


def test_deconv_auto_lambda_reference_default_call(golden, monkeypatch):
    """`deconv(y, t_r, hrf)` -- lbda=None, 1000 x 1000 iterations, tol 1e-6, wind 6 -- against the reference's own run of
    that call (three noise levels on golden case 1, one on case 2; up to 10^6 inner iterations per run, the window
    rule deciding each inner solve's length).  Three runs follow the reference to 1e-6 or better through all outer
    iterations (1000, 1000 and 196: the alpha window fires there); the fourth is the one where alpha passes through
    7e-4 (AUTO_LBDA_CHAOTIC_AFTER, tests/test_oracle_golden.py): compared over the outer iterations before that."""
    import time
    g = golden("auto_lbda")
    for tag, case, sigma, kw in auto_lbda_runs(g, default=True):
        t0 = time.perf_counter()
        x, z, dz, J, R, G = _deconv_with_sigma(monkeypatch, g[case + "_y"], g[case + "_hrf"], sigma)
        dt = time.perf_counter() - t0
        n = len(g["J_" + tag])
        assert len(J) == n, (tag, len(J), n)
        upto = AUTO_LBDA_CHAOTIC_AFTER.get(tag, n)
        eJ, eR, eG = (rel(np.array(v)[:upto], g[k + tag][:upto]) for v, k in ((J, "J_"), (R, "R_"), (G, "G_")))
        e_dz, e_x = rel(dz, g["dz_" + tag]), rel(x, g["x_" + tag])
        print("%s: %d outer iterations in %.1f s; J %.1e R %.1e G %.1e (first %d), diff_z %.1e x %.1e"
              % (tag, n, dt, eJ, eR, eG, upto, e_dz, e_x))
        assert max(eJ, eR, eG) < 1e-6, tag
        if tag not in AUTO_LBDA_CHAOTIC_AFTER:
            assert e_dz < 1e-6 and e_x < 1e-6 and rel(z, g["z_" + tag]) < 1e-6, tag
        else:
            assert np.isfinite(dz).all()


def test_negative_lambda_is_the_float64_path_only(solver, golden):
    """pybold/bold_signal.py:66 with a negative threshold grows every entry; `pb_fista_solve_d` restates it, the
    float32-FIR entry point refuses a negative scalar lambda instead of clamping silently."""
    g = golden("case1")
    y, hrf, lip = g["y"], g["hrf"], float(g["lipschitz"])
    Yd = torch.from_numpy(y[None, :]).cuda()
    ref = orc.fista_batch(y[None, :], hrf, -0.7, 1.0 / lip, 60)
    for force in (None, "generic"):
        W, _, _ = solver.fista_solve(Yd, hrf, -0.7, 1.0 / lip, 60, force=force)
        assert rel(W.cpu().numpy(), ref) < 1e-11
    # anti-shrinkage: no zero survives but the last sample's (gradient exactly 0 there: h[0] = 0, sign(0) = 0)
    assert np.abs(ref[0, :-1]).min() > 0.0 and ref[0, -1] == 0.0 and float(W[0, -1]) == 0.0
    with pytest.raises(ValueError):
        solver.fista_solve(Yd.float(), hrf, -0.7, 1.0 / lip, 60)
    with pytest.raises(ValueError):
        solver.fista_solve(Yd.float(), hrf, np.array([-0.7]), 1.0 / lip, 60)
