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


# --------------------------------------------------------------------------------------------
# Round 5: partition before solving, for every call shape (pb_fista_solve_ex)
# --------------------------------------------------------------------------------------------
def _mixed_batch(V, seed):
    from pybold_amd import data
    hrf = orc.spm_hrf(1.0, 1.0, 30.0)[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=5, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=seed)
    lip = 0.9 * orc.spectral_radius_est(orc._MatrixFreeH(hrf), np.random.RandomState(0).randn(Y.shape[1]))
    return Y, hrf, 1.0 / lip


def _oracle_rows(Y, hrf, lam, step, n_it, rows):
    from oracle import c_oracle
    lam_rows = lam[rows] if np.ndim(lam) else lam
    W, _, _ = c_oracle.fista_batch(Y[rows].cpu().numpy().astype(np.float64), hrf, lam_rows, step, n_it, threads=16)
    return W


def test_partition_before_solving_every_call_shape(solver):
    """20 000 voxels with ONE scalar lambda placed at the batch's median of 0.13 lambda_max,v: half of the problems are
    dense (matrix pipe), half sparse (vector forms).  The partitioned call -- plain, with the cost trace, with the
    window rule as certificate, with the _loops_deconv rule, with per-problem lambdas with and without the caller's
    lambda_max -- returns what the vector dispatch returns (<= 4e-6 per problem: two arithmetics), every n_done set,
    and a random sample equals the C float64 oracle within eps = 1e-5."""
    V, n_it = 20000, 300
    Y, hrf, step = _mixed_batch(V, 11)
    lmax = solver.lambda_max(Y, hrf)
    lam_s = float((0.13 * lmax).median())
    rows = np.random.RandomState(3).choice(V, 256, replace=False)
    ref = _oracle_rows(Y, hrf, lam_s, step, n_it, rows)
    ok = np.linalg.norm(ref, axis=1) > 0
    for kw in (dict(), dict(want_J=True), dict(want_J=True, stop="window", tol=1e-6, wind=6), dict(stop="window", tol=1e-6, wind=6)):
        W, J, nd = solver.fista_solve(Y, hrf, lam_s, step, n_it, **kw)
        Wv, Jv, ndv = solver.fista_solve(Y, hrf, lam_s, step, n_it, force="valu", **kw)
        Wn, Jn, ndn = solver.fista_solve(Y, hrf, lam_s, step, n_it, force="nopart", **kw)
        assert int(nd.min()) == n_it and int(nd.max()) == n_it, (kw, int(nd.min()), int(nd.max()))
        scale = Wv.norm(dim=1).clamp_min(1e-300)
        e_v = float(((W - Wv).norm(dim=1) / scale).max())
        e_n = float(((Wn - Wv).norm(dim=1) / scale).max())
        e_o = rel_rows(W[rows].cpu().numpy()[ok], ref[ok]).max()
        print("partitioned %-60s vs vector dispatch %.2e (round-4 dispatch: %.2e), vs oracle %.2e" % (kw, e_v, e_n, e_o))
        assert e_v < 4e-6 and e_o < 1e-5, kw
        if J is not None:
            assert bool(torch.isfinite(J).all())
            assert float(((J - Jv).abs() / Jv.abs().clamp_min(1e-30)).max()) < 1e-4
    # the _loops_deconv rule: stop iterations and iterates of the exact rule
    Wl, _, ndl = solver.fista_solve(Y, hrf, lam_s, step, n_it, stop="loops", tol=2e-3)
    Wlv, _, ndlv = solver.fista_solve(Y, hrf, lam_s, step, n_it, stop="loops", tol=2e-3, force="valu")
    assert int((ndl - ndlv).abs().max()) <= 1 and float((ndl != ndlv).float().mean()) < 0.01      # (criterion on 22-bit operators: ADVICE r4)
    same = (ndl == ndlv)
    assert float(((Wl - Wlv).norm(dim=1) / Wlv.norm(dim=1).clamp_min(1e-300))[same].max()) < 1e-5
    # per-problem lambdas: a path of three values per voxel around the class boundary
    c = torch.tensor([0.02, 0.13, 0.5], dtype=torch.float64, device=Y.device)
    lam_p = (lmax[:, None] * c[None, :]).reshape(-1)
    for lm in (None, lmax):
        Wp, _, ndp = solver.fista_solve(Y, hrf, lam_p, step, n_it, y_rep=3, lmax=lm)
        Wpv, _, _ = solver.fista_solve(Y, hrf, lam_p, step, n_it, y_rep=3, force="valu")
        assert int(ndp.min()) == n_it
        sc = Wpv.norm(dim=1)
        nz = sc > 0
        assert float(((Wp - Wpv).norm(dim=1)[nz] / sc[nz]).max()) < 4e-6
        assert bool((Wp[~nz] == 0).all())


def test_partition_extremes_and_small_lists(solver):
    """All dense, all sparse, one class of a single problem, a handed-back list (constant series at lambda_max / 10 fail
    the accuracy guard) -- and the measurement aids PB_FLAG_ONLY_DENSE / _ONLY_SPARSE leave the other class untouched."""
    V, n_it = 9000, 200
    Y, hrf, step = _mixed_batch(V, 5)
    lmax = solver.lambda_max(Y, hrf)
    ref_rows = np.arange(0, V, 97)
    for c, what in ((0.01, "all dense"), (0.9, "all sparse")):
        lam = lmax * c
        W, _, nd = solver.fista_solve(Y, hrf, lam, step, n_it, lmax=lmax)
        ref = _oracle_rows(Y, hrf, lam.cpu().numpy(), step, n_it, ref_rows)
        okr = np.linalg.norm(ref, axis=1) > 0
        assert int(nd.min()) == n_it and rel_rows(W[ref_rows].cpu().numpy()[okr], ref[okr]).max() < 1e-5, what
    lam = lmax * 0.9
    lam[1234] = lmax[1234] * 0.01                       # a dense class of ONE problem
    W, _, nd = solver.fista_solve(Y, hrf, lam, step, n_it, lmax=lmax)
    ref = _oracle_rows(Y, hrf, lam.cpu().numpy(), step, n_it, np.array([1233, 1234, 1235]))
    assert int(nd.min()) == n_it and rel_rows(W[1233:1236].cpu().numpy(), ref).max() < 1e-5
    # measurement aids: "matrix-pipe launches only" / "vector launches only" -- together they cover every problem once
    Vb = 24000
    Yb, _, _ = _mixed_batch(Vb, 6)
    lmb = solver.lambda_max(Yb, hrf)
    lam = torch.where(torch.arange(Vb, device=Yb.device) % 2 == 0, lmb * 0.01, lmb * 0.9)      # 12 000 dense: above half a round
    W0 = torch.full((Vb, Yb.shape[1]), 1e-3, dtype=torch.float64, device=Yb.device)       # (a warm start inside every form's range)
    Wd, _, _ = solver.fista_solve(Yb, hrf, lam, step, n_it, lmax=lmb, W0=W0, force="path_dense")
    Ws, _, _ = solver.fista_solve(Yb, hrf, lam, step, n_it, lmax=lmb, W0=W0, force="path_sparse")
    touched_d, touched_s = ~(Wd == 1e-3).all(dim=1), ~(Ws == 1e-3).all(dim=1)
    assert bool((touched_d ^ touched_s).all())                              # every problem by exactly one of the two
    assert bool(touched_d[0::2].all()) and not bool(touched_d[1::2].any())  # here: the dense class whole on the matrix pipe
    # handed-back problems: constant series are dense by the ratio test but fail the accuracy guard at the end
    Yc = Y.clone()
    Yc[::3] = 5.0
    lmc = solver.lambda_max(Yc, hrf)
    lam = lmc * 0.1
    W, _, nd = solver.fista_solve(Yc, hrf, lam, step, n_it, lmax=lmc)
    Wv, _, _ = solver.fista_solve(Yc, hrf, lam, step, n_it, force="valu")
    _, _, ndo = solver.fista_solve(Yc, hrf, lam, step, n_it, force="mfmaonly")
    assert int(nd.min()) == n_it
    print("handed back by the matrix-pipe pass: %.1f %% of %d" % (100.0 * float((ndo < 0).float().mean()), V))
    sc = Wv.norm(dim=1).clamp_min(1e-300)
    assert float(((W - Wv).norm(dim=1) / sc).max()) < 1e-5
    flagged = ndo < 0
    assert bool(flagged.any())
    # re-solved rows = an exact vector form's, bit for bit (which one -- single row or one problem per wave -- the
    # device-side plan of the handed-back list decides from its length)
    W1, _, _ = solver.fista_solve(Yc, hrf, lam, step, n_it, force="fast1")
    Ww, _, _ = solver.fista_solve(Yc, hrf, lam, step, n_it, force="wide")
    same = (W[flagged] == W1[flagged]).all(dim=1) | (W[flagged] == Ww[flagged]).all(dim=1)
    assert bool(same.all())


def test_range_guard_between_its_samples(solver, golden):
    """The matrix-pipe kernel samples its float16 range guard (first pass of a warm start, every 8th iteration, the
    last): could a COLD start leave the range in iterations 1..6, come back, and never be flagged?
    For a cold start r_0 = -y and every series is scaled so that max |y| sits 2x .. 4x below the limit, so it would take a
    residual of twice max |y|.  Swept on the CPU oracle (step up to 2 / rho, the largest for which the plain gradient
    step is non-expansive; seven families incl. steps, ramps, alternating signs, end spikes): max |r_k| <= 1.23 max |y|
    in iterations 1..6; beyond 2 / rho the recurrence diverges for good and the periodic check sees it.  Here the same
    sweep on the kernel: every problem is EITHER within eps of the oracle OR handed back (n_done = -1); none is silently
    wrong, and up to 2 / rho none is handed back."""
    g = golden("case1")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    rho = lip / 0.9
    N = len(g["y"])
    t = np.arange(N)
    rng = np.random.RandomState(0)
    fams = [g["y"], np.where(t % 2 == 0, 1.0, -1.0), np.r_[np.zeros(N - 1), 1.0], np.r_[1.0, np.zeros(N - 1)],
            (t > 150).astype(float), rng.randn(N), t / float(N), 1e3 * (t > 150) + rng.randn(N)]
    Y = np.stack([f * s for f in fams for s in (1.0, 3.7e-3, 2.9e4)])               # (scales: the per-series power of two differs)
    Yd = torch.from_numpy(np.tile(Y, (4, 1)).astype(np.float32)).cuda()              # 96 problems: six waves
    Yo = Yd.cpu().numpy().astype(np.float64)
    for mult in (1.0, 1.5, 1.8, 1.9, 2.2, 2.6):
        for lb in (0.0, 1.0):
            step = mult / rho
            W, _, nd = solver.fista_solve(Yd, hrf, lb, step, 40, force="mfmaonly")
            ref = orc.fista_batch(Yo, hrf, lb, step, 40)
            back = (nd < 0).cpu().numpy()
            err = rel_rows(W.cpu().numpy(), ref)
            ok = np.isfinite(ref).all(axis=1) & (np.linalg.norm(ref, axis=1) > 0)
            # (beyond 2 / rho the recurrence is unstable and amplifies ANY rounding -- the float32 vector forms' too: what
            # is asked there is that nothing far off gets through unflagged)
            eps = 1e-5 if mult < 2.0 else 1e-3
            silently_wrong = ok & ~back & ~(err < eps)
            print("step %.1f / rho, lambda %.0f: handed back %2d of %d, worst error among the kept %.1e"
                  % (mult, lb, back.sum(), len(back), err[ok & ~back].max() if (ok & ~back).any() else 0.0))
            assert not silently_wrong.any(), (mult, lb, np.nonzero(silently_wrong)[0], err[silently_wrong])
            if mult < 2.0 and lb == 0.0:
                assert not back.any(), (mult, np.nonzero(back)[0])
            if mult >= 2.6:
                assert back.any()


def test_partition_for_series_of_600_scans(solver):
    """The reference's demo length (examples/synth_data/deconv.py:46: 600 scans): 12 000 voxels with one scalar lambda at
    the batch's median of 0.13 lambda_max,v -- dense class on whole passes of the split matrix-pipe form, sparse class on
    the pair form over two slots, handed-back problems compacted -- plain, with the cost trace, with the window rule;
    per-problem lambdas; against the vector dispatch (<= 4e-6) and the C oracle (<= 1e-5)."""
    from pybold_amd import data
    V, n_it = 12000, 200
    hrf = orc.spm_hrf(1.0, 1.0, 30.0)[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=10, tr=1.0, hrf=hrf, nb_events=9, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=21)
    assert Y.shape[1] == 600
    lip = 0.9 * orc.spectral_radius_est(orc._MatrixFreeH(hrf), np.random.RandomState(0).randn(600))
    step = 1.0 / lip
    lmax = solver.lambda_max(Y, hrf)
    lam_s = float((0.13 * lmax).median())
    rows = np.random.RandomState(5).choice(V, 128, replace=False)
    ref = _oracle_rows(Y, hrf, lam_s, step, n_it, rows)
    for kw in (dict(), dict(want_J=True), dict(want_J=True, stop="window", tol=1e-6, wind=6)):
        W, J, nd = solver.fista_solve(Y, hrf, lam_s, step, n_it, **kw)
        Wv, Jv, _ = solver.fista_solve(Y, hrf, lam_s, step, n_it, force="valu", **kw)
        assert int(nd.min()) == n_it and int(nd.max()) == n_it, (kw, int(nd.min()))
        e_v = float(((W - Wv).norm(dim=1) / Wv.norm(dim=1).clamp_min(1e-300)).max())
        e_o = rel_rows(W[rows].cpu().numpy(), ref).max()
        print("600 scans, partitioned %-60s vs vector dispatch %.2e, vs oracle %.2e" % (kw, e_v, e_o))
        assert e_v < 4e-6 and e_o < 1e-5, kw
        if J is not None:
            assert float(((J - Jv).abs() / Jv.abs().clamp_min(1e-30)).max()) < 1e-4
    c = torch.tensor([0.02, 0.5], dtype=torch.float64, device=Y.device)
    lam_p = (lmax[:, None] * c[None, :]).reshape(-1)
    Wp, _, ndp = solver.fista_solve(Y, hrf, lam_p, step, n_it, y_rep=2)
    Wpv, _, _ = solver.fista_solve(Y, hrf, lam_p, step, n_it, y_rep=2, force="valu")
    assert int(ndp.min()) == n_it
    assert float(((Wp - Wpv).norm(dim=1) / Wpv.norm(dim=1).clamp_min(1e-300)).max()) < 4e-6
    # the dense half did run on the matrix pipe (other bits than the vector forms), the sparse half on the vector forms
    assert not bool((Wp[0::2] == Wpv[0::2]).all())


def test_backtracked_step_opt_in_mode(solver, golden):
    """SURVEY row g2 (BASELINE's "Lipschitz-backtracked step"; the reference itself has a constant step only): the opt-in
    `pb_fista_solve_backtrack_d` against its own float64 NumPy statement -- the number of step reductions, the final step
    and the iterate of every problem whose acceptance tests all sit away from equality; with a start step <= 1 / L it is
    the constant-step solver bit for bit in its decisions (no reduction) and within 1e-12 in its iterates."""
    g = golden("grid")
    hrf, lip = g["hrf"], float(g["lip_s0"])
    rho = lip / 0.9
    Y = np.stack([g["y_s%d" % s] * a for s in range(4) for a in (1.0, 0.2, -3.0)])
    Yd = torch.from_numpy(Y).cuda()
    for step0, lb in ((1.0 / rho, 1.0), (16.0 / rho, 1.0), (300.0 / rho, 0.1), (5.0 / rho, np.linspace(0.05, 5.0, len(Y)))):
        W, step, halv = solver.fista_solve_backtrack(Yd, hrf, lb, step0, 80)
        Wo, so, ho, margin = orc.fista_backtrack_batch(Y, hrf, lb, step0, 80)
        robust = margin > 1e-9
        assert robust.sum() >= len(Y) - 2, margin
        assert (halv.cpu().numpy()[robust] == ho[robust]).all() and (step.cpu().numpy()[robust] == so[robust]).all()
        assert rel_rows(W.cpu().numpy()[robust], Wo[robust]).max() < 1e-10
        if step0 <= 1.0 / rho:
            assert int(halv.max()) == 0
            Wc, _, _ = solver.fista_solve(Yd, hrf, lb, step0, 80)               # the constant-step float64 solver
            assert rel_rows(W.cpu().numpy(), Wc.cpu().numpy()).max() < 1e-12
        else:
            assert int(halv.min()) >= 2 and float(step.max()) <= 2.0 / rho * 1.0001


def test_plain_pb_fista_solve_partitions_on_the_librarys_own_workspace(solver):
    """`pb_fista_solve` (the round-1 signature, no workspace argument: what a C caller and the torch.ops shim use) partitions
    too, on a workspace the library owns: same bits as `pb_fista_solve_ex` with the caller's workspace, on a mixed batch,
    twice in a row and on a second stream (one workspace per device and stream)."""
    from pybold_amd import _lib, torch_ops
    V, n_it = 20000, 100
    Y, hrf, step = _mixed_batch(V, 31)
    lmax = solver.lambda_max(Y, hrf)
    lam_s = float((0.19 * lmax).median())
    W_ex, _, nd_ex = solver.fista_solve(Y, hrf, lam_s, step, n_it)
    W_np, _, _ = solver.fista_solve(Y, hrf, lam_s, step, n_it, force="nopart")
    assert not torch.equal(W_ex, W_np)                      # (the partition did move the sparse half to other kernels)
    for stream in (torch.cuda.current_stream(), torch.cuda.Stream()):
        with torch.cuda.stream(stream):
            for _ in range(2):
                W, J, nd = torch_ops.fista_solve(Y, hrf, lam_s, step, n_it)     # -> pb_fista_solve
                stream.synchronize()
                assert torch.equal(W, W_ex) and int(nd.min()) == n_it


def test_ill_conditioned_series_are_solved_in_float64(solver, golden):
    """Series the operator H = K_h . cumsum barely sees -- alternating signs, sinusoids of period 3 / 4, high-pass noise,
    such a series plus 1e-3 .. 1e-2 of an ordinary one: coherence lambda_max / (max|y| sum|c|) of 1e-3 .. 4e-3 against
    3e-2 .. 1 for ordinary data -- lose digits in every arithmetic narrower than float64: up to 5e-5 on the matrix pipe
    and 3e-5 on the float32 vector forms (tools/r5_conditioning_probe.py, profiles/r5_conditioning_probe.txt: eps is
    1e-5).  A partitioned call marks them in its lambda_max pass and solves them in float64 (the register-resident
    kernel of fista_exact.h on their list; the LDS kernel for a window other than 6): through the
    DEFAULT dispatch every family is within eps on diff_z, z AND x, at lambda = 0, 0.05 lambda_max and 1, with the cost
    trace and the window rule too; ordinary series next to them keep their kernels (timing: a handful of rows)."""
    g = golden("case1")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    N, step = 300, 1.0 / lip
    t = np.arange(N)
    rng = np.random.RandomState(1)
    alt = np.where(t % 2 == 0, 1.0, -1.0)
    hp = rng.randn(N)
    fams = [g["y"], rng.randn(N), alt, np.sin(2 * np.pi * t / 3), np.sin(2 * np.pi * t / 4), np.sin(2 * np.pi * t / 6),
            np.sin(2 * np.pi * t / 10), np.diff(np.r_[0.0, hp]), np.diff(np.r_[0.0, 0.0, hp], n=2), alt + 1e-3 * g["y"],
            alt + 1e-2 * g["y"], alt + 0.1 * g["y"], alt * (1 + np.sin(2 * np.pi * t / 100)), np.ones(N), 37.0 * alt, 1e-3 * alt]
    Y = np.stack(fams)
    reps = 4096 // len(fams) + 1
    Yd = torch.from_numpy(np.tile(Y, (reps, 1)).astype(np.float32)).cuda()          # 4 112 problems: a partitioned call
    Yo = Yd[:len(fams)].cpu().numpy().astype(np.float64)
    lmax = solver.lambda_max(Yd, hrf)
    worst_default, worst_unguarded = 0.0, 0.0
    for lam in (0.0, 0.05 * lmax, 1.0):
        lam_o = lam[:len(fams)].cpu().numpy() if torch.is_tensor(lam) else lam
        ref = orc.fista_batch(Yo, hrf, lam_o, step, 500)
        xr, zr = orc.fista_outputs(ref, hrf)
        for kw in (dict(), dict(want_J=True, stop="window", tol=1e-7, wind=6), dict(want_J=True, stop="window", tol=1e-7, wind=4)):
            for force in (None, "noill"):
                W, J, nd = solver.fista_solve(Yd, hrf, lam, step, 500, force=force, **kw)
                assert int(nd.min()) == 500
                X, Z = solver.fista_outputs(W[:len(fams)].contiguous(), hrf)
                e = max(rel_rows(W[:len(fams)].cpu().numpy(), ref).max(), rel_rows(Z.cpu().numpy(), zr).max(),
                        rel_rows(X.cpu().numpy(), xr).max())
                if force is None:
                    assert e < 1e-5, (kw, e)
                    worst_default = max(worst_default, e)
                    assert torch.equal(W[:len(fams)], W[len(fams):2 * len(fams)])       # (the same series anywhere in the batch)
                else:
                    worst_unguarded = max(worst_unguarded, e)
    print("ill-conditioned families through the default dispatch: worst %.1e (guard off: %.1e)" % (worst_default, worst_unguarded))
    assert worst_unguarded > 1e-5                       # (the guard is what holds eps here)


# --------------------------------------------------------------------------------------------
# Round 5: series of 641 .. 1 280 scans on the matrix pipe (fista_mfma4.h: one series over the four waves of a workgroup)
@pytest.mark.parametrize("n,k", [(641, 30), (700, 27), (768, 33), (769, 16), (900, 30), (1000, 2), (1024, 30), (1025, 30),
                                 (1200, 28), (1216, 33), (1217, 30), (1279, 30), (1280, 32),
                                 (700, 34), (900, 40), (1200, 42), (1216, 48), (641, 48), (1250, 42), (1280, 48)])
def test_four_wave_matrix_pipe_form_matches_oracle(solver, n, k):
    """`fista_mfma4_kernel`: 16 problems per workgroup of four waves, wave j owning blocks j A .. j A + A - 1 of 32 samples
    (A = ceil(N / 128): 6 .. 10; the last wave holds the end of the series and the padding behind it) -- HCP-length
    runs (examples/icassp_2019/validation.py:41-48).  Warm and cold start against the C float64 oracle, nothing handed
    back on ordinary data, any batch position the same bits; the cost trace; the window rule as a certificate.  HRFs of
    34 .. 48 taps (short TR): three near tiles -- the tiles reach TWO blocks into the neighbouring wave --: plain solves,
    the cost trace, the certificate."""
    from oracle import c_oracle
    rng = np.random.RandomState(n + k)
    hrf = orc.spm_hrf(1.0, 1.0, float(k), False)[0][:k] if k >= 20 else (np.hanning(k + 2)[1:-1] * 0.3 if k > 2 else np.array([0.0, 0.7])[:k])
    assert len(hrf) == k
    lip = orc.gram_lipschitz(hrf, n)
    Yv = rng.randn(40, n)
    W0 = 0.01 * rng.randn(40, n)
    Yh = Yv.astype(np.float32).astype(np.float64)
    Yd, W0d = torch.from_numpy(Yv.astype(np.float32)).cuda(), torch.from_numpy(W0).cuda()
    Wo, _, _ = c_oracle.fista_batch(Yh, hrf, 0.3, 1.0 / lip, 200, W0=W0, threads=4)
    W, _, nd = solver.fista_solve(Yd, hrf, 0.3, 1.0 / lip, 200, W0=W0d, force="mfma2only")
    assert int(nd.min()) == 200 and int(nd.max()) == 200            # nothing handed back
    assert rel_rows(W.cpu().numpy(), Wo).max() < 3e-6
    Wc, Jc, ndc = solver.fista_solve(Yd, hrf, 0.3, 1.0 / lip, 200, force="mfma2only", want_J=True)     # cold start, cost trace
    Woc, Joc, _ = c_oracle.fista_batch(Yh, hrf, 0.3, 1.0 / lip, 200, threads=4, want_J=True)
    assert int(ndc.min()) == 200 and rel_rows(Wc.cpu().numpy(), Woc).max() < 3e-6
    assert np.abs(Jc.cpu().numpy() / Joc - 1.0).max() < 2e-5            # (float32 cost trace, as on the other matrix-pipe forms)
    Wp, _, _ = solver.fista_solve(Yd, hrf, 0.3, 1.0 / lip, 200, force="mfma2only")
    assert torch.equal(Wp, Wc)                                       # (the rotated loop of the cost trace: the same iterate)
    # another batch position, another workgroup: the same bits
    perm = torch.from_numpy(np.r_[np.arange(23, 40), np.arange(23)]).cuda()
    W2, _, _ = solver.fista_solve(Yd[perm].contiguous(), hrf, 0.3, 1.0 / lip, 200, W0=W0d[perm].contiguous(), force="mfma2only")
    assert torch.equal(W2, W[perm])
    # the vector form behind it (remainders, re-solves: one problem per wave, strips of 10 .. 20 samples) against the oracle too
    Wv0, _, _ = solver.fista_solve(Yd, hrf, 0.3, 1.0 / lip, 200, W0=W0d, force="valu")
    assert rel_rows(Wv0.cpu().numpy(), Wo).max() < 3e-6
    # the window rule (far from firing) as a certificate: cleared everywhere; a tolerance it fires at: handed back, re-solved
    if True:                                        # (641 .. 1 280 scans: the re-solve's exact rule holds strips of up to 20 samples per lane)
        Wk, Jk, ndk = solver.fista_solve(Yd, hrf, 0.3, 1.0 / lip, 200, force="mfma2certonly", want_J=True, stop="window", tol=1e-9, wind=6)
        assert int(ndk.min()) == 200 and torch.equal(Wk, Wc)
        # (the exact rule on the one-problem-per-wave form: tests/test_gpu_parity.py pins that one to the reference's stops)
        Wf, Jf, ndf = solver.fista_solve(Yd, hrf, 0.3, 1.0 / lip, 200, force="mfma2cert", want_J=True, stop="window", tol=2e-2, wind=6)
        Wv, Jv, ndv = solver.fista_solve(Yd, hrf, 0.3, 1.0 / lip, 200, force="valu", want_J=True, stop="window", tol=2e-2, wind=6)
        assert torch.equal(ndf, ndv) and int(ndv.min()) < 200 and torch.equal(Wf, Wv)
    # the library's own dispatch: whole passes of 4 096 problems and remainders above 5/8 of one on this form
    assert "four waves" in solver.which_kernel(n, k, 4096)


def test_partition_and_guard_for_series_of_1200_scans(solver):
    """641 .. 1 280 scans through the DEFAULT dispatch (a partitioned call: lambda_max pass with 21-sample strips, dense
    class on whole passes of the four-wave form, sparse class and remainders on the one-problem-per-wave form, ill-conditioned
    series in float64): a mixed batch -- per-problem lambda on both sides of the class boundary, an alternating series, an
    all-zero one -- against the C oracle; with the cost trace and the window rule too."""
    from oracle import c_oracle
    from pybold_amd import data
    n, P = 1200, 4500
    hrf = orc.spm_hrf(1.0, 1.0, 28.0, False)[0][:28]
    lip = orc.gram_lipschitz(hrf, n)
    Y, _, _ = data.gen_rnd_bloc_bold_batch(P, dur=n / 60.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=11)
    Y = Y[:, :n].contiguous()
    Y[17] = torch.from_numpy(np.where(np.arange(n) % 2 == 0, 1.0, -1.0)).float().cuda()
    Y[33] = 0.0
    lmax = solver.lambda_max(Y, hrf)
    rng = np.random.RandomState(0)
    frac = np.where(rng.rand(P) < 0.7, 0.02, 0.5)
    frac[:64] = np.where(np.arange(64) % 2 == 0, 0.02, 0.5)
    lam = torch.from_numpy(frac).cuda() * lmax
    lam[17], lam[33] = 0.1, 1.0
    idx = np.r_[np.arange(64), rng.choice(np.arange(64, P), 64, replace=False)]
    Yo = Y[idx].cpu().numpy().astype(np.float64)
    Wo, Jo, _ = c_oracle.fista_batch(Yo, hrf, lam[idx].cpu().numpy(), 1.0 / lip, 150, threads=8, want_J=True)
    nz = np.linalg.norm(Wo, axis=1) > 0
    for kw in (dict(), dict(want_J=True), dict(want_J=True, stop="window", tol=1e-8, wind=6)):
        W, J, nd = solver.fista_solve(Y, hrf, lam, 1.0 / lip, 150, **kw)
        # (the all-zero series meets the window rule at once: 0 / (0 + 1e-10) < tol at the first test, bold_signal.py:86-95)
        expect = torch.full_like(nd, 150)
        if "stop" in kw:
            expect[33] = 8
        bad = torch.nonzero(nd != expect).flatten()
        assert len(bad) == 0, (kw, bad[:16].tolist(), nd[bad[:16]].tolist(), (lam / lmax)[bad[:16]].tolist())
        Wn = W[idx].cpu().numpy()
        assert rel_rows(Wn[nz], Wo[nz]).max() < 1e-5 and np.abs(Wn[~nz]).max() == 0.0, kw
        if J is not None:
            assert np.abs(J[idx].cpu().numpy()[nz] / Jo[nz] - 1.0).max() < 5e-5
    # one scalar lambda: everything dense but the two odd rows
    W, _, nd = solver.fista_solve(Y, hrf, 0.05, 1.0 / lip, 150)
    Wo1, _, _ = c_oracle.fista_batch(Yo, hrf, 0.05, 1.0 / lip, 150, threads=8)
    assert int(nd.min()) == 150 and rel_rows(W[idx].cpu().numpy()[nz], Wo1[nz]).max() < 1e-5


@pytest.mark.parametrize("n,k", [(600, 30), (330, 27), (640, 33), (1200, 28), (700, 30), (600, 42), (640, 48), (1200, 42), (300, 42)])
def test_loops_rule_inside_the_split_forms(solver, n, k):
    """`_loops_deconv`'s criterion (pybold/bold_signal.py:267-273) in full inside `fista_mfma2_kernel<..., LOOPS>` (311 .. 640
    scans) and `fista_mfma4_kernel<..., LOOPS>` (641 .. 1 280): every wave adds up its share of the two float64 norms beside the
    update, the shares meet in LDS at the barrier that ends the iteration; a problem that meets the rule is written out at
    that moment and its lanes keep iterating.  Stop iterations and iterates against the float64 oracle on series whose
    stops spread over the run (a criterion within ~1e-6 of `tol` may cross one iteration apart: +-1 on 2 % at most);
    through the library's own dispatch too (whole passes on the split form, the rest on the one-problem-per-wave form)."""
    rng = np.random.RandomState(n + k)
    hrf = orc.spm_hrf(1.0, 1.0, float(k), False)[0][:k]
    lip = orc.gram_lipschitz(hrf, n)
    V = 96
    Z = np.zeros((V, n))
    for v in range(V):
        for _ in range(max(5, n // 60)):
            o = rng.randint(0, n - 20)
            Z[v, o:o + rng.randint(8, 16)] = 1.0
    X = orc.causal_conv(hrf, Z)
    noise = rng.randn(V, n)
    noise *= (np.linalg.norm(X, axis=1) / np.linalg.norm(noise, axis=1) / np.sqrt(10 ** (np.linspace(-5, 20, V) / 10)))[:, None]
    Y = torch.from_numpy((X + noise).astype(np.float32)).cuda()
    Yh = Y.cpu().numpy().astype(np.float64)
    # (the rule compares (1 + beta) ||clamp(u, +-th)|| with ||w'||: it fires where lambda is small against the signal)
    lmed = float(np.median(orc.lambda_max(Yh, hrf)))
    stops = set()
    for lbda, tol in ((0.003 * lmed, 1e-3), (0.003 * lmed, 7e-4)):
        Wo, ndo = orc.loops_batch(Yh, hrf, lbda, 1.0 / lip, 300, tol)
        W, _, nd = solver.fista_solve(Y, hrf, lbda, 1.0 / lip, 300, stop="loops", tol=tol, force="mfma2only")
        ndn = nd.cpu().numpy()
        same = ndn == ndo
        assert np.abs(ndn - ndo).max() <= 1 and same.mean() >= 0.97, (lbda, tol, ndn, ndo)       # (nothing handed back: dense solutions)
        assert rel_rows(W.cpu().numpy()[same], Wo[same]).max() < 1e-5
        stops |= set(ndo.tolist())
    assert len(stops) > 10 and min(stops) < 300
    # the default dispatch at a batch size that fills passes
    reps = 6000 // V + 1
    Yl = Y.repeat(reps, 1)[:6000].contiguous()
    assert ("split over" in solver.which_kernel(n, k, 6000, stop="loops")) or ("four waves" in solver.which_kernel(n, k, 6000, stop="loops"))
    Wl, _, ndl = solver.fista_solve(Yl, hrf, lbda, 1.0 / lip, 300, stop="loops", tol=tol)
    ref_nd = ndo[np.arange(6000) % V]
    assert np.abs(ndl.cpu().numpy() - ref_nd).max() <= 1 and (ndl.cpu().numpy() == ref_nd).mean() >= 0.97
    ok = (ndl.cpu().numpy() == ref_nd)[:V]
    assert rel_rows(Wl[:V].cpu().numpy()[ok], Wo[ok]).max() < 1e-5


@pytest.mark.parametrize("n,k,P", [(600, 27, 8192 + 150), (1200, 28, 4096 + 90), (400, 20, 6000), (600, 40, 8192 + 150), (1000, 44, 4096 + 90)])
def test_shared_hrf_z_step_of_long_series_on_the_split_forms(solver, n, k, P):
    """`pb_fista_solve_pp` with ONE HRF and step in device memory (the blind step's z-step, bd_shared) at 321 .. 1 280 scans:
    whole passes on `fista_mfma2_kernel<..., TAPS_DEV>` / `fista_mfma4_kernel<..., TAPS_DEV>`, the rest and what the guards
    hand back on the one-problem-per-wave form -- against the float64 oracle and against the vector dispatch; HRFs of 34+ taps
    with three near tiles (the cumulative taps beyond lag 63 built on the device)."""
    rng = np.random.RandomState(n + k)
    h = orc.spm_hrf(0.9, 20.0 / k, 20.0, False)[0][:k]            # (a 20 s HRF sampled at k points)
    assert len(h) == k
    lip = orc.gram_lipschitz(h, n)
    Yb = torch.from_numpy(rng.randn(P, n).astype(np.float32)).cuda()
    taps, stepc = torch.from_numpy(h.copy()).cuda(), torch.from_numpy(np.array([1.0 / lip])).cuda()
    lam = 0.02 * float(np.median(orc.lambda_max(Yb[:64].cpu().numpy().astype(np.float64), h)))
    Wp, ndp = solver.fista_solve_pp(Yb, taps, stepc, lam, 60)
    assert int(ndp.min()) == 60
    idx = np.r_[0, 15, 16, P - 1, rng.choice(P, 12, replace=False)]
    Wop = orc.fista_batch(Yb.cpu().numpy()[idx].astype(np.float64), h, lam, 1.0 / lip, 60)
    assert rel_rows(Wp.cpu().numpy()[idx], Wop).max() < 1e-5
    Wv, _ = solver.fista_solve_pp(Yb, taps, stepc, lam, 60, force="valu")
    d = ((Wp - Wv).norm(dim=1) / Wv.norm(dim=1)).cpu().numpy()
    assert d.max() < 4e-6 and d[:4096].max() > 0.0            # (and it is not the vector form that ran the whole passes)


def test_bd_shared_graph_on_a_long_series(solver):
    """`BdSharedGraph` at 1 000 scans: the z-steps run on `fista_mfma4_kernel<..., TAPS_DEV>` (more than 64 KB of dynamic LDS,
    asked for inside the capture) -- the captured loop replays to the eager loop's bits and follows the vector dispatch."""
    from pybold_amd import data, distributed
    from pybold_amd.hrf_model import spm_hrf
    t_r, dur, V, n = 0.75, 20.0, 4500, 1000
    h_true = spm_hrf(0.8, t_r, dur, False)[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=n * t_r / 60.0, tr=t_r, hrf=h_true, nb_events=12, avg_dur=12.0,
                                           std_dur=1.0, snr=10.0, seed=3)
    Y = Y[:, :n].contiguous()
    We, he, de = distributed.bd_shared(Y, t_r, lbda=1.7, hrf_dur=dur, nb_iter=3, nb_inner=30)
    runner = distributed.BdSharedGraph(Y, t_r, lbda=1.7, hrf_dur=dur, nb_iter=3, nb_inner=30)
    assert runner.graph is not None, runner.fallback
    for _ in range(2):
        runner.launch()
    W, h, d = runner.result()
    assert torch.equal(W, We) and np.array_equal(d["theta"], de["theta"])
    assert de["theta"][-1] < de["theta"][0]
    assert "four waves" in solver.which_kernel(n, len(h_true), V)


def test_regularisation_path_of_a_long_series(solver):
    """BASELINE config 5's call shape (one series, a grid of lambdas: y shared by `y_rep` problems) at 700 scans: the partitioned
    dispatch puts the dense end of every path on the four-wave form, the sparse end on the one-problem-per-wave form; every
    problem against the C oracle."""
    from oracle import c_oracle
    from pybold_amd import data
    n, V, L = 700, 300, 16
    hrf = orc.spm_hrf(1.0, 1.0, 28.0, False)[0][:28]
    lip = orc.gram_lipschitz(hrf, n)
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=n / 60.0, tr=1.0, hrf=hrf, nb_events=8, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=5)
    Y = Y[:, :n].contiguous()
    lmax = solver.lambda_max(Y, hrf)
    grid = torch.logspace(-2.0, 0.0, L, dtype=torch.float64, device="cuda")
    lam = (lmax[:, None] * grid[None, :]).reshape(-1)
    W, _, nd = solver.fista_solve(Y, hrf, lam, 1.0 / lip, 200, y_rep=L)
    assert int(nd.min()) == 200 and int(nd.max()) == 200
    idx = np.r_[np.arange(2 * L), np.random.RandomState(0).choice(V * L, 96, replace=False)]
    Yo = Y.cpu().numpy().astype(np.float64)[idx // L]
    Wo, _, _ = c_oracle.fista_batch(Yo, hrf, lam[idx].cpu().numpy(), 1.0 / lip, 200, threads=8)
    # (at lambda = lambda_max the prox is 0 at every iteration, the iterate is not: the reference's momentum runs on the
    # gradient-step point, w_{k+1} = -beta_k u_k there -- pybold/bold_signal.py:65,72; the oracle restates that)
    Wn = W[idx].cpu().numpy()
    assert rel_rows(Wn, Wo).max() < 1e-5
    Wv, _, _ = solver.fista_solve(Y, hrf, lam, 1.0 / lip, 200, y_rep=L, force="valu")
    d = ((W - Wv).norm(dim=1) / (Wv.norm(dim=1) + 1e-300)).cpu().numpy()
    assert d.max() < 2e-5 and (d > 0).mean() > 0.3                 # (a good part of the path ran on the matrix pipe)


@pytest.mark.parametrize("case", ["hcp", "long42"])
def test_long_series_against_the_reference_itself(solver, golden, case):
    """The REAL reference's `deconv` on 1 200 scans (TR 0.72 s, 20 s HRF = 28 taps: examples/icassp_2019/validation.py:41-48) and on
    900 scans with a 42-tap HRF (tests/golden/make_golden_r5_long.py) against `fista_mfma4_kernel` -- two near tiles / three --:
    iterate, outputs, cost trace; the window rule through the certificate + exact re-solve stops where the reference stopped."""
    g = golden("long_series")
    y, hrf, lip = g[case + "_y"], g[case + "_hrf"], float(g[case + "_lipschitz"])
    n = len(y)
    assert "four waves" in solver.which_kernel(n, len(hrf), 4096, want_J=True, stop="window")
    Yd = torch.from_numpy(np.tile(y.astype(np.float32), (40, 1))).cuda()
    y32 = y.astype(np.float32).astype(np.float64)
    in_err = np.linalg.norm(y32 - y) / np.linalg.norm(y)            # (the batch API takes float32 series)
    for lbda in (0.5, 2.0):
        tag = "%s_l%g_n100" % (case, lbda)
        W, J, nd = solver.fista_solve(Yd, hrf, lbda, 1.0 / lip, 100, want_J=True, force="mfma2only")
        assert int(nd.min()) == 100
        X, Z = solver.fista_outputs(W, hrf)
        for a, k in ((W, "dz_"), (Z, "z_"), (X, "x_")):
            e = np.linalg.norm(a[7].cpu().numpy() - g[k + tag]) / np.linalg.norm(g[k + tag])
            assert e < 1e-5, (tag, k, e, in_err)
        Jn = J[7].cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(Jn / Jn[0], g["J_" + tag], rtol=5e-5)
        tag = "%s_l%g_n400_es" % (case, lbda)
        Ww, Jw, ndw = solver.fista_solve(Yd, hrf, lbda, 1.0 / lip, 400, want_J=True, stop="window", tol=1e-2, wind=6, force="mfma2cert")
        assert int(ndw.min()) == int(ndw.max()) == len(g["J_" + tag])
        e = np.linalg.norm(Ww[7].cpu().numpy() - g["dz_" + tag]) / np.linalg.norm(g["dz_" + tag])
        assert e < 1e-5, (tag, e)
