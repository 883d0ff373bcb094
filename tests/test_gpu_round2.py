"""GPU tests of the round-2 additions, all through the C ABI: empty batches in every entry
point, the all-float64 forms (1-D calls of the reference API), lambda_max, inf_norm and the
theta-step on normal equations (pb_hrf_normal_eq + pb_theta_fit)."""
import numpy as np
import pytest
import torch

from oracle import pybold_oracle as orc

pytestmark = pytest.mark.gpu

T_R, HRF_DUR = 0.75, 20.0


def rel_rows(a, b):
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    return (np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-300)).max()


def dev32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def dev64(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


@pytest.fixture(scope="module")
def solver():
    from pybold_amd import solver
    assert torch.cuda.is_available()
    return solver


def blind_problem(n_vox=8, n=300, seed=0, theta=0.8, noise=0.05):
    rng = np.random.RandomState(seed)
    h = orc.spm_hrf(theta, T_R, HRF_DUR, False)[0]
    Z = np.zeros((n_vox, n))
    for v in range(n_vox):
        for _ in range(5):
            o = rng.randint(0, n - 30)
            Z[v, o:o + rng.randint(8, 20)] += 1.0
    Y = orc.causal_conv(h, Z) + noise * rng.randn(n_vox, n)
    return Z, Y, h


def test_empty_batch_in_every_wrapper(solver, golden):
    """V = 0 / P = 0 / M = 0 is a no-op in every entry point (torch hands out NULL data
    pointers for empty tensors); a shard may be empty (distributed.shard_bounds)."""
    hrf = golden("case1")["hrf"]
    N, K = 300, len(hrf)
    Y0, Y0d = dev32(np.zeros((0, N))), dev64(np.zeros((0, N)))
    W0 = dev64(np.zeros((0, N)))
    for Y in (Y0, Y0d):
        W, J, nd = solver.fista_solve(Y, hrf, 1.0, 1e-6, 5, want_J=True, stop="window", tol=1e-3)
        assert W.shape == (0, N) and J.shape == (0, 5) and nd.shape == (0,)
        r2, l1 = solver.fista_stats(W0, Y, hrf)
        assert r2.shape == (0,) and l1.shape == (0,)
        assert solver.hrf_cost(W0, Y, np.stack([hrf, hrf])).shape == (2, 0)
        assert solver.hrf_cost_pv(W0, Y, dev64(np.zeros((3, 0, K)))).shape == (3, 0)
        assert solver.lambda_max(Y, hrf).shape == (0,)
        ne = solver.hrf_normal_eq(W0, Y, K)
        assert ne.shape == (K * K + K + 1,) and float(ne.abs().max()) == 0.0     # zeros, not garbage
        assert solver.hrf_normal_eq(W0, Y, K, per_voxel=True).shape == (0, K * K + K + 1)
    for fn in (solver.integ_op, solver.integ_adj):
        assert fn(W0).shape == (0, N)
    for fn in (solver.conv, solver.corr, solver.op_forward, solver.op_adjoint):
        assert fn(W0, hrf).shape == (0, N)
    X, Z = solver.fista_outputs(W0, hrf)
    assert X.shape == (0, N) and Z.shape == (0, N)
    X, Z = solver.fista_outputs_pp(W0, dev64(np.zeros((0, K))))
    assert X.shape == (0, N)
    W, nd = solver.fista_solve_pp(Y0, dev64(np.zeros((0, K))), dev64(np.zeros(0)), 1.0, 5)
    assert W.shape == (0, N)
    W, nd = solver.fista_solve_pp(Y0, dev64(hrf), dev64([1e-6]), 1.0, 5)      # shared taps, no voxel
    assert W.shape == (0, N)
    assert solver.gram_frobenius_batch(dev64(np.zeros((0, K))), N).shape == (0,)
    assert solver.spm_hrf_batch(dev64(np.zeros(0)), T_R, HRF_DUR).shape[0] == 0
    assert solver.inf_norm_rows(W0).shape == (0, N)
    th, c, t = solver.theta_fit(dev64(np.zeros((0, 27 * 27 + 27 + 1))), T_R, HRF_DUR, (0.6, 1.9))
    assert th.shape == (0,) and t.shape == (0, 27)
    # the batched blind solver and the auto-lambda branch on an empty batch
    from pybold_amd import blind
    X, Z, W, taps, d = blind.bd_batch(Y0, T_R, nb_iter=2)
    assert X.shape == (0, N) and taps.shape[0] == 0


@pytest.mark.parametrize("force", ["fast", "generic"])
def test_float64_solver_is_the_reference_arithmetic(solver, golden, force):
    """pb_fista_solve_d: float64 end to end -- register-resident one-problem-per-wave kernel
    ("fast") and LDS kernel ("generic") -- golden grid at 1e-10 (the float32-FIR kernels are
    held to 1e-5), cost trace at 1e-11."""
    g = golden("grid")
    Y = np.stack([g["y_s%d" % s] for s in range(4)])
    hrf, lip = g["hrf"], float(g["lip_s0"])
    for lbda in (0.1, 1.0, 10.0):
        for nit in (1, 2, 3, 10, 500):
            W, J, nd = solver.fista_solve(dev64(Y), hrf, lbda, 1.0 / lip, nit, want_J=True, force=force)
            ref = np.stack([g["dz_s%d_l%g_n%d" % (s, lbda, nit)] for s in range(4)])
            assert rel_rows(W.cpu().numpy(), ref) < 1e-10, (lbda, nit)
            assert J.dtype == torch.float64
            if nit == 10 and lbda == 1.0:
                for s in range(4):
                    Js = J[s].cpu().numpy()
                    np.testing.assert_allclose(Js / Js[0], g["J_s%d_l1_n10" % s], rtol=1e-11)
    g1 = golden("case1")
    W = dev64(g1["diff_z"][None])
    r2, l1 = solver.fista_stats(W, dev64(g1["y"][None]), g1["hrf"])
    assert float(r2[0]) == pytest.approx(np.sum(np.square(g1["x"] - g1["y"])), rel=1e-12)
    with pytest.raises(ValueError):
        solver.fista_solve(dev64(Y), hrf, 1.0, 1.0 / lip, 5, force="fast2")


@pytest.mark.parametrize("force", ["fast", "generic"])
def test_float64_stop_rules_and_shapes(solver, golden, force):
    """Both float64 kernels: golden stop iterations of the window rule (26/77/191/311) and of
    the _loops_deconv rule, cost trace, per-problem lambda with shared y rows, warm start,
    series of 600 scans (S = 10 entry) and shapes without a register-resident entry."""
    from pybold_amd._lib import PyboldHipError
    g = golden("early_stop")
    y, hrf, lip = g["y"], g["hrf"], float(g["lipschitz"])
    Yb = np.stack([y, 3.0 * y, y])
    for tol, n_ref in ((0.1, 26), (0.03, 77), (0.01, 191), (0.005, 311)):
        W, J, nd = solver.fista_solve(dev64(Yb), hrf, 1.0, 1.0 / lip, 1000, want_J=True, stop="window",
                                      tol=tol, wind=6, force=force)
        nd = nd.cpu().numpy()
        assert nd[0] == n_ref and nd[2] == n_ref, (force, tol, nd)
        assert rel_rows(W.cpu().numpy()[[0, 2]], np.stack([g["dz_%g" % tol]] * 2)) < 1e-10
        Jn = J.cpu().numpy()
        assert np.isfinite(Jn[0, :n_ref]).all() and np.isnan(Jn[0, n_ref:]).all()
    gl = golden("loops_deconv")
    yl, hl = gl["y"], gl["h"]
    from pybold_amd.utils import gram_frobenius
    stepl = 1.0 / gram_frobenius(hl, len(yl))
    for nit in (1, 2, 5, 100):
        for es_on, tol in ((False, 1e-12), (True, 1e-2), (True, 1e-3)):
            W, _, _ = solver.fista_solve(dev64(yl[None]), hl, 1.7, stepl, nit, stop="loops" if es_on else None,
                                         tol=tol, force=force)
            assert rel_rows(W.cpu().numpy(), gl["w_n%d_es%d_tol%g" % (nit, es_on, tol)][None]) < 1e-9, (nit, es_on, tol)
    W, _, _ = solver.fista_solve(dev64(yl[None]), hl, 1.7, stepl, 5, W0=dev64(gl["w0"][None]), force=force)
    assert rel_rows(W.cpu().numpy(), gl["w_warm_n5"][None]) < 1e-9
    # per-problem lambda, shared rows
    gg = golden("grid")
    Y = np.stack([gg["y_s%d" % s] for s in range(4)])
    lam = np.tile([0.1, 1.0, 10.0], 4)
    W, _, _ = solver.fista_solve(dev64(Y), gg["hrf"], lam, 1.0 / float(gg["lip_s0"]), 10, y_rep=3, force=force)
    W = W.cpu().numpy().reshape(4, 3, -1)
    for s in range(4):
        for i, lb in enumerate((0.1, 1.0, 10.0)):
            assert rel_rows(W[s, i], gg["dz_s%d_l%g_n10" % (s, lb)]) < 1e-10
    # other shapes: N = 600 (S = 10), N = 37 / K = 5, and beyond the register-resident table
    rng = np.random.RandomState(7)
    for N, K in ((600, 30), (37, 5), (640, 32), (700, 30), (300, 40)):
        hk = rng.randn(K) * 0.3
        Yk = rng.randn(6, N)
        A = orc.toeplitz_from_kernel(hk, N, N).dot(np.tril(np.ones((N, N))))
        lk = 1.05 * np.linalg.norm(A, 2) ** 2
        ref = orc.fista_batch(Yk, hk, 0.05, 1.0 / lk, 30)
        has_entry = N <= 640 and K <= 32
        if force == "fast" and not has_entry:
            with pytest.raises(PyboldHipError):
                solver.fista_solve(dev64(Yk), hk, 0.05, 1.0 / lk, 30, force="fast")
            continue
        W, _, _ = solver.fista_solve(dev64(Yk), hk, 0.05, 1.0 / lk, 30, force=force)
        assert np.abs(W.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-10, (force, N, K)


def test_one_d_api_calls_match_goldens_tightly(golden):
    """1-D calls of the reference API run float64 end to end: deconv / _loops_deconv /
    hrf_fit_err reproduce the goldens at 1e-9 (batches: 1e-5)."""
    import pybold_amd
    g = golden("case1")
    np.random.seed(0)
    x, z, dz, J, _, _ = pybold_amd.deconv(g["y"], 1.0, g["hrf"], lbda=1.0, nb_iter=500,
                                          early_stopping=False)
    errs = (rel_rows(dz, g["diff_z"]), rel_rows(z, g["z"]), rel_rows(x, g["x"]))
    print("1-D deconv vs golden (diff_z, z, x):", errs)
    assert max(errs) < 1e-9
    np.testing.assert_allclose(J, g["J"], rtol=1e-10)
    gl = golden("loops_deconv")
    H = pybold_amd.toeplitz_from_kernel(gl["h"], len(gl["y"]), len(gl["y"]))
    w = pybold_amd._loops_deconv(gl["y"], np.zeros_like(gl["y"]), H, 1.7, 100, False, 4, 1e-12)
    assert rel_rows(w, gl["w_n100_es0_tol1e-12"]) < 1e-9
    gh = golden("hrf_estim")
    for theta, err in zip(gh["thetas"], gh["errs"]):
        got = pybold_amd.hrf_fit_err(theta, gh["z"], gh["y"], float(gh["t_r"]), float(gh["hrf_dur"]))
        assert got == pytest.approx(err, rel=1e-12)


def test_lambda_max(solver, golden):
    g = golden("grid")
    Y = np.stack([g["y_s%d" % s] for s in range(4)])
    hrf, lip = g["hrf"], float(g["lip_s0"])
    ref = orc.lambda_max(Y, hrf)
    np.testing.assert_allclose(solver.lambda_max(dev64(Y), hrf).cpu().numpy(), ref, rtol=1e-12)
    lm = solver.lambda_max(dev32(Y), hrf)
    np.testing.assert_allclose(lm.cpu().numpy(), orc.lambda_max(Y.astype(np.float32), hrf), rtol=1e-12)
    # at lambda_max the first prox step thresholds everything (float32 FIRs: margin 1e-5;
    # the float64 kernel: 1e-12); just below it something survives
    W, _, _ = solver.fista_solve(dev32(Y), hrf, lm * (1 + 1e-5), 1.0 / lip, 1)
    assert float(W.abs().max()) == 0.0
    W, _, _ = solver.fista_solve(dev64(Y), hrf, solver.lambda_max(dev64(Y), hrf) * (1 + 1e-12), 1.0 / lip, 1)
    assert float(W.abs().max()) == 0.0
    W, _, _ = solver.fista_solve(dev32(Y), hrf, lm * 0.99, 1.0 / lip, 1)
    assert (W.abs().amax(dim=1) > 0).all()


def test_inf_norm_like_reference():
    """pybold/utils.py:112-138 (and pybold/tests/test_utils.py:8-15: max |inf_norm(x)| == 1):
    1-D, 2-D along both axes, 3-D, lists; NumPy in -> NumPy out, CUDA in -> CUDA out."""
    from pybold_amd.utils import inf_norm
    rng = np.random.RandomState(3)
    a1, a2, a3 = rng.randn(1000) * 7, rng.randn(37, 300) * 3, rng.randn(4, 5, 60)
    np.testing.assert_allclose(inf_norm(a1), orc.inf_norm(a1), rtol=1e-15)
    assert np.max(np.abs(inf_norm(a1))) == pytest.approx(1.0, abs=1e-7)
    np.testing.assert_allclose(inf_norm(a2), orc.inf_norm(a2), rtol=1e-15)
    np.testing.assert_allclose(inf_norm(a2, axis=0), orc.inf_norm(a2, axis=0), rtol=1e-15)
    np.testing.assert_allclose(inf_norm(a3), orc.inf_norm(a3), rtol=1e-15)
    out = inf_norm([a1, a2, a3])
    ref = orc.inf_norm([a1, a2, a3])
    assert isinstance(out, list) and all(np.allclose(o, r, rtol=1e-15) for o, r in zip(out, ref))
    t = inf_norm(dev64(a2))
    assert torch.is_tensor(t) and t.is_cuda
    np.testing.assert_allclose(t.cpu().numpy(), orc.inf_norm(a2), rtol=1e-15)
    z = inf_norm(np.zeros(10))
    assert np.all(z == 0.0)
    bad = a1.copy()
    bad[5] = np.nan
    assert np.isnan(inf_norm(bad)).all()            # np.max propagates NaN
    with pytest.raises(ValueError):
        inf_norm(np.zeros((2, 2, 2, 2)))


@pytest.mark.parametrize("n,K_dur", [(300, 20.0), (240, 20.0), (100, 15.0), (20, 20.0)])
def test_normal_equations_match_dense_oracle(solver, n, K_dur):
    t_hrf = solver.hrf_sample_times(T_R, K_dur)
    K = len(t_hrf)
    rng = np.random.RandomState(n)
    Z = np.cumsum((rng.rand(11, n) < 0.06) * rng.randn(11, n), axis=1)
    Y = rng.randn(11, n)
    G, b, yy = orc.hrf_normal_eq(Z, Y, K)
    ref = np.concatenate([G.ravel(), b, [yy]])
    ne = solver.hrf_normal_eq(dev64(Z), dev64(Y), K).cpu().numpy()
    np.testing.assert_allclose(ne, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    ne32 = solver.hrf_normal_eq(dev64(Z), dev32(Y), K).cpu().numpy()
    np.testing.assert_allclose(ne32, ref, rtol=1e-5, atol=1e-6 * np.abs(ref).max())
    pv = solver.hrf_normal_eq(dev64(Z), dev64(Y), K, per_voxel=True).cpu().numpy()
    for v in range(11):
        Gv, bv, yyv = orc.hrf_normal_eq(Z[v], Y[v], K)
        refv = np.concatenate([Gv.ravel(), bv, [yyv]])
        np.testing.assert_allclose(pv[v], refv, rtol=1e-12, atol=1e-12 * np.abs(refv).max())
    np.testing.assert_allclose(pv.sum(axis=0), ref, rtol=1e-11, atol=1e-11 * np.abs(ref).max())
    # a tiny work buffer (one block) gives the same sums
    one = solver.hrf_normal_eq(dev64(Z), dev64(Y), K,
                               work=torch.empty(K * K + K + 1, dtype=torch.float64, device="cuda"))
    np.testing.assert_allclose(one.cpu().numpy(), ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())


def test_theta_fit_matches_direct_minimiser_and_lbfgsb(solver):
    from scipy.optimize import fmin_l_bfgs_b
    Z, Y, h_true = blind_problem()
    K = len(h_true)
    ne = solver.hrf_normal_eq(dev64(Z), dev64(Y), K)
    theta, cost, taps = solver.theta_fit(ne, T_R, HRF_DUR, (0.6, 1.9))
    th_ref, f_ref = orc.shared_theta_argmin(Z, Y, T_R, HRF_DUR, (0.6, 1.9))
    print("shared theta: device %.10f direct-cost oracle %.10f" % (float(theta[0]), th_ref))
    assert float(theta[0]) == pytest.approx(th_ref, abs=2e-7)
    assert float(cost[0]) == pytest.approx(f_ref, rel=1e-8)
    np.testing.assert_allclose(taps[0].cpu().numpy(), orc.spm_hrf(float(theta[0]), T_R, HRF_DUR, False)[0],
                               rtol=1e-10, atol=1e-14)
    G, b, yy = orc.hrf_normal_eq(Z, Y, K)
    th_o, f_o, _ = orc.theta_fit_normal_eq(G, b, yy, T_R, HRF_DUR, (0.6, 1.9))
    assert float(theta[0]) == pytest.approx(th_o, abs=1e-9)       # the restated algorithm itself
    # one set per voxel = the reference's per-voxel theta-step (bold_signal.py:329-333)
    pv = solver.hrf_normal_eq(dev64(Z), dev64(Y), K, per_voxel=True)
    thetas, costs, _ = solver.theta_fit(pv, T_R, HRF_DUR, (0.6, 1.9))
    for v in range(len(Z)):
        t_ref, f_l, _ = fmin_l_bfgs_b(func=orc.hrf_fit_err, x0=1.9, args=(Z[v], Y[v], T_R, HRF_DUR),
                                      bounds=[(0.6, 1.9)], approx_grad=True, maxiter=999, pgtol=1e-12)
        assert float(thetas[v]) == pytest.approx(float(t_ref[0]), abs=5e-5)   # L-BFGS-B's own accuracy
        assert float(costs[v]) <= float(f_l) * (1 + 1e-9)
        assert float(costs[v]) == pytest.approx(orc.hrf_fit_err(float(thetas[v]), Z[v], Y[v], T_R, HRF_DUR),
                                                rel=1e-8)
    # accuracy levels: one scan of 64 candidates (spacing 0.02) + parabola, then / 56 per level
    for n_refine, tol in ((1, 2e-3), (2, 2e-6), (4, 2e-7)):
        th_n, f_n, _ = solver.theta_fit(ne, T_R, HRF_DUR, (0.6, 1.9), n_refine=n_refine)
        assert float(th_n[0]) == pytest.approx(th_ref, abs=tol), n_refine
        assert float(f_n[0]) >= f_ref * (1 - 1e-9)
    # minimiser on a bound
    th_b, _, _ = solver.theta_fit(ne, T_R, HRF_DUR, (1.2, 1.9))
    assert float(th_b[0]) == pytest.approx(1.2, abs=1e-12)


def test_shared_taps_solver_equals_host_taps_solver(solver, golden):
    """pb_fista_solve_pp with ldt = 0 (taps and step read from device memory, shared by all
    problems) is bitwise the per-problem-taps kernel with equal rows and 1e-5 from the oracle."""
    g = golden("loops_deconv")
    rng = np.random.RandomState(0)
    Y = g["y"][None] * (0.5 + rng.rand(37, 1))
    h = g["h"]
    from pybold_amd.utils import gram_frobenius
    step = 1.0 / gram_frobenius(h, Y.shape[1])
    Wp, _ = solver.fista_solve_pp(dev32(Y), dev64(np.tile(h, (37, 1))), dev64(np.full(37, step)), 1.7, 50,
                                  force="fast1")
    W1, _ = solver.fista_solve_pp(dev32(Y), dev64(h), dev64([step]), 1.7, 50, force="fast1")
    assert torch.equal(W1, Wp)                     # the same kernel form: bitwise
    Wd, _ = solver.fista_solve_pp(dev32(Y), dev64(np.tile(h, (37, 1))), dev64(np.full(37, step)), 1.7, 50)
    assert rel_rows(Wd.cpu().numpy(), Wp.cpu().numpy()) < 1e-6    # library dispatch: one problem per wave here
    Ws, _ = solver.fista_solve_pp(dev32(Y), dev64(h), dev64([step]), 1.7, 50)   # library dispatch (one per wave here)
    assert rel_rows(Ws.cpu().numpy(), Wp.cpu().numpy()) < 1e-6
    ref = orc.fista_batch(Y.astype(np.float32).astype(np.float64), h, 1.7, step, 50)
    assert rel_rows(Ws.cpu().numpy(), ref) < 1e-5
    Wg, _ = solver.fista_solve_pp(dev32(Y), dev64(h), dev64([step]), 1.7, 50, force="generic")
    assert rel_rows(Wg.cpu().numpy(), ref) < 1e-6
    # inplace=True iterates in the caller's buffer
    W0 = torch.zeros((37, Y.shape[1]), dtype=torch.float64, device="cuda")
    Wi, _ = solver.fista_solve_pp(dev32(Y), dev64(h), dev64([step]), 1.7, 50, W0=W0, inplace=True)
    assert Wi.data_ptr() == W0.data_ptr() and torch.equal(Wi, Ws)


def test_bd_shared_device_theta_step(solver):
    """Config-4 structure on a small batch: device theta-step vs the L-BFGS-B variant and
    vs the oracle run of the same loop; V = 1 equals the reference-style per-voxel fit."""
    from pybold_amd import distributed
    Z, Y, h_true = blind_problem(n_vox=64, noise=0.1)
    Yd = dev32(Y)
    kw = dict(lbda=0.5, hrf_dur=HRF_DUR, nb_iter=6, nb_inner=60)
    W, h, d = distributed.bd_shared(Yd, T_R, **kw)
    W2, h2, d2 = distributed.bd_shared(Yd, T_R, theta_solver="lbfgsb", **kw)
    print("bd_shared theta device:", d["theta"], "\n          theta lbfgsb:", d2["theta"])
    assert (np.diff(d["J"]) < 0).all()
    np.testing.assert_allclose(d["theta"], d2["theta"], atol=2e-4)     # L-BFGS-B stops early (factr 1e7)
    np.testing.assert_allclose(d["J"], d2["J"], rtol=1e-5)
    assert rel_rows(h, h2) < 1e-3
    assert d["J"][-1] <= d2["J"][-1] * (1 + 1e-7)                      # never a worse minimiser
    assert 0.8 < d["theta"][-1] < d["theta"][1] <= 1.9        # from theta_0 = 2.0 towards the true 0.8


def test_vector_theta0_in_bd_batch(solver):
    from pybold_amd import blind
    Z, Y, _ = blind_problem(n_vox=6)
    t0 = np.linspace(0.9, 1.8, 6)
    X, Zd, W, taps, d = blind.bd_batch(dev32(Y), T_R, lbda=0.5, theta_0=t0, hrf_dur=HRF_DUR, nb_iter=2)
    X1, _, W1, _, d1 = blind.bd_batch(dev32(Y[2:3]), T_R, lbda=0.5, theta_0=float(t0[2]),
                                      hrf_dur=HRF_DUR, nb_iter=2)
    assert torch.equal(W[2:3], W1)                       # voxel 2 is independent of its neighbours
    with pytest.raises(ValueError):
        blind.bd_batch(dev32(Y), T_R, theta_0=np.ones(5), nb_iter=1)
    with pytest.raises(ValueError):
        blind.bd_batch(dev32(Y), T_R, theta_0=np.full(6, 2.5), nb_iter=1)


def test_deconv_auto_lambda_at_reference_defaults(golden):
    """deconv(lbda=None) with the reference's DEFAULT settings (nb_iter=1000,
    nb_sub_iter=1000, tol=1e-6, wind=6: pybold/bold_signal.py:13-14) -- up to 1000 outer x
    1000 inner iterations per voxel, some voxels leaving early on the alpha rule -- against the
    C form of the oracle's restatement of :99-214 (same noise level).  1-D and a 12-voxel
    batch, both on the float64 kernel (the stop decisions of this branch sit on a knife edge,
    see bold_signal._deconv_auto_lbda).  The update alpha += mu (||x-y||^2 - N sigma^2),
    lbda = 1/(2 alpha) is itself unstable when alpha comes close to 0 (lbda explodes and
    amplifies rounding noise by 1/alpha): voxels whose lambda stays below 20 are compared
    tightly, the others only loosely.  (The branch stays parity-unpinned: no reference-side
    vectors exist, see DESIGN.)"""
    import time
    import pybold_amd
    from oracle import c_oracle
    g = golden("grid")
    hrf, lip = g["hrf"], float(g["lip_s0"])
    Y = np.stack([g["y_s%d" % (1 + s % 3)] * (1.0 + 0.05 * (s // 3)) for s in range(12)])
    sigma = np.array([orc.mad_daub_noise_est(y) for y in Y])
    Wo, Jo, Ro, Go, n_outer = c_oracle.deconv_auto_lbda_batch(Y, hrf, sigma, lip, threads=16)
    lam = (Jo - 0.5 * Ro) / Go
    robust = np.nanmax(lam, axis=1) < 20.0
    assert robust.sum() >= 6 and n_outer[0] == 1000 and n_outer.min() > 6
    t0 = time.perf_counter()
    np.random.seed(0)
    x, z, dz, J, R, G = pybold_amd.deconv(Y[0], 1.0, hrf, lbda=None)
    t1 = time.perf_counter()
    assert isinstance(J, list) and len(J) == 1000
    e = rel_rows(dz, Wo[0])
    print("deconv(lbda=None) defaults, 1-D: %.1f s, rel err diff_z %.2e, J %.2e"
          % (t1 - t0, e, np.abs(np.array(J) / Jo[0] - 1).max()))
    assert e < 1e-8
    np.testing.assert_allclose(J, Jo[0], rtol=1e-9)
    np.random.seed(0)
    X, Z, W, Jb, Rb, Gb = pybold_amd.deconv(Y, 1.0, hrf, lbda=None)
    t2 = time.perf_counter()
    errs = np.linalg.norm(W - Wo, axis=1) / np.linalg.norm(Wo, axis=1)
    print("deconv(lbda=None) defaults, 12-voxel batch: %.1f s, rel err diff_z per voxel %s (robust: %s)"
          % (t2 - t1, np.array2string(errs, precision=1), robust.astype(int)))
    assert Jb.shape == (1000, 12) and errs[robust].max() < 1e-8 and np.isfinite(W).all()
    # the same voxels leave the outer loop at the same outer iteration (NaN padding after it)
    np.testing.assert_array_equal(np.isnan(Jb.T)[robust], np.isnan(Jo)[robust])
    np.testing.assert_array_equal((~np.isnan(Jb.T)).sum(axis=1)[robust], n_outer[robust])
    np.testing.assert_allclose(Jb.T[robust], Jo[robust], rtol=1e-8)


def test_shared_taps_pair_kernel_at_scale(solver, golden):
    """Shared HRF read from device memory on a machine-filling batch: the pair kernel with
    fast FIRs carries the whole rounds, the per-problem-taps kernel the remainder; both against
    the C oracle and against the host-taps solver."""
    from oracle import c_oracle
    from pybold_amd.utils import gram_frobenius
    g = golden("loops_deconv")
    h = g["h"]                                   # K = 27, h[0] = 0
    rng = np.random.RandomState(1)
    P, N = 20000, 300
    Y = rng.randn(P, N).astype(np.float32)
    step = 1.0 / gram_frobenius(h, N)
    Yd = torch.from_numpy(Y).cuda()
    Ws, _ = solver.fista_solve_pp(Yd, dev64(h), dev64([step]), 1.7, 60)
    # (same arithmetic on both sides: the host-taps call without the device-side partition of round 5 -- white noise at
    # lambda = 1.7 straddles the class boundary, and the partitioned call solves the sparse half on the float32 forms)
    Wh, _, _ = solver.fista_solve(Yd, h, 1.7, step, 60, force="nopart")
    assert rel_rows(Ws.cpu().numpy(), Wh.cpu().numpy()) < 1e-6
    idx = np.concatenate([rng.choice(16384, 24, replace=False), 16384 + rng.choice(P - 16384, 24, replace=False)])
    Wo, _, _ = c_oracle.fista_batch(Y[idx].astype(np.float64), h, 1.7, step, 60, threads=8)
    assert rel_rows(Ws.cpu().numpy()[idx], Wo) < 1e-5
    Wd, _, _ = solver.fista_solve(Yd, h, 1.7, step, 60)                    # the default (partitioned) dispatch
    assert rel_rows(Wd.cpu().numpy()[idx], Wo) < 1e-5
    for force in ("fast2", "fast1"):
        Wf, _ = solver.fista_solve_pp(Yd, dev64(h), dev64([step]), 1.7, 60, force=force)
        assert rel_rows(Wf.cpu().numpy()[idx], Wo) < 1e-5, force


def orc_fista_one(y, hrf, lbda, step, n_iter, w0):
    from oracle import pybold_oracle as orc
    return orc.fista_batch(y[None], hrf, lbda, step, n_iter, W0=w0[None])[0]


def test_side_stream_remainder(solver, golden):
    """The plan of the VECTOR forms (force="valu": since round 3 plain solves of this shape go to the
    matrix-pipe form above half a round).  8 192 < P < 16 384 problems: half a round of pair waves on the caller's stream with the
    remainder beside it on the library's side stream (fork/join by events).  Where the
    one-stream plan uses the same pieces the results are bitwise equal; results are complete in
    the caller's stream order (consumed at once, no synchronisation); repeated calls; a
    non-default caller stream; graph capture falls back to one stream."""
    from oracle import c_oracle
    g = golden("case1")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    step = 1.0 / lip
    rng = np.random.RandomState(3)
    for P in (8200, 10000, 12288, 12500, 12800):
        Y = torch.from_numpy(rng.randn(P, 300).astype(np.float32)).cuda()
        n_main, main_k, tail_k = solver.launch_plan(300, 30, P, force="valu")
        assert n_main == 8192 and "two problems per row" in main_k
        W, _, _ = solver.fista_solve(Y, hrf, 1.0, step, 40, force="valu")
        s1 = W.abs().sum()                       # consumer on the caller's stream, right away
        Wq, _, _ = solver.fista_solve(Y, hrf, 1.0, step, 40, force="valuseq")
        # (the one-stream plan uses the same pieces up to 12 288 problems, other kernels beyond)
        if P <= 12288:
            assert torch.equal(W, Wq), P
            assert float(s1) == float(Wq.abs().sum())
        else:
            assert float(((W - Wq).norm(dim=1) / Wq.norm(dim=1)).max()) < 1e-6
            assert float(s1) == float(W.abs().sum())
        idx = np.concatenate([rng.choice(8192, 8, replace=False), 8192 + rng.choice(P - 8192, 8, replace=False)])
        Wo, _, _ = c_oracle.fista_batch(Y.cpu().numpy()[idx].astype(np.float64), hrf, 1.0, step, 40, threads=4)
        assert rel_rows(W.cpu().numpy()[idx], Wo) < 1e-5
    # a quarter to three eighths of a round (alone, or after whole rounds): single-row waves with
    # one-problem waves beside them
    for P in (4500, 6144, 21000):
        Y = torch.from_numpy(rng.randn(P, 300).astype(np.float32)).cuda()
        n_main, main_k, tail_k = solver.launch_plan(300, 30, P, force="valu")
        assert n_main == (16384 if P == 21000 else 4096) and ("one problem per wave" in tail_k or P == 21000)
        W, _, _ = solver.fista_solve(Y, hrf, 1.0, step, 30, force="valu")
        s1 = W.abs().sum()
        Wq, _, _ = solver.fista_solve(Y, hrf, 1.0, step, 30, force="valuseq")
        assert float(((W - Wq).norm(dim=1) / Wq.norm(dim=1)).max()) < 1e-6 and float(s1) == float(W.abs().sum())
        idx = np.r_[0, 4095, 4096, P - 1, rng.choice(P, 6, replace=False)]
        Wo, _, _ = c_oracle.fista_batch(Y.cpu().numpy()[idx].astype(np.float64), hrf, 1.0, step, 30, threads=4)
        assert rel_rows(W.cpu().numpy()[idx], Wo) < 1e-5
    # whole rounds first, then the concurrent group closes the plan (same forms per problem as
    # the one-stream plan: 24 576 on the pair kernel, 424 one per wave)
    for P in (25000, 41500):
        Y = torch.from_numpy(rng.randn(P, 300).astype(np.float32)).cuda()
        n_main, main_k, tail_k = solver.launch_plan(300, 30, P, force="valu")
        assert n_main == (P // 8192) * 8192 and "two problems per row" in main_k and "one problem per wave" in tail_k
        W, _, _ = solver.fista_solve(Y, hrf, 1.0, step, 30, force="valu")
        s1 = W.abs().sum()
        Wq, _, _ = solver.fista_solve(Y, hrf, 1.0, step, 30, force="valuseq")
        assert torch.equal(W, Wq) and float(s1) == float(Wq.abs().sum()), P
    # the cost trace through the concurrent group (every piece writes its own rows of J)
    Y = torch.from_numpy(rng.randn(12500, 300).astype(np.float32)).cuda()
    W, J, _ = solver.fista_solve(Y, hrf, 1.0, step, 30, want_J=True, force="valu")
    Wq, Jq, _ = solver.fista_solve(Y, hrf, 1.0, step, 30, want_J=True, force="valuseq")
    assert bool(torch.isfinite(J).all())
    assert float(((W - Wq).norm(dim=1) / Wq.norm(dim=1)).max()) < 1e-6
    assert float(((J - Jq).abs() / Jq.abs()).max()) < 1e-5
    idx = np.r_[0, 8191, 8192, 12287, 12288, 12499]
    Wo, Jo, _ = c_oracle.fista_batch(Y.cpu().numpy()[idx].astype(np.float64), hrf, 1.0, step, 30, want_J=True, threads=4)
    assert rel_rows(W.cpu().numpy()[idx], Wo) < 1e-5
    assert np.abs(J.cpu().numpy()[idx] / Jo - 1.0).max() < 1e-5
    # shared y (y_rep), per-problem lambda and a warm start across the split: 600 voxels x 20
    # lambdas = 12 000 problems, the pieces start at problem offsets inside a voxel's group
    Yv = torch.from_numpy(rng.randn(600, 300).astype(np.float32)).cuda()
    lam = np.tile(np.logspace(-2, 0, 20), 600)
    W0 = torch.from_numpy(0.01 * rng.randn(12000, 300)).cuda()
    W, _, n_done = solver.fista_solve(Yv, hrf, lam, step, 30, W0=W0, y_rep=20, force="valu")
    Wq, _, _ = solver.fista_solve(Yv, hrf, lam, step, 30, W0=W0, y_rep=20, force="valuseq")
    assert torch.equal(W, Wq) and int(n_done.min()) == 30
    idx = np.r_[0, 8191, 8192, 8193, 11999, rng.choice(12000, 8, replace=False)]
    Wo = np.stack([orc_fista_one(Yv[i // 20].cpu().numpy().astype(np.float64), hrf, lam[i], step, 30,
                                 W0[i].cpu().numpy()) for i in idx])
    assert rel_rows(W.cpu().numpy()[idx], Wo) < 1e-5
    # back-to-back calls on a non-default stream, each result consumed in stream order
    Y = torch.from_numpy(rng.randn(10000, 300).astype(np.float32)).cuda()
    ref, _, _ = solver.fista_solve(Y, hrf, 1.0, step, 25, force="valuseq")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    sums = []
    with torch.cuda.stream(st):
        for _ in range(5):
            W, _, _ = solver.fista_solve(Y, hrf, 1.0, step, 25, force="valu")
            sums.append((W - ref).abs().max())
    st.synchronize()
    assert all(float(s) == 0.0 for s in sums)
    # graph capture: no cross-stream work is captured, the replay equals the eager result
    plan = solver.FistaPlan(Y, hrf, 1.0, step, 25, force="valu")
    plan.run()
    torch.cuda.synchronize()
    eager = plan.W.clone()
    cs = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cs):
        plan.run()
        cs.synchronize()
        with torch.cuda.graph(graph, stream=cs):
            plan.run()
    plan.W.fill_(3.0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(plan.W, eager) and torch.equal(eager, ref)


def test_host_pipeline_matches_resident_solve():
    """Chunked H2D / solve / D2H on three streams (solver.HostPipeline) against the
    device-resident solve of the same rows: identical where a chunk is laid out like the
    whole batch (whole rounds of the pair kernel), 1e-6 where the remainder forms differ;
    ragged last chunk, per-problem lambda, float32 / float64 / device-resident outputs."""
    from pybold_amd import solver
    from pybold_amd.hrf_model import spm_hrf
    hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
    step = 1.0 / 723876.27
    g = torch.Generator(device="cuda").manual_seed(5)
    V, N, n_iter = 40000, 300, 60
    Y = torch.randn(V, N, device="cuda", generator=g)
    lam = 0.2 + torch.rand(V, device="cuda", generator=g, dtype=torch.float64)
    Yh = Y.cpu().pin_memory()
    ref, _, _ = solver.fista_solve(Y, hrf, lam, step, n_iter)
    scale = float(ref.abs().max())
    for out_dtype, chunk in ((torch.float64, 16384), (torch.float32, 16384), (None, 16384), (torch.float64, 5000)):
        pipe = solver.HostPipeline(V, N, hrf, lam, step, n_iter, chunk=chunk, out_dtype=out_dtype)
        for _ in range(2):                                  # slots and events are reusable
            out = pipe.run(Yh)
        got = out.cuda().double() if out_dtype is not None else out
        err = float((got - ref).abs().max()) / scale
        assert err < (1e-6 if out_dtype != torch.float32 else 2e-6), (out_dtype, chunk, err)
        # (until round 4 the first 32 768 rows ran the same kernel form either way and compared bitwise; a partitioned call
        # -- white noise at lambda ~ 0.7: mixed classes, half of the series kept off the matrix pipe by the conditioning
        # guard -- places a row by its position in the call's lists: chunked and whole calls agree to the forms' accuracy)
    # enqueue-only form is ordered before later work on the current stream
    assert solver.round_size(N, len(hrf)) == 16384
    pipe = solver.HostPipeline(V, N, hrf, lam, step, n_iter, out_dtype=None)
    assert pipe.chunk == V                                   # under three rounds: one chunk, one stream
    assert solver.HostPipeline(60000, N, hrf, 1.0, step, n_iter, out_dtype=None).chunk == 16384
    W = pipe.run(Yh, sync=False)
    tot = W.abs().sum()
    assert abs(float(tot) - float(ref.abs().sum())) < 1e-6 * float(ref.abs().sum())
    with pytest.raises(ValueError):
        pipe.run(Yh[:10])


def test_c_abi_from_plain_c(solver, golden, tmp_path):
    """examples/c_abi_demo (C99, linked against libpybold_hip.so and the HIP runtime only; built
    by __graft_entry__.build()) runs pb_fista_solve without Python or PyTorch in the process:
    its iterates equal those of the Python host side bit for bit."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "c_abi_demo")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build_c_demo()
    g = golden("case1")
    hrf, lip = np.ascontiguousarray(g["hrf"], dtype=np.float64), float(g["lipschitz"])
    rng = np.random.RandomState(8)
    V, N, n_iter = 1000, 300, 50
    Y = rng.randn(V, N).astype(np.float32)
    Y.tofile(tmp_path / "y.f32")
    hrf.tofile(tmp_path / "taps.f64")
    out = subprocess.run([exe, str(tmp_path / "y.f32"), str(V), str(N), str(tmp_path / "taps.f64"), str(len(hrf)),
                          repr(1.0 / lip), "1.0", str(n_iter), str(tmp_path / "w.f64")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "sum|diff_z|" in out.stdout
    Wc = np.fromfile(tmp_path / "w.f64", dtype=np.float64).reshape(V, N)
    W, _, _ = solver.fista_solve(torch.from_numpy(Y).cuda(), hrf, 1.0, 1.0 / lip, n_iter)
    assert np.array_equal(Wc, W.cpu().numpy())
    # error path: text from pb_last_error, non-zero exit code
    bad = subprocess.run([exe, str(tmp_path / "y.f32"), str(V), str(N), str(tmp_path / "taps.f64"), str(len(hrf)),
                          "-1.0", "1.0", str(n_iter), str(tmp_path / "w2.f64")], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 1 and "step must be positive" in bad.stderr


@pytest.mark.parametrize("force", [None, "fast1", "fast2", "fast2d", "wide", "generic", "f64", "f64generic"])
def test_cold_start_flag_equals_zero_warm_start(solver, golden, force):
    """PB_FLAG_COLD_START (W0=None: the kernels start from w = 0 without reading w_dev, the
    caller does not clear it) against an explicit all-zero warm start: bitwise, every kernel
    form, cost trace and stop rule included; FistaPlan.run() on a buffer full of NaNs."""
    g = golden("case1")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    rng = np.random.RandomState(12)
    V = 300 if force in ("generic", "f64generic") else 9000
    Y = torch.from_numpy(rng.randn(V, 300).astype(np.float32)).cuda()
    kw = dict(want_J=True, stop="window", tol=1e-3)
    if force in ("f64", "f64generic"):
        Y = Y.double()
        f = None if force == "f64" else "generic"
    else:
        f = force
    zeros = torch.zeros((V, 300), dtype=torch.float64, device="cuda")
    for extra in ({}, kw):
        if extra and f in ("fast2", "fast2d"):
            continue                                       # the pair form has no stop rule
        Wc, Jc, nc = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, 40, force=f, **extra)
        Ww, Jw, nw = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, 40, W0=zeros, force=f, **extra)
        assert torch.equal(Wc, Ww) and torch.equal(nc, nw)
        if Jc is not None:
            assert torch.equal(torch.nan_to_num(Jc), torch.nan_to_num(Jw))
    if force not in ("f64", "f64generic"):
        plan = solver.FistaPlan(Y, hrf, 1.0, 1.0 / lip, 40, force=f)
        plan.W.fill_(float("nan"))
        plan.run()
        ref, _, _ = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, 40, W0=zeros, force=f)
        assert torch.equal(plan.W, ref)
        plan.launch()                                      # continues from the current iterate
        ref2, _, _ = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, 40, W0=ref, force=f)
        assert torch.equal(plan.W, ref2)


@pytest.mark.parametrize("n,K", [(1200, 5), (37, 1), (600, 64), (300, 127), (9, 9), (2000, 33), (1025, 8), (64, 17)])
def test_normal_equations_any_tap_count(solver, n, K):
    """Shapes that move the thread layout of normal_eq_sum_kernel: K <= 8 (a role spans two
    waves), K = 127 (32 roles of 8 threads), series longer than the prefetch registers hold,
    series shorter than the HRF."""
    rng = np.random.RandomState(n + K)
    V = 7
    Z = np.cumsum((rng.rand(V, n) < 0.1) * rng.randn(V, n), axis=1)
    Y = rng.randn(V, n)
    G, b, yy = orc.hrf_normal_eq(Z, Y, K)
    ref = np.concatenate([G.ravel(), b, [yy]])
    ne = solver.hrf_normal_eq(dev64(Z), dev64(Y), K).cpu().numpy()
    np.testing.assert_allclose(ne, ref, rtol=1e-11, atol=1e-12 * np.abs(ref).max())
    pv = solver.hrf_normal_eq(dev64(Z), dev64(Y), K, per_voxel=True).cpu().numpy()
    np.testing.assert_allclose(pv.sum(axis=0), ref, rtol=1e-10, atol=1e-11 * np.abs(ref).max())


@pytest.mark.parametrize("force", [None, "fast1", "fast2", "wide", "generic", "f64"])
def test_padded_rows_and_views(solver, golden, force):
    """Leading dimensions larger than the series (row views of wider buffers, the layout an
    allocator with padded pitches hands over): y float32 / float64 with ld > N, per-problem
    lambda as a strided view, warm start from a view -- same results as contiguous copies,
    nothing written outside the N columns."""
    g = golden("case1")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    rng = np.random.RandomState(21)
    V, N, pad = 523, 300, 21
    f64 = force == "f64"
    dt = torch.float64 if f64 else torch.float32
    buf = torch.from_numpy(rng.randn(V, N + pad)).to(dt).cuda()
    Yv = buf[:, 3:3 + N]                                            # stride(0) = N + pad, offset 3
    assert Yv.stride(0) == N + pad and not Yv.is_contiguous()
    lam_buf = torch.from_numpy(0.2 + rng.rand(V, 2)).cuda()
    lam = lam_buf[:, 1]                                             # strided 1-D view
    W0buf = torch.from_numpy(0.01 * rng.randn(V, N + 5)).cuda()
    W0 = W0buf[:, :N]
    f = None if f64 else force
    Wv, Jv, _ = solver.fista_solve(Yv, hrf, lam, 1.0 / lip, 35, W0=W0, want_J=True, force=f)
    Wc, Jc, _ = solver.fista_solve(Yv.contiguous(), hrf, lam.contiguous(), 1.0 / lip, 35, W0=W0.contiguous(),
                                   want_J=True, force=f)
    assert torch.equal(Wv, Wc) and torch.equal(Jv, Jc)
    assert torch.equal(W0buf[:, :N], W0) and torch.equal(buf[:, 3:3 + N], Yv)     # inputs untouched
    ref = orc.fista_batch(Yv.cpu().numpy().astype(np.float64), hrf, lam.cpu().numpy(), 1.0 / lip, 35,
                          W0=W0.cpu().numpy())
    assert rel_rows(Wv.cpu().numpy(), ref) < (1e-10 if f64 else 1e-5)
    # outputs / stats / operators on views
    X, Z = solver.fista_outputs(Wv, hrf)
    np.testing.assert_allclose(Z.cpu().numpy(), np.cumsum(Wv.cpu().numpy(), axis=1), rtol=1e-12, atol=1e-12)
    lm_v = solver.lambda_max(Yv, hrf)
    lm_c = solver.lambda_max(Yv.contiguous(), hrf)
    assert torch.equal(lm_v, lm_c)


def test_two_host_threads_share_the_side_stream(solver, golden):
    """Two host threads, each on a stream of its own, solving batches that use the library's
    (one per device) side stream: the fork/join is serialised inside the library, every result
    equals the one-stream result of the same input."""
    import threading
    g = golden("case1")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    gens = [torch.Generator(device="cuda").manual_seed(s) for s in (1, 2)]
    Ys = [torch.randn(10000 + 1250 * i, 300, device="cuda", generator=gens[i]) for i in range(2)]
    refs = [solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, 20, force="seq")[0] for Y in Ys]
    torch.cuda.synchronize()
    errs = [[], []]

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(8):
                    W, _, _ = solver.fista_solve(Ys[i], hrf, 1.0, 1.0 / lip, 20)
                    errs[i].append(float((W - refs[i]).abs().max() / refs[i].abs().max()))
            st.synchronize()
        except Exception as e:                                   # surfaced below
            errs[i].append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(2):
        assert len(errs[i]) == 8 and all(isinstance(e, float) and e < 1e-6 for e in errs[i]), errs[i]
