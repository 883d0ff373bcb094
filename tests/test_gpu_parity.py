"""GPU parity tests: the HIP kernels, called through the C ABI
(libpybold_hip.so via pybold_amd._lib), against the float64 CPU oracle and the
golden vectors captured from the reference.

Tolerance (north_star): relative L2 error <= 1e-5 per voxel on diff_z, z, x
(float32 FIR/scans on device, float64 iterate; the reference is float64).
The operator surface (.op/.adj) is float64 on device and held to 1e-12.
"""
import numpy as np
import pytest
import torch

from oracle import pybold_oracle as orc

pytestmark = pytest.mark.gpu

EPS = 1.0e-5


def rel_rows(a, b):
    a, b = np.atleast_2d(a), np.atleast_2d(b)
    return (np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-300)).max()


@pytest.fixture(scope="module")
def pa():
    import pybold_amd
    from pybold_amd import solver
    assert torch.cuda.is_available()
    return pybold_amd, solver


def dev32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def dev64(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def grid_inputs(golden):
    g = golden("grid")
    Y = np.stack([g["y_s%d" % s] for s in range(4)])
    return g, Y, g["hrf"], float(g["lip_s0"])


@pytest.mark.parametrize("force", ["fast1", "fast2", "fast2d", "generic"])
def test_fista_vs_golden_grid(pa, golden, force):
    """lambda x iterations grid of SURVEY 8c case 3 (captures k=0 and the aliasing)."""
    _, solver = pa
    g, Y, hrf, lip = grid_inputs(golden)
    for lbda in (0.1, 1.0, 10.0):
        for nit in (1, 2, 3, 10, 500):
            W, _, n_done = solver.fista_solve(dev32(Y), hrf, lbda, 1.0 / lip, nit, force=force)
            W = W.cpu().numpy()
            ref = np.stack([g["dz_s%d_l%g_n%d" % (s, lbda, nit)] for s in range(4)])
            assert rel_rows(W, ref) < EPS, (force, lbda, nit, rel_rows(W, ref))
            assert (n_done.cpu().numpy() == nit).all()
            # and against the oracle's batched form
            Wo = orc.fista_batch(Y, hrf, lbda, 1.0 / lip, nit)
            assert rel_rows(W, Wo) < EPS


@pytest.mark.parametrize("force", ["fast1", "fast2", "fast2d", "generic"])
def test_outputs_and_cost_trace(pa, golden, force):
    _, solver = pa
    g = golden("case1")
    y, hrf, lip = g["y"], g["hrf"], float(g["lipschitz"])
    Yb = np.stack([y, 0.5 * y, y])           # voxel 0 and 2 = the golden case (A and B slots)
    W, J, _ = solver.fista_solve(dev32(Yb), hrf, 1.0, 1.0 / lip, 500, want_J=True, force=force)
    Jall = J.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(Jall[2] / Jall[2][0], g["J"], rtol=2e-5)
    W, J = W[:1].contiguous(), J[:1]
    X, Z = solver.fista_outputs(W, hrf)
    assert rel_rows(W.cpu().numpy(), g["diff_z"]) < EPS
    assert rel_rows(Z.cpu().numpy(), g["z"]) < EPS
    assert rel_rows(X.cpu().numpy(), g["x"]) < EPS
    J = J.cpu().numpy()[0].astype(np.float64)
    np.testing.assert_allclose(J / J[0], g["J"], rtol=2e-5)


def test_per_problem_lambda_and_y_rep(pa, golden):
    """Regularisation path: several lambdas per voxel share one y row (config 5)."""
    _, solver = pa
    g, Y, hrf, lip = grid_inputs(golden)
    lbdas = np.array([0.1, 1.0, 10.0])
    lam = np.tile(lbdas, Y.shape[0])
    W, _, _ = solver.fista_solve(dev32(Y), hrf, lam, 1.0 / lip, 10, y_rep=3, force="fast")
    W = W.cpu().numpy().reshape(Y.shape[0], 3, -1)
    for s in range(4):
        for i, lbda in enumerate(lbdas):
            assert rel_rows(W[s, i], g["dz_s%d_l%g_n10" % (s, lbda)]) < EPS


def test_warm_start_chunks_equal_one_run(pa, golden):
    """Two launches of 250 iterations (momentum table offset by the caller) would
    differ from one of 500; a warm start with restarted momentum must equal the
    oracle's warm-started run."""
    _, solver = pa
    g, Y, hrf, lip = grid_inputs(golden)
    W1, _, _ = solver.fista_solve(dev32(Y), hrf, 1.0, 1.0 / lip, 7, force="fast")
    W2, _, _ = solver.fista_solve(dev32(Y), hrf, 1.0, 1.0 / lip, 5, W0=W1, force="fast")
    Wo = orc.fista_batch(Y, hrf, 1.0, 1.0 / lip, 7)
    Wo = orc.fista_batch(Y, hrf, 1.0, 1.0 / lip, 5, W0=Wo)
    assert rel_rows(W2.cpu().numpy(), Wo) < EPS


@pytest.mark.parametrize("N,K", [(300, 30), (290, 30), (304, 30), (17, 5), (40, 30), (300, 1),
                                 (33, 27), (1, 1), (16, 2), (600, 30), (284, 28), (384, 32),
                                 (240, 27), (160, 30), (128, 16), (300, 27), (300, 32), (608, 29),
                                 (700, 30), (300, 40), (300, 48), (150, 45), (1000, 41), (300, 49)])
def test_edge_shapes_fast_and_generic_agree_with_oracle(pa, N, K):
    _, solver = pa
    rng = np.random.RandomState(N * 100 + K)
    hrf = rng.randn(K) * 0.3
    Y = rng.randn(5, N)
    H = orc.DenseH(hrf, N, N)
    lip = 1.1 * np.linalg.norm(H.K.dot(np.tril(np.ones((N, N)))), 2) ** 2 + 1e-12
    Wo = orc.fista_batch(Y.astype(np.float32).astype(np.float64), hrf, 0.05, 1.0 / lip, 40)
    for force in ("fast1", "fast2", "fast2d", "generic"):
        if force != "generic" and not solver.has_fast_path(N, K):
            continue
        W, _, _ = solver.fista_solve(dev32(Y), hrf, 0.05, 1.0 / lip, 40, force=force)
        err = np.abs(W.cpu().numpy() - Wo).max() / (np.abs(Wo).max() + 1e-30)
        assert err < EPS, (force, N, K, err)


@pytest.mark.parametrize("N,K", [(609, 30), (700, 30), (1200, 28), (1216, 30), (1217, 32), (2000, 5),
                                 (2432, 32), (768, 32), (650, 1)])
def test_long_series_one_problem_per_wave(pa, N, K):
    """Series longer than 608 scans (e.g. HCP runs of 1 200 frames): the 64-lane form of the
    register-resident kernel (wave_shr/shl:1 halo chains, row_bcast scans) vs the oracle
    and vs the generic kernel, plain and with the cost trace."""
    _, solver = pa
    assert solver.has_fast_path(N, K)
    assert "one problem per wave" in solver.which_kernel(N, K, 7)
    rng = np.random.RandomState(N + K)
    hrf = rng.randn(K) * 0.3
    Y = rng.randn(7, N)
    Y32 = Y.astype(np.float32).astype(np.float64)
    A = orc.toeplitz_from_kernel(hrf, N, N).dot(np.tril(np.ones((N, N))))
    lip = 1.05 * np.linalg.norm(A, 2) ** 2
    lam = 0.05 * (0.5 + rng.rand(7))
    ref = orc.fista_batch(Y32, hrf, lam, 1.0 / lip, 30)
    W, J, n_done = solver.fista_solve(dev32(Y), hrf, lam, 1.0 / lip, 30, want_J=True, force="fast")
    assert np.abs(W.cpu().numpy() - ref).max() / np.abs(ref).max() < EPS
    assert (n_done.cpu().numpy() == 30).all()
    Wg, Jg, _ = solver.fista_solve(dev32(Y), hrf, lam, 1.0 / lip, 30, want_J=True, force="generic")
    np.testing.assert_allclose(J.cpu().numpy(), Jg.cpu().numpy(), rtol=2e-5)
    W2, _, _ = solver.fista_solve(dev32(Y), hrf, lam, 1.0 / lip, 30, force="fast")
    assert np.abs(W2.cpu().numpy() - ref).max() / np.abs(ref).max() < EPS
    # stop rules in the one-problem-per-wave form agree with the LDS kernel (float64)
    for stop, tol in (("loops", 0.3), ("window", 0.05)):
        if stop == "window" and N > 1216:
            continue                     # needs the S = 38 entry: window rule stays on the LDS kernel
        Wf, _, nf = solver.fista_solve(dev32(Y), hrf, lam, 1.0 / lip, 60, stop=stop, tol=tol, force="fast")
        Wr, _, nr = solver.fista_solve(dev32(Y), hrf, lam, 1.0 / lip, 60, stop=stop, tol=tol, force="generic")
        assert torch.equal(nf, nr), (stop, nf, nr)
        assert np.abs(Wf.cpu().numpy() - Wr.cpu().numpy()).max() / np.abs(ref).max() < EPS
        assert int(nf.min()) < 60        # the rule actually fired


def test_lambda_zero_and_huge(pa, golden):
    """lambda = 0 (no prox) and a lambda that thresholds everything: because the
    momentum uses the gradient-step point the iterate is then
    -beta_k * u_k, not 0 -- exactly what the reference returns."""
    _, solver = pa
    g, Y, hrf, lip = grid_inputs(golden)
    Y32 = Y.astype(np.float32).astype(np.float64)
    for lbda in (0.0, 1e9):
        for force in ("fast1", "fast2", "fast2d", "generic"):
            W, _, _ = solver.fista_solve(dev32(Y), hrf, lbda, 1.0 / lip, 20, force=force)
            Wo = orc.fista_batch(Y32, hrf, lbda, 1.0 / lip, 20)
            assert rel_rows(W.cpu().numpy(), Wo) < EPS, (lbda, force)
    W, _, _ = solver.fista_solve(dev32(Y), hrf, 1e9, 1.0 / lip, 1)
    assert float(W.abs().max()) == 0.0        # beta_0 = 0: first iterate is the prox point


def test_many_problems_partial_last_block(pa, golden):
    """P not a multiple of 16 (nor of 2): idle rows in the last workgroup, and an
    unpaired last problem in the two-problems-per-row kernel."""
    _, solver = pa
    g, Y, hrf, lip = grid_inputs(golden)
    Yb = np.tile(Y, (9, 1))[:35] * np.linspace(0.5, 1.5, 35)[:, None]
    Wo = orc.fista_batch(Yb.astype(np.float32).astype(np.float64), hrf, 1.0, 1.0 / lip, 25)
    for force in ("fast1", "fast2", "fast2d"):
        W, _, n_done = solver.fista_solve(dev32(Yb), hrf, 1.0, 1.0 / lip, 25, force=force)
        assert rel_rows(W.cpu().numpy(), Wo) < EPS, force
        assert (n_done.cpu().numpy() == 25).all()
    # per-problem lambda and shared y rows through the pair kernel
    lam = np.tile([0.1, 1.0, 10.0], 35)
    Wref = orc.fista_batch(np.repeat(Yb, 3, axis=0).astype(np.float32).astype(np.float64), hrf, lam,
                           1.0 / lip, 10)
    for force in ("fast2", "fast2d", "fast1", "wide", None):
        W, _, _ = solver.fista_solve(dev32(Yb), hrf, lam, 1.0 / lip, 10, y_rep=3, force=force)
        assert rel_rows(W.cpu().numpy(), Wref) < EPS, force


def test_loops_deconv_goldens(pa, golden):
    pybold_amd, _ = pa
    g = golden("loops_deconv")
    y, h = g["y"], g["h"]
    H = pybold_amd.toeplitz_from_kernel(h, len(y), len(y))
    for nit in (1, 2, 5, 100):
        for es_on, tol in ((False, 1e-12), (True, 1e-2), (True, 1e-3)):
            w = pybold_amd._loops_deconv(y, np.zeros_like(y), H, 1.7, nit, es_on, 4, tol)
            ref = g["w_n%d_es%d_tol%g" % (nit, es_on, tol)]
            assert rel_rows(w, ref) < EPS, (nit, es_on, tol)
    w = pybold_amd._loops_deconv(y, g["w0"], H, 1.7, 5, False, 4, 1e-12)
    assert rel_rows(w, g["w_warm_n5"]) < EPS


def test_deconv_end_to_end_case1(pa, golden):
    pybold_amd, _ = pa
    g = golden("case1")
    np.random.seed(0)
    x, z, dz, J, R, G = pybold_amd.deconv(g["y"], 1.0, g["hrf"], lbda=1.0, nb_iter=500,
                                          early_stopping=False)
    assert R is None and G is None
    assert x.dtype == np.float64 and x.shape == g["x"].shape
    assert rel_rows(dz, g["diff_z"]) < EPS
    assert rel_rows(z, g["z"]) < EPS
    assert rel_rows(x, g["x"]) < EPS
    assert len(J) == 500
    np.testing.assert_allclose(J, g["J"], rtol=2e-5)


def test_deconv_early_stopping_goldens(pa, golden):
    pybold_amd, _ = pa
    g = golden("early_stop")
    for tol, n_ref in ((0.1, 26), (0.03, 77), (0.01, 191), (0.005, 311)):
        np.random.seed(0)
        x, z, dz, J, _, _ = pybold_amd.deconv(g["y"], 1.0, g["hrf"], lbda=1.0, nb_iter=1000,
                                              early_stopping=True, tol=tol, wind=6)
        assert len(J) == n_ref, (tol, len(J))
        assert rel_rows(dz, g["dz_%g" % tol]) < EPS
        assert rel_rows(x, g["x_%g" % tol]) < EPS


@pytest.mark.parametrize("force", ["fast1", "wide", "generic", None])
def test_window_rule_in_the_batch_kernels(pa, golden, force):
    """The golden stop iterations (26 / 77 / 191 / 311) through the float32-FIR batch kernels
    (1-D API calls run the float64 kernel): single-row and one-problem-per-wave forms with
    the increment ring, the LDS kernel, and the library's own choice; a row with another
    scale stops on its own."""
    _, solver = pa
    g = golden("early_stop")
    y, hrf, lip = g["y"], g["hrf"], float(g["lipschitz"])
    Yb = np.stack([y, y, 3.0 * y, y, y[::-1].copy()])
    for tol, n_ref in ((0.1, 26), (0.03, 77), (0.01, 191), (0.005, 311)):
        W, J, nd = solver.fista_solve(dev32(Yb), hrf, 1.0, 1.0 / lip, 1000, want_J=True, stop="window",
                                      tol=tol, wind=6, force=force)
        nd = nd.cpu().numpy()
        assert (nd[[0, 1, 3]] == n_ref).all(), (force, tol, nd)
        assert rel_rows(W.cpu().numpy()[[0, 1, 3]], np.stack([g["dz_%g" % tol]] * 3)) < EPS
        Jn = J.cpu().numpy()
        assert np.isfinite(Jn[0, :n_ref]).all() and np.isnan(Jn[0, n_ref:]).all()
        ref = orc.deconv_fixed_lbda(Yb[2].astype(np.float32).astype(np.float64), hrf, 1.0, nb_iter=1000,
                                    tol=tol, lipschitz=lip, dense=False)
        assert nd[2] == ref[4], (force, tol, nd[2], ref[4])


def test_deconv_batch_matches_single(pa, golden):
    pybold_amd, _ = pa
    g, Y, hrf, lip = grid_inputs(golden)
    np.random.seed(0)
    X, Z, W, J, _, _ = pybold_amd.deconv(Y, 1.0, hrf, lbda=1.0, nb_iter=10, early_stopping=False)
    assert W.shape == Y.shape and J.shape == (4, 10)
    for s in range(4):
        assert rel_rows(W[s], g["dz_s%d_l1_n10" % s]) < EPS
        np.testing.assert_allclose(J[s], g["J_s%d_l1_n10" % s], rtol=2e-5)


def test_deconv_auto_lambda_matches_oracle(pa, golden):
    """lbda=None branch (pybold/bold_signal.py:99-214): GPU vs the oracle's
    restatement, same noise estimate; 1-D (lists, as the reference) and batched."""
    pybold_amd, _ = pa
    g, Y, hrf, lip = grid_inputs(golden)
    kw = dict(early_stopping=True, tol=1e-3, wind=6, nb_iter=14, nb_sub_iter=40)
    refs = []
    for s in range(4):
        y32 = Y[s].astype(np.float32).astype(np.float64)
        sigma = orc.mad_daub_noise_est(Y[s])
        refs.append(orc.deconv_auto_lbda(y32, hrf, sigma, lip, **kw))
    np.random.seed(0)
    x, z, dz, J, R, G = pybold_amd.deconv(Y[1], 1.0, hrf, lbda=None, **kw)
    xr, zr, wr, Jr, Rr, Gr = refs[1]
    assert isinstance(J, list) and len(J) == len(Jr)
    assert rel_rows(dz, wr) < EPS and rel_rows(x, xr) < EPS and rel_rows(z, zr) < EPS
    np.testing.assert_allclose(J, Jr, rtol=1e-5)
    np.testing.assert_allclose(R, Rr, rtol=1e-5)
    np.testing.assert_allclose(G, Gr, rtol=1e-5)
    np.random.seed(0)
    X, Z, W, Jb, Rb, Gb = pybold_amd.deconv(Y, 1.0, hrf, lbda=None, **kw)
    for s in range(4):
        assert rel_rows(W[s], refs[s][2]) < EPS, s
        n_out = len(refs[s][3])
        np.testing.assert_allclose(Jb[:n_out, s], refs[s][3], rtol=1e-5)
        assert np.isnan(Jb[n_out:, s]).all()


def test_fista_stats(pa, golden):
    _, solver = pa
    g = golden("case1")
    W = dev64(g["diff_z"][None])
    r2, l1 = solver.fista_stats(W, dev32(g["y"][None]), g["hrf"])
    y32 = g["y"].astype(np.float32).astype(np.float64)
    assert float(r2[0]) == pytest.approx(np.sum(np.square(g["x"] - y32)), rel=1e-10)
    assert float(l1[0]) == pytest.approx(np.abs(g["diff_z"]).sum(), rel=1e-12)


def test_device_tensors_in_device_tensors_out(pa, golden):
    pybold_amd, _ = pa
    g, Y, hrf, lip = grid_inputs(golden)
    np.random.seed(0)
    X, Z, W, J, _, _ = pybold_amd.deconv(dev32(Y), 1.0, hrf, lbda=1.0, nb_iter=10,
                                         early_stopping=False)
    assert all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float64 for t in (X, Z, W, J))
    for s in range(4):
        assert rel_rows(W[s].cpu().numpy(), g["dz_s%d_l1_n10" % s]) < EPS
        np.testing.assert_allclose(J[s].cpu().numpy(), g["J_s%d_l1_n10" % s], rtol=2e-5)
    np.random.seed(0)
    x, z, w, J, _, _ = pybold_amd.deconv(dev32(Y[2]), 1.0, hrf, lbda=1.0, nb_iter=10,
                                         early_stopping=False)
    assert w.shape == (300,) and w.is_cuda and len(J) == 10


def test_operator_surface_goldens(pa, golden):
    pybold_amd, _ = pa
    g = golden("operators")
    for tag in "abcde":
        k, x = g[tag + "_k"], g[tag + "_x"]
        n = len(x)
        H = pybold_amd.ConvAndLinear(pybold_amd.DiscretInteg(), k, dim_in=n, dim_out=n)
        np.testing.assert_allclose(H.op(x), g[tag + "_op"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(H.adj(x), g[tag + "_adj"], rtol=1e-12, atol=1e-12)
        I = pybold_amd.DiscretInteg()
        np.testing.assert_allclose(I.op(x), g[tag + "_integ_op"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(I.adj(x), g[tag + "_integ_adj"], rtol=1e-12, atol=1e-12)
    # rectangular Toeplitz product of pybold/tests/test_convolution.py:169-176
    _, solver = pa
    sig, k = g["rect_sig"], g["rect_k"]
    out = solver.conv(dev64(k[None]), sig, dim_out=len(sig)).cpu().numpy()[0]
    np.testing.assert_allclose(out, g["rect_conv"], rtol=1e-12, atol=1e-12)
    # its transpose against the dense matrix
    r = np.random.RandomState(1).randn(len(sig))
    out = solver.corr(dev64(r[None]), sig, dim_in=len(k)).cpu().numpy()[0]
    np.testing.assert_allclose(out, g["rect_T"].T.dot(r), rtol=1e-12, atol=1e-12)


def test_simple_convolve_definitions(pa, golden):
    """simple_convolve / simple_retro_convolve against the reference's loop-form values
    (the ground truth of pybold/tests/test_convolution.py) incl. the rectangular case."""
    pybold_amd, _ = pa
    g = golden("operators")
    for tag in "abcde":
        k, x = g[tag + "_k"], g[tag + "_x"]
        np.testing.assert_allclose(pybold_amd.simple_convolve(k, x), g[tag + "_conv"],
                                   rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(pybold_amd.simple_retro_convolve(k, x), g[tag + "_retro"],
                                   rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(
        pybold_amd.simple_convolve(g["rect_sig"], g["rect_k"], dim_out=len(g["rect_sig"])),
        g["rect_conv"], rtol=1e-12, atol=1e-12)
    X = np.stack([g["a_x"], -g["a_x"]])
    out = pybold_amd.simple_convolve(g["a_k"], X)
    np.testing.assert_allclose(out[1], -g["a_conv"], rtol=1e-12, atol=1e-12)


def test_operator_identities_like_reference_tests(pa):
    """toeplitz @ x == conv, toeplitz.T @ x == retro-conv, adjointness
    (pybold/tests/test_convolution.py:29-36,95-102; test_linear.py:12-28)."""
    pybold_amd, solver = pa
    rng = np.random.RandomState(5)
    for n, K in ((500, 30), (180, 60), (64, 64), (1000, 7)):
        k, a, b = rng.randn(K), rng.randn(n), rng.randn(n)
        T = orc.toeplitz_from_kernel(k, n, n)
        np.testing.assert_allclose(solver.conv(dev64(a[None]), k).cpu().numpy()[0], T.dot(a),
                                   rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(solver.corr(dev64(a[None]), k).cpu().numpy()[0], T.T.dot(a),
                                   rtol=1e-11, atol=1e-11)
        H = pybold_amd.ConvAndLinear(pybold_amd.DiscretInteg(), k, n, n)
        assert np.dot(H.op(a), b) == pytest.approx(np.dot(a, H.adj(b)), rel=1e-9)
        I = pybold_amd.DiscretInteg()
        np.testing.assert_allclose(I.op(a), np.cumsum(a), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(I.adj(a), np.flipud(np.cumsum(np.flipud(a))), rtol=1e-12,
                                   atol=1e-12)


def test_spectral_radius_matches_golden(pa, golden):
    pybold_amd, _ = pa
    g = golden("case1")
    n = len(g["y"])
    H = pybold_amd.ConvAndLinear(pybold_amd.DiscretInteg(), g["hrf"], n, n)
    np.random.seed(0)
    rho = pybold_amd.spectral_radius_est(H, (n,))
    assert 0.9 * rho == pytest.approx(float(g["lipschitz"]), rel=1e-10)


def test_fused_power_iteration_equals_host_loop(pa, golden):
    pybold_amd, solver = pa
    g = golden("case1")
    n = len(g["y"])
    rho, n_it = solver.spectral_radius(g["x0"], g["hrf"])
    assert 0.9 * rho == pytest.approx(float(g["lipschitz"]), rel=1e-12)
    ref = orc.spectral_radius_est(orc.DenseH(g["hrf"], n, n), g["x0"])
    assert rho == pytest.approx(ref, rel=1e-12) and 1 <= n_it <= 30

    class Wrapped:                       # a foreign .op/.adj object takes the generic host loop
        def __init__(self, H):
            self.H = H

        def op(self, x):
            return self.H.op(x)

        def adj(self, x):
            return self.H.adj(x)
    H = pybold_amd.ConvAndLinear(pybold_amd.DiscretInteg(), g["hrf"], n, n)
    np.random.seed(0)
    a = pybold_amd.spectral_radius_est(H, (n,))
    np.random.seed(0)
    b = pybold_amd.spectral_radius_est(Wrapped(H), (n,))
    assert a == pytest.approx(b, rel=1e-12)
    # not converged within nb_iter: still returns the last norm, like the reference
    np.random.seed(0)
    c = pybold_amd.spectral_radius_est(H, (n,), nb_iter=2)
    np.random.seed(0)
    d = pybold_amd.spectral_radius_est(Wrapped(H), (n,), nb_iter=2)
    assert c == pytest.approx(d, rel=1e-12)


def test_hrf_fit_err_and_estim(pa, golden):
    """1-D calls compute in float64 end to end (pb_hrf_cost_d): the tolerances are the ones
    the CPU oracle itself is held to (tests/test_oracle_golden.py)."""
    pybold_amd, _ = pa
    g = golden("hrf_estim")
    t_r, dur = float(g["t_r"]), float(g["hrf_dur"])
    for theta, err in zip(g["thetas"], g["errs"]):
        got = pybold_amd.hrf_fit_err(theta, g["z"], g["y"], t_r, dur)
        assert got == pytest.approx(err, rel=1e-10)
    h, J = pybold_amd.hrf_estim(g["z"], g["y"], t_r, dur)
    print("hrf_estim: rel err of h vs golden %.3e" % rel_rows(h, g["h"]))
    assert rel_rows(h, g["h"]) < 1e-6
    # the objective on the reference's 131-point theta grid (round-2 fixture)
    g2 = golden("round2")
    for theta, err in list(zip(g2["fit_theta"], g2["fit_err"]))[::10]:
        got = pybold_amd.hrf_fit_err(theta, g2["fit_z"], g2["fit_y"], 0.75, 20.0)
        assert got == pytest.approx(err, rel=1e-10)


def test_bd_five_outer_iterations(pa, golden):
    """Golden bd run (SciPy L-BFGS-B in the loop, as the reference): cost trace, HRF, outputs
    AND the dilation after every outer iteration, at the oracle's own tolerances."""
    pybold_amd, _ = pa
    g, g2 = golden("bd"), golden("round2")
    x, z, dz, h, d = pybold_amd.bd(g["y"], float(g["t_r"]), lbda=float(g["lbda"]),
                                   hrf_dur=float(g["hrf_dur"]), nb_iter=int(g["nb_iter"]))
    errs = dict(J=np.abs(d["J"] / g["J"] - 1).max(), r=np.abs(d["r"] / g["r"] - 1).max(),
                theta=np.abs(d["theta"] - g2["bd_theta"]).max(), h=rel_rows(h, g["h"]),
                x=rel_rows(x, g["x"]), z=rel_rows(z, g["z"]), diff_z=rel_rows(dz, g["diff_z"]))
    print("bd vs golden:", {k: "%.2e" % v for k, v in errs.items()})
    np.testing.assert_allclose(d["J"], g["J"], rtol=1e-6)
    np.testing.assert_allclose(d["r"], g["r"], rtol=1e-6)
    np.testing.assert_allclose(d["g"], g["g"], rtol=1e-5)
    np.testing.assert_allclose(d["theta"], g2["bd_theta"], rtol=0, atol=2e-6)
    assert errs["h"] < 1e-5 and errs["x"] < 1e-5 and errs["diff_z"] < 1e-4


def test_bd_warm_start_from_block_signal(pa, golden):
    """z_0 / theta_0 warm start (pybold/bold_signal.py:291-301), interior dilations."""
    pybold_amd, _ = pa
    g = golden("round2")
    x, z, dz, h, d = pybold_amd.bd(g["bdw_y"], 0.75, lbda=1.7, theta_0=1.0, z_0=g["bdw_z0"],
                                   hrf_dur=20.0, nb_iter=3)
    errs = dict(theta=np.abs(d["theta"] - g["bdw_theta"]).max(), h=rel_rows(h, g["bdw_h"]),
                x=rel_rows(x, g["bdw_x"]), diff_z=rel_rows(dz, g["bdw_diff_z"]))
    print("bd (warm start) vs golden:", {k: "%.2e" % v for k, v in errs.items()})
    np.testing.assert_allclose(d["theta"], g["bdw_theta"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(d["J"], g["bdw_J"], rtol=1e-6)
    np.testing.assert_allclose(d["r"], g["bdw_r"], rtol=1e-6)
    np.testing.assert_allclose(d["g"], g["bdw_g"], rtol=1e-5)
    assert errs["h"] < 1e-5 and errs["x"] < 1e-5 and errs["diff_z"] < 1e-4


def test_default_settings_deconv_golden(pa, golden):
    """The reference's DEFAULT deconv settings (early_stopping=True, tol=1e-6, wind=6,
    nb_iter=1000) on the golden input run all 1000 iterations; same here, in 1-D (float64
    kernel) and through the batch kernels with the window rule armed."""
    pybold_amd, solver = pa
    g = golden("early_stop")
    assert int(g["n_default"]) == 1000
    np.random.seed(0)
    x, z, dz, J, _, _ = pybold_amd.deconv(g["y"], 1.0, g["hrf"], lbda=1.0)
    assert len(J) == 1000 and rel_rows(dz, g["dz_default"]) < 1e-9
    Yb = np.stack([g["y"], 0.5 * g["y"], g["y"][::-1].copy(), g["y"]])
    np.random.seed(0)
    X, Z, W, Jb, _, _ = pybold_amd.deconv(Yb, 1.0, g["hrf"], lbda=1.0)
    assert Jb.shape == (4, 1000)
    assert rel_rows(W[0], g["dz_default"]) < EPS and rel_rows(W[3], g["dz_default"]) < EPS
    assert "register-resident" in solver.which_kernel(300, 30, 4, want_J=True, stop="window", wind=6)


def test_plan_is_graph_capture_safe(pa, golden):
    """pb_fista_solve allocates nothing and never synchronises: a FistaPlan can be
    captured in a HIP graph and replayed (launch-bound inner loops)."""
    _, solver = pa
    g, Y, hrf, lip = grid_inputs(golden)
    plan = solver.FistaPlan(dev32(Y), hrf, 1.0, 1.0 / lip, 10)
    plan.run()
    torch.cuda.synchronize()
    eager = plan.W.clone()
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        plan.run()                       # warm the capture stream
        stream.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            plan.run()
    plan.W.fill_(7.0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(plan.W, eager)
    ref = np.stack([g["dz_s%d_l1_n10" % s] for s in range(4)])
    assert rel_rows(plan.W.cpu().numpy(), ref) < EPS


def test_empty_batch_is_a_no_op(pa, golden):
    _, solver = pa
    g, Y, hrf, lip = grid_inputs(golden)
    W, J, n_done = solver.fista_solve(dev32(Y[:0]), hrf, 1.0, 1.0 / lip, 5, want_J=True)
    assert W.shape == (0, 300) and J.shape == (0, 5) and n_done.shape == (0,)
    X, Z = solver.fista_outputs(W, hrf)
    assert X.shape == (0, 300)


def test_inputs_are_not_modified(pa, golden):
    pybold_amd, solver = pa
    g, Y, hrf, lip = grid_inputs(golden)
    y0 = Y[0].copy()
    np.random.seed(0)
    pybold_amd.deconv(Y[0], 1.0, hrf, lbda=1.0, nb_iter=3, early_stopping=False)
    np.testing.assert_array_equal(Y[0], y0)                      # y untouched (reference contract)
    W0 = dev64(np.random.RandomState(0).randn(4, 300) * 0.01)
    keep = W0.clone()
    solver.fista_solve(dev32(Y), hrf, 1.0, 1.0 / lip, 3, W0=W0)
    assert torch.equal(W0, keep)                                 # warm start is copied
    # non-contiguous / float64 inputs are accepted through the reference-style API
    Yt = np.asfortranarray(Y)
    np.random.seed(0)
    _, _, W, _, _, _ = pybold_amd.deconv(Yt, 1.0, hrf, lbda=1.0, nb_iter=10, early_stopping=False)
    for s in range(4):
        assert rel_rows(W[s], g["dz_s%d_l1_n10" % s]) < EPS


@pytest.mark.parametrize("force", ["fast1", "fast2", "fast2d", "generic"])
def test_bad_voxels_do_not_contaminate_neighbours(pa, golden, force):
    """Voxels are independent problems: NaN / Inf / huge values in one of them must
    leave every other voxel of the wave, row pair and workgroup bit-identical."""
    _, solver = pa
    g, Y, hrf, lip = grid_inputs(golden)
    Yb = np.tile(Y, (5, 1)).astype(np.float32)            # 20 voxels
    clean, _, _ = solver.fista_solve(dev32(Yb), hrf, 1.0, 1.0 / lip, 30, force=force)
    bad = Yb.copy()
    bad[3, 17] = np.nan
    bad[8, :] = np.inf
    bad[13, :] *= 1e30
    W, _, _ = solver.fista_solve(dev32(bad), hrf, 1.0, 1.0 / lip, 30, force=force)
    keep = [i for i in range(20) if i not in (3, 8, 13)]
    assert torch.equal(W[keep], clean[keep])
    assert bool(torch.isnan(W[3]).any()) and not bool(torch.isfinite(W[8]).all())
    # scale equivariance at extreme amplitudes (float32 range of y, float64 iterate)
    for scale in (1e-20, 1e12):
        Ws, _, _ = solver.fista_solve(dev32(Yb * np.float32(scale)), hrf, scale, 1.0 / lip, 30, force=force)
        rel = ((Ws / scale - clean).norm(dim=1) / clean.norm(dim=1)).max()
        assert float(rel) < 1e-5, (scale, float(rel))


def test_errors_are_loud(pa, golden):
    _, solver = pa
    from pybold_amd._lib import PyboldHipError
    g, Y, hrf, lip = grid_inputs(golden)
    with pytest.raises(PyboldHipError):
        solver.fista_solve(dev32(Y), hrf, 1.0, -1.0, 5)              # bad step
    with pytest.raises(PyboldHipError):
        solver.fista_solve(dev32(Y), np.ones(5000), 1.0, 1.0, 5, force="fast")
    with pytest.raises(TypeError):
        solver.fista_solve(torch.zeros(4, 300), hrf, 1.0, 1.0, 5)   # CPU tensor


@pytest.mark.parametrize("force", ["fast1", "wide"])
def test_window_rule_resolution(pa, golden, force):
    """How finely the in-kernel window rule (float32 increments, packed float32 norms)
    resolves the criterion: with `tol` set 1e-5 (relative) above / below the oracle's
    criterion value at some iteration k0, the float32-FIR kernels stop exactly where the
    float64 oracle does, on either side (measured: still exact at 1e-6)."""
    _, solver = pa
    g = golden("early_stop")
    y, hrf, lip = g["y"].astype(np.float32).astype(np.float64), g["hrf"], float(g["lipschitz"])
    # criterion trace of the reference recurrence (oracle restatement, float64)
    H = orc._MatrixFreeH(hrf)
    Hty, step, th = H.adj(y), 1.0 / lip, 1.0 / lip
    w, hist, t_old, crit = np.zeros_like(y), [], 1.0, {}
    for k in range(400):
        u = w - step * (H.adj(H.op(w)) - Hty)
        if k > 0 and hist:
            hist[-1] = u
        p = orc.soft_threshold(u, th)
        t = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t_old ** 2))
        w = p + (t_old - 1.0) / t * (p - (u if k > 0 else 0.0))
        t_old = t
        hist.append(w)
        hist = hist[-6:]
        if k > 6:
            old, new = np.mean(hist[:-3], axis=0), np.mean(hist[-3:], axis=0)
            crit[k] = np.linalg.norm(new - old) / (np.linalg.norm(new) + 1e-10)
    for k0 in (40, 150, 300):
        assert all(crit[k] > crit[k0] * 1.002 for k in range(7, k0)), "criterion not monotone here"
        k_next = min(k for k in range(k0 + 1, 400) if crit[k] < crit[k0] * (1 - 1e-5))
        for tol, n_ref in ((crit[k0] * (1 + 1e-5), k0 + 1), (crit[k0] * (1 - 1e-5), k_next + 1)):
            W, _, nd = solver.fista_solve(dev32(np.stack([y, y])), hrf, 1.0, step, 1000, stop="window",
                                          tol=tol, wind=6, force=force)
            assert nd.cpu().numpy().tolist() == [n_ref, n_ref], (force, k0, tol, nd, n_ref)
