"""GPU tests of the per-voxel-HRF pieces: device HRF model, Lipschitz kernel,
per-voxel cost, per-problem-taps solver, and the batched ``bd``."""
import numpy as np
import pytest
import torch

from oracle import pybold_oracle as orc

pytestmark = pytest.mark.gpu


def d64(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def d32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def test_device_hrf_model_matches_reference_values(golden):
    from pybold_amd import solver
    g = golden("spm_hrf")
    for i in (2, 3, 5):                       # the un-normalised golden cases
        delta, t_r, dur, norm = g["p%d" % i]
        assert not norm
        h = solver.spm_hrf_batch(d64([delta, delta]), t_r, dur).cpu().numpy()
        np.testing.assert_allclose(h[0], g["h%d" % i], rtol=1e-12, atol=1e-15)
        np.testing.assert_array_equal(h[0], h[1])
    thetas = np.linspace(0.5, 2.0, 31)
    H = solver.spm_hrf_batch(d64(thetas), 0.75, 20.0).cpu().numpy()
    for th, h in zip(thetas, H):
        np.testing.assert_allclose(h, orc.spm_hrf(th, 0.75, 20.0, False)[0], rtol=1e-12, atol=1e-15)


def test_gram_frobenius_kernel(golden):
    from pybold_amd import solver
    thetas = np.array([0.6, 0.9, 1.3, 2.0])
    H = np.stack([orc.spm_hrf(t, 0.75, 20.0, False)[0] for t in thetas])
    for n in (240, 300, 17):
        L = solver.gram_frobenius_batch(d64(H), n).cpu().numpy()
        ref = np.array([orc.gram_lipschitz(h, n) for h in H])
        np.testing.assert_allclose(L, ref, rtol=1e-12)
    # the FIR closed form on the corners of its case analysis (K = 1, 2; K = N, K = N - 1; long
    # series; more taps than scans: the HRF is truncated to the series)
    rng = np.random.RandomState(5)
    for K, n in ((1, 5), (2, 2), (2, 9), (3, 50), (30, 30), (30, 31), (30, 32), (48, 2432), (127, 128), (40, 12)):
        Hr = rng.randn(6, K)
        L = solver.gram_frobenius_batch(d64(Hr), n).cpu().numpy()
        ref = np.array([orc.gram_lipschitz(h, n) for h in Hr])
        np.testing.assert_allclose(L, ref, rtol=1e-11, err_msg=str((K, n)))


def test_per_voxel_cost_and_outputs():
    from pybold_amd import solver
    rng = np.random.RandomState(0)
    V, N, K, C = 7, 120, 27, 3
    Z, Y = rng.randn(V, N), rng.randn(V, N)
    T = rng.randn(C, V, K) * 0.2
    cost = solver.hrf_cost_pv(d64(Z), d32(Y), d64(T)).cpu().numpy()
    Y32 = Y.astype(np.float32).astype(np.float64)
    for c in range(C):
        for v in range(V):
            ref = 0.5 * np.sum(np.square(Y32[v] - orc.causal_conv(T[c, v], Z[v])))
            assert cost[c, v] == pytest.approx(ref, rel=1e-12)
    X, Zc = solver.fista_outputs_pp(d64(Z), d64(T[0]))
    for v in range(V):
        np.testing.assert_allclose(Zc.cpu().numpy()[v], np.cumsum(Z[v]), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(X.cpu().numpy()[v], orc.causal_conv(T[0, v], np.cumsum(Z[v])),
                                   rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("force", ["fast", "generic"])
def test_solver_with_per_problem_taps(golden, force):
    from pybold_amd import solver
    g = golden("loops_deconv")
    y = g["y"]
    n = len(y)
    thetas = np.array([2.0, 0.7, 1.1, 1.6, 0.9])
    H = np.stack([orc.spm_hrf(t, 0.75, 20.0, False)[0] for t in thetas])
    Y = np.stack([y * s for s in (1.0, 0.5, -1.0, 2.0, 1.5)])
    steps = 1.0 / np.array([orc.gram_lipschitz(h, n) for h in H])
    W, n_done = solver.fista_solve_pp(d32(Y), d64(H), d64(steps), 1.7, 100, force=force)
    W = W.cpu().numpy()
    for v in range(len(thetas)):
        ref = orc.fista_batch(Y[v:v + 1].astype(np.float32).astype(np.float64), H[v], 1.7, steps[v], 100)[0]
        assert np.linalg.norm(W[v] - ref) / np.linalg.norm(ref) < 1e-5, (force, v)
    # voxel 0 is the golden _loops_deconv case (theta = 2.0, unscaled y)
    ref = g["w_n100_es0_tol1e-12"]
    assert np.linalg.norm(W[0] - ref) / np.linalg.norm(ref) < 1e-5
    # stop rule with per-problem taps
    W2, n_done = solver.fista_solve_pp(d32(Y[:1]), d64(H[:1]), d64(steps[:1]), 1.7, 100, stop="loops",
                                       tol=1e-2, force=force)
    ref = g["w_n100_es1_tol0.01"]
    assert np.linalg.norm(W2.cpu().numpy()[0] - ref) / np.linalg.norm(ref) < 1e-5


def test_per_problem_taps_long_series():
    """pb_fista_solve_pp on a 1 000-scan series: one-problem-per-wave form vs LDS kernel vs oracle."""
    from pybold_amd import solver
    rng = np.random.RandomState(4)
    V, n = 5, 1000
    thetas = np.array([0.6, 0.9, 1.2, 1.5, 1.9])
    H = np.stack([orc.spm_hrf(t, 0.72, 20.0, False)[0] for t in thetas])
    Y = rng.randn(V, n)
    steps = 1.0 / np.array([orc.gram_lipschitz(h, n) for h in H])
    Wf, _ = solver.fista_solve_pp(d32(Y), d64(H), d64(steps), 0.5, 40, force="fast")
    Wg, _ = solver.fista_solve_pp(d32(Y), d64(H), d64(steps), 0.5, 40, force="generic")
    for v in range(V):
        ref = orc.fista_batch(Y[v:v + 1].astype(np.float32).astype(np.float64), H[v], 0.5, steps[v], 40)[0]
        assert np.linalg.norm(Wf.cpu().numpy()[v] - ref) / np.linalg.norm(ref) < 1e-5
        assert np.linalg.norm(Wg.cpu().numpy()[v] - ref) / np.linalg.norm(ref) < 1e-9


def test_section_search_finds_the_scipy_minimiser(golden):
    from pybold_amd import blind
    g = golden("hrf_estim")
    t_r, dur = float(g["t_r"]), float(g["hrf_dur"])
    Z = np.stack([g["z"], g["z"]])
    Y = np.stack([g["y"], 0.5 * g["y"]])
    theta, cost, taps = blind.fit_dilations(d64(Z), d64(Y), t_r, dur, [(0.6, 1.9)])
    h = orc.spm_hrf(float(theta[0]), t_r, dur, False)[0]
    np.testing.assert_allclose(taps[0].cpu().numpy(), h, rtol=1e-10, atol=1e-14)
    print("fit_dilations vs hrf_estim: rel err of h %.2e" % (np.linalg.norm(h - g["h"]) / np.linalg.norm(g["h"])))
    assert np.linalg.norm(h - g["h"]) / np.linalg.norm(g["h"]) < 1e-4     # hrf_estim's answer
    assert float(cost[0]) <= min(g["J"]) * (1 + 1e-9)                     # at least as good
    # the pass-per-candidate search of round 1 lands on the same dilations
    th_grid, c_grid = blind.fit_dilations_grid(d64(Z), d64(Y), t_r, dur, [(0.6, 1.9)])
    np.testing.assert_allclose(theta.cpu().numpy(), th_grid.cpu().numpy(), atol=2e-7)
    th32, _, _ = blind.fit_dilations(d64(Z), d32(Y), t_r, dur, [(0.6, 1.9)])   # float32 y of a batch
    np.testing.assert_allclose(th32.cpu().numpy(), theta.cpu().numpy(), atol=1e-6)


def test_bd_batch_tracks_single_voxel_bd(golden):
    import pybold_amd
    g = golden("bd")
    y, t_r, dur, lbda = g["y"], float(g["t_r"]), float(g["hrf_dur"]), float(g["lbda"])
    Y = np.stack([y, 0.8 * y, y[::-1].copy()])
    x, z, w, h, d = pybold_amd.bd(Y, t_r, lbda=lbda, hrf_dur=dur, nb_iter=5)
    assert x.shape == Y.shape and h.shape == (3, len(g["h"])) and d["J"].shape == (7, 3)
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    errs = dict(J=np.abs(d["J"][:, 0] / g["J"] - 1).max(), h=rel(h[0], g["h"]), x=rel(x[0], g["x"]),
                diff_z=rel(w[0], g["diff_z"]))
    print("bd batch (float32 y, float32 FIRs, exact theta minimiser) vs golden:",
          {k: "%.2e" % v for k, v in errs.items()})
    # golden = voxel 0.  The batch path differs from the reference by float32 storage of y,
    # float32 FIRs in the z-step and an exact theta minimiser where L-BFGS-B stops ~1e-5 early
    assert errs["J"] < 1e-6 and errs["h"] < 1e-5 and errs["x"] < 1e-5 and errs["diff_z"] < 1e-5
    # and against the SciPy-in-the-loop single-voxel path for another voxel
    xs, zs, ws, hs, ds = pybold_amd.bd(Y[2], t_r, lbda=lbda, hrf_dur=dur, nb_iter=5)
    np.testing.assert_allclose(d["J"][:, 2], ds["J"], rtol=1e-5)
    assert rel(h[2], hs) < 1e-5
