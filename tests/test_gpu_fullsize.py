"""Full-size parity (BASELINE config 3: 100k voxels x 300 scans) through
size-independent properties of the recurrence, plus an oracle check on a random
sample.  The oracle cannot run 5e7 voxel-iterations in seconds; these
properties hold for the exact recurrence and pin the batched kernel at scale:

  * batch independence: a voxel's result does not depend on which batch it is
    solved in (bitwise);
  * odd symmetry: solve(-y) == -solve(y) (bitwise: fma, clamp and the scans are odd);
  * positive homogeneity: solve(a*y, a*lbda) == a*solve(y, lbda);
  * lambda = 0 linearity: solve(y1 + y2) == solve(y1) + solve(y2) without the prox.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu

V, N, LIP = 100000, 300, 723876.2744579345


@pytest.fixture(scope="module")
def setup():
    from pybold_amd import data, solver
    from pybold_amd.hrf_model import spm_hrf
    hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=5, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0,
                                           std_dur=1.0, snr=1.0, seed=11)
    return solver, hrf, Y


def rel_rows_t(a, b):
    return float(((a - b).norm(dim=1) / (b.norm(dim=1) + 1e-300)).max())


def test_full_batch_properties(setup):
    solver, hrf, Y = setup
    step = 1.0 / LIP
    for force in ("fast1", "fast2", "fast2d", "mfma"):  # the register-resident kernel forms
        W, _, n_done = solver.fista_solve(Y, hrf, 1.0, step, 60, force=force)
        assert W.shape == (V, N) and bool(torch.isfinite(W).all()) and int(n_done.min()) == 60
        # batch independence (bitwise): odd-sized slice, different workgroup/row packing
        lo, hi = 31337, 31337 + 4099
        Ws, _, _ = solver.fista_solve(Y[lo:hi].contiguous(), hrf, 1.0, step, 60, force=force)
        assert torch.equal(Ws, W[lo:hi]), force
        # odd symmetry (bitwise; the matrix unit's float32 accumulation is not sign-symmetric in its last bit)
        Wn, _, _ = solver.fista_solve(-Y, hrf, 1.0, step, 60, force=force)
        assert (torch.equal(Wn, -W) if force != "mfma" else rel_rows_t(Wn, -W) < 1e-6), force
        # positive homogeneity
        Wh, _, _ = solver.fista_solve(Y * 4.0, hrf, 4.0, step, 60, force=force)  # power of two: exact
        assert torch.equal(Wh, 4.0 * W), force
        Wh, _, _ = solver.fista_solve(Y * 3.0, hrf, 3.0, step, 60, force=force)
        assert rel_rows_t(Wh, 3.0 * W) < 1e-5
    # the kernel forms agree (different summation order inside the FIR; float16 split products)
    W1, _, _ = solver.fista_solve(Y, hrf, 1.0, step, 60, force="fast1")
    assert rel_rows_t(W, W1) < 2e-6


def test_full_batch_linearity_without_prox(setup):
    solver, hrf, Y = setup
    step = 1.0 / LIP
    Y1, Y2 = Y[: V // 2], Y[V // 2:]
    W1, _, _ = solver.fista_solve(Y1, hrf, 0.0, step, 40)
    W2, _, _ = solver.fista_solve(Y2, hrf, 0.0, step, 40)
    W12, _, _ = solver.fista_solve(Y1 + Y2, hrf, 0.0, step, 40)
    assert rel_rows_t(W12, W1 + W2) < 1e-5


def test_full_batch_sample_against_oracle(setup):
    """500 iterations on all 100k voxels; 96 random voxels checked against the C oracle."""
    solver, hrf, Y = setup
    step = 1.0 / LIP
    W, _, _ = solver.fista_solve(Y, hrf, 1.0, step, 500)
    idx = np.random.RandomState(0).choice(V, 96, replace=False)
    Ys = Y[torch.from_numpy(idx).cuda()].cpu().numpy().astype(np.float64)
    Wo, _, _ = c_oracle.fista_batch(Ys, hrf, 1.0, step, 500, threads=8)
    Wg = W[torch.from_numpy(idx).cuda()].cpu().numpy()
    err = (np.linalg.norm(Wg - Wo, axis=1) / np.linalg.norm(Wo, axis=1)).max()
    assert err < 1e-5, err
    # z and x too
    X, Z = solver.fista_outputs(W[torch.from_numpy(idx).cuda()].contiguous(), hrf)
    Zo = np.cumsum(Wo, axis=1)
    assert (np.linalg.norm(Z.cpu().numpy() - Zo, axis=1) / np.linalg.norm(Zo, axis=1)).max() < 1e-5


def test_regularisation_path_config5_slice(setup):
    """(voxel, lambda) problems sharing y rows: each equals its own single-lambda solve."""
    solver, hrf, Y = setup
    step = 1.0 / LIP
    Ysub = Y[:2000].contiguous()
    lbdas = np.logspace(-2, 0, 20)
    lam = np.tile(lbdas, Ysub.shape[0])
    Wp, _, _ = solver.fista_solve(Ysub, hrf, lam, step, 50, y_rep=20)
    Wp = Wp.reshape(Ysub.shape[0], 20, N)
    for i in (0, 7, 19):
        Wi, _, _ = solver.fista_solve(Ysub, hrf, float(lbdas[i]), step, 50)
        assert rel_rows_t(Wp[:, i, :], Wi) < 1e-6       # may run on different kernels
    idx = np.array([3, 777, 1999])                      # and against the C oracle
    Ys = Ysub[torch.from_numpy(idx).cuda()].cpu().numpy().astype(np.float64)
    for i in (0, 19):
        Wo, _, _ = c_oracle.fista_batch(Ys, hrf, float(lbdas[i]), step, 50, threads=4)
        Wg = Wp[torch.from_numpy(idx).cuda(), i, :].cpu().numpy()
        assert (np.linalg.norm(Wg - Wo, axis=1) / np.linalg.norm(Wo, axis=1)).max() < 1e-5


# ---- every BASELINE config at full size (round 2) ----------------------------------------
def sample_vs_oracle(W, Y, idx, hrf, lbda, step, n_iter, W0=None):
    sel = torch.from_numpy(idx).cuda()
    Ys = Y[sel].cpu().numpy().astype(np.float64)
    lb = lbda if np.ndim(lbda) == 0 else np.asarray(lbda)[idx]
    Wo, _, _ = c_oracle.fista_batch(Ys, hrf, lb, step, n_iter, W0=W0, threads=8)
    Wg = W[sel].cpu().numpy()
    return float((np.linalg.norm(Wg - Wo, axis=1) / (np.linalg.norm(Wo, axis=1) + 1e-300)).max())


def test_config2_ten_thousand_voxels(setup):
    """BASELINE config 2: 10k voxels x 300 scans, fixed HRF, L1 deconv, 500 iterations; the
    dispatch decision for this size is what the library reports and is register-resident;
    64 voxels against the C oracle."""
    solver, hrf, Y = setup
    step = 1.0 / LIP
    Y2 = Y[:10000].contiguous()
    name = solver.which_kernel(N, len(hrf), 10000)
    assert "register-resident" in name
    W, _, n_done = solver.fista_solve(Y2, hrf, 1.0, step, 500)
    assert int(n_done.min()) == 500 and bool(torch.isfinite(W).all())
    idx = np.random.RandomState(2).choice(10000, 64, replace=False)
    err = sample_vs_oracle(W, Y2, idx, hrf, 1.0, step, 500)
    print("config 2 (%s): max rel err of 64 voxels vs C oracle %.2e" % (name, err))
    assert err < 1e-5
    # the same voxels inside the 100k batch of config 3 (other kernel form / packing)
    W3, _, _ = solver.fista_solve(Y, hrf, 1.0, step, 500)
    assert rel_rows_t(W3[:10000], W) < 1e-6


def test_config5_regularisation_path_full_size(setup):
    """BASELINE config 5: 50k voxels x 20 lambdas = 10^6 problems sharing y rows,
    lambda = logspace(-2, 0, 20) * lambda_max,v with lambda_max,v = ||H^T y_v||_inf from the
    device helper; 500 iterations; a sample of (voxel, lambda) problems vs the C
    oracle; at lambda_max the path ends in the all-zero solution."""
    solver, hrf, Y = setup
    step = 1.0 / LIP
    V5, L = 50000, 20
    Y5 = Y[:V5].contiguous()
    lmax = solver.lambda_max(Y5, hrf)                                   # (V5,) on the device
    grid = torch.logspace(-2, 0, L, dtype=torch.float64, device="cuda")
    lam = (lmax[:, None] * grid[None, :]).reshape(-1)                   # (V5 * L,)
    W, _, n_done = solver.fista_solve(Y5, hrf, lam, step, 500, y_rep=L)
    assert W.shape == (V5 * L, N) and int(n_done.min()) == 500 and bool(torch.isfinite(W).all())
    assert "register-resident" in solver.which_kernel(N, len(hrf), V5 * L)
    rng = np.random.RandomState(5)
    vox, li = rng.choice(V5, 48, replace=False), rng.randint(0, L, 48)
    Ys = Y5[torch.from_numpy(vox).cuda()].cpu().numpy().astype(np.float64)
    lams = lam.cpu().numpy()[vox * L + li]
    Wo, _, _ = c_oracle.fista_batch(Ys, hrf, lams, step, 500, threads=8)
    Wg = W[torch.from_numpy(vox * L + li).cuda()].cpu().numpy()
    err = (np.linalg.norm(Wg - Wo, axis=1) / (np.linalg.norm(Wo, axis=1) + 1e-300)).max()
    print("config 5: max rel err of 48 (voxel, lambda) problems vs C oracle %.2e" % err)
    assert err < 1e-5
    # top of the path: the first prox step thresholds everything at lambda_max (beta_0 = 0)
    W1, _, _ = solver.fista_solve(Y5[:512].contiguous(), hrf, lmax[:512] * (1 + 1e-5), step, 1)
    assert float(W1.abs().max()) == 0.0
    # sparsity grows along the path (more weight on the L1 term)
    Wv = W.reshape(V5, L, N)[:256]
    l1 = Wv.abs().sum(dim=2)
    assert bool((l1[:, -1] <= l1[:, 0]).all())


def test_config4_shared_hrf_full_size():
    """BASELINE config 4: 50k voxels, N = 300, K = 27 (TR 0.75 s, 20 s HRF), true dilation
    0.7, theta_0 = 2.0, lambda = 1.7, 20 outer x 100 inner iterations, Frobenius step, shared
    theta fitted on the device.  Cost decreases monotonically; the normal equations of two
    25k-voxel shards add up to those of the whole batch (what the all-reduce computes) and
    give the same dilation; a 64-voxel sample of the first z-step matches the oracle."""
    from oracle import pybold_oracle as orc
    from pybold_amd import data, distributed, solver
    from pybold_amd.utils import gram_frobenius
    t_r, dur, V4 = 0.75, 20.0, 50000
    h_true = orc.spm_hrf(0.7, t_r, dur, False)[0]
    Y4, _, _ = data.gen_rnd_bloc_bold_batch(V4, dur=3.75, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0,
                                            std_dur=1.0, snr=10.0, seed=4)
    assert Y4.shape == (V4, 300) and len(h_true) == 27
    W, h, d = distributed.bd_shared(Y4, t_r, lbda=1.7, theta_0=2.0, hrf_dur=dur, nb_iter=20, nb_inner=100)
    print("config 4: theta trajectory", np.round(d["theta"], 5), "J[-1] = %.6f" % d["J"][-1])
    assert len(d["J"]) == 22 and d["J"][0] == 1.0 and (np.diff(d["J"]) < 0).all()
    assert 0.6 <= d["theta"][-1] < d["theta"][1] <= 1.9
    np.testing.assert_allclose(h, orc.spm_hrf(float(d["theta"][-1]), t_r, dur, False)[0], rtol=1e-10, atol=1e-14)
    # shard additivity of the theta-step at full size (2 x 25k == 50k)
    Z = solver.integ_op(W)
    ne = solver.hrf_normal_eq(Z, Y4, 27)
    ne2 = solver.hrf_normal_eq(Z[:25000], Y4[:25000], 27) + solver.hrf_normal_eq(Z[25000:], Y4[25000:], 27)
    np.testing.assert_allclose(ne2.cpu().numpy(), ne.cpu().numpy(), rtol=1e-11)
    th, f, _ = solver.theta_fit(ne, t_r, dur, (0.6, 1.9))
    th2, f2, _ = solver.theta_fit(ne2, t_r, dur, (0.6, 1.9))
    assert abs(float(th[0]) - float(th2[0])) < 1e-9
    assert float(f[0]) == pytest.approx(float(solver.hrf_cost(Z, Y4, orc.spm_hrf(float(th[0]), t_r, dur, False)[0]).sum()),
                                        rel=1e-7)                       # quadratic form == direct cost
    # the whole loop against the float64 oracle: 256 voxels of the batch solved alone, the dilation
    # after each of the first 3 outer iterations (z-steps, normal equations, 1-D search)
    from oracle.shared_ops import OracleOps

    class _One:
        world_size, rank = 1, 0

        @staticmethod
        def allreduce_(t):
            return t
    sub = torch.from_numpy(np.sort(np.random.RandomState(7).choice(V4, 256, replace=False))).cuda()
    Ys = Y4[sub].contiguous()
    Wg, hg, dg = distributed.bd_shared(Ys, t_r, lbda=1.7, theta_0=2.0, hrf_dur=dur, nb_iter=3, nb_inner=100)
    Wo, ho, do = distributed.bd_shared(Ys.cpu(), t_r, lbda=1.7, theta_0=2.0, hrf_dur=dur, nb_iter=3, nb_inner=100,
                                       ops=OracleOps(300, t_r, dur), comm=_One())
    print("config 4, 256 voxels alone: theta GPU", np.round(dg["theta"], 8), "oracle", np.round(do["theta"], 8))
    assert np.abs(np.asarray(dg["theta"]) - np.asarray(do["theta"])).max() < 1e-6
    np.testing.assert_allclose(dg["J"], do["J"], rtol=1e-6)
    assert float(((Wg.cpu() - Wo).norm(dim=1) / Wo.norm(dim=1)).max()) < 1e-5
    # first z-step (theta_0 = 2.0, from zero) of a 64-voxel sample against the oracle
    W0, h0, d0 = distributed.bd_shared(Y4, t_r, lbda=1.7, theta_0=2.0, hrf_dur=dur, nb_iter=0, nb_inner=100)
    h20 = orc.spm_hrf(2.0, t_r, dur, False)[0]
    np.testing.assert_allclose(h0, h20, rtol=1e-10, atol=1e-14)
    idx = np.random.RandomState(4).choice(V4, 1024, replace=False)
    err = sample_vs_oracle(W0, Y4, idx, h20, 1.7, 1.0 / gram_frobenius(h20, 300), 100)
    print("config 4: first z-step, max rel err of 1024 voxels vs C oracle %.2e" % err)
    assert err < 1e-5


def test_four_million_problems_in_one_call(setup):
    """Maximum sizes: 4.2 M problems in one call (200 k voxels x 21 lambdas; the iterate alone
    is 10 GB) -- 64-bit row offsets, grids beyond 2^31 lanes, the ragged end of every piece of
    the launch plan.  Rows solved in the big call equal the same rows solved in a small one
    (bitwise where the small batch runs the same kernel form, 1e-6 otherwise) and a sample
    matches the C oracle; more than 2^25 problems are refused, not truncated."""
    solver, hrf, Y = setup
    reps = 21
    Yb = torch.cat([Y, -Y], dim=0)                                     # 200 000 voxels
    P = Yb.shape[0] * reps
    lam = torch.logspace(-2, 0, reps, dtype=torch.float64, device="cuda").repeat(Yb.shape[0])
    n_iter = 6
    W, _, n_done = solver.fista_solve(Yb, hrf, lam, 1.0 / LIP, n_iter, y_rep=reps)
    assert W.shape == (P, N) and int(n_done.min()) == n_iter and bool(torch.isfinite(W).all())
    n_main, _, _ = solver.launch_plan(N, len(hrf), P)
    rng = np.random.RandomState(4)
    rows = np.unique(np.r_[0, 1, n_main - 1, n_main, P - 2, P - 1, rng.randint(0, P, 40)])
    rows = rows[(rows >= 0) & (rows < P)]
    vox, k = rows // reps, rows % reps
    Ys = Yb[torch.from_numpy(vox).cuda()]
    lam_s = lam[torch.from_numpy(rows).cuda()]
    Ws, _, _ = solver.fista_solve(Ys, hrf, lam_s, 1.0 / LIP, n_iter)
    assert rel_rows_t(W[torch.from_numpy(rows).cuda()], Ws) < 1e-6
    Wo, _, _ = c_oracle.fista_batch(Ys.cpu().numpy().astype(np.float64), hrf, lam_s.cpu().numpy(), 1.0 / LIP,
                                    n_iter, threads=4)
    assert rel_rows_t(W[torch.from_numpy(rows).cuda()].cpu(), torch.from_numpy(Wo)) < 1e-5
    # odd symmetry across the two halves of the batch (same lambda index): bitwise
    half = Y.shape[0] * reps
    # (the default dispatch partitions the call on the device since round 5: a problem and its mirror image sit at
    # different places of their lists and may run on different kernel forms -- equal within the forms' accuracy; on the
    # vector dispatch every row runs the same arithmetic wherever it sits: bitwise)
    a, b = W[:1000 * reps], -W[half:half + 1000 * reps]
    assert float(((a - b).norm(dim=1) / a.norm(dim=1).clamp_min(1e-300)).max()) < 4e-6
    Wv, _, _ = solver.fista_solve(Yb, hrf, lam, 1.0 / LIP, n_iter, y_rep=reps, force="valu")
    assert torch.equal(Wv[:1000 * reps], -Wv[half:half + 1000 * reps])
    del W
    # the guard, through the raw C ABI (no 80 GB iterate needed: it fires before any access)
    from pybold_amd import _lib
    lib = _lib.load()
    taps = np.ascontiguousarray(hrf, dtype=np.float64)
    w1 = torch.zeros((1, N), dtype=torch.float64, device="cuda")
    betas = torch.zeros((1,), dtype=torch.float64, device="cuda")
    rc = lib.pb_fista_solve(Yb.data_ptr(), N, (1 << 25) + 1, w1.data_ptr(), N, (1 << 25) + 1, N,
                            taps.ctypes.data, None, taps.size, 1.0 / LIP, 1.0, None, betas.data_ptr(), 1,
                            None, 0, 0, 0.0, 0, None, 0, None)
    assert rc != 0 and b"2^25" in lib.pb_last_error()


# ---- round 5: the big oracle samples inside the driver-run record -------------------
def _errs_vs_oracle(solver, W, Y, idx, hrf, lbda, step, n_iter, y_rep=1, threads=16):
    """max relative L2 over the sampled problems on diff_z, z and x against the C float64 oracle."""
    sel = torch.from_numpy(idx).cuda()
    Ys = Y[torch.from_numpy(idx // y_rep).cuda()].cpu().numpy().astype(np.float64)
    lb = lbda if np.ndim(lbda) == 0 else np.asarray(lbda)[idx]
    Wo, _, _ = c_oracle.fista_batch(Ys, hrf, lb, step, n_iter, threads=threads)
    Wg = W[sel].contiguous()
    Xg, Zg = solver.fista_outputs(Wg, hrf)
    Zo = np.cumsum(Wo, axis=1)
    Xo = np.stack([np.convolve(z, hrf)[:len(z)] for z in Zo])
    nz = np.linalg.norm(Wo, axis=1) > 0

    def worst(a, b):
        return float((np.linalg.norm(a - b, axis=1)[nz] / np.linalg.norm(b, axis=1)[nz]).max())
    assert bool((Wg.cpu().numpy()[~nz] == 0).all())                 # exact zeros where the oracle has them
    return worst(Wg.cpu().numpy(), Wo), worst(Zg.cpu().numpy(), Zo), worst(Xg.cpu().numpy(), Xo)


def test_config3_four_thousand_voxels_against_the_oracle(setup):
    """BASELINE config 3 as the bench runs it (100 000 voxels, lambda = 1, 500 iterations, default dispatch): 4 352
    random voxels against the C oracle on diff_z, z AND x -- 256 of them from the remainder launch (the voxels behind
    the whole matrix-pipe rounds), 1 024 from the last round of the main launch, 3 072 from anywhere."""
    solver, hrf, Y = setup
    step = 1.0 / LIP
    W, _, nd = solver.fista_solve(Y, hrf, 1.0, step, 500)
    assert int(nd.min()) == 500 and int(nd.max()) == 500
    n_main, main, tail = solver.launch_plan(N, len(hrf), V, force="nopart")
    assert 0 < n_main < V and "matrix pipe" in main
    rng = np.random.RandomState(2025)
    idx = np.unique(np.r_[n_main + rng.choice(V - n_main, 256, replace=False),
                          n_main - 16384 + rng.choice(16384, 1024, replace=False),
                          rng.choice(V, 3072, replace=False)])
    e = _errs_vs_oracle(solver, W, Y, idx, hrf, 1.0, step, 500)
    print("config 3: %d voxels vs the C oracle: diff_z %.2e  z %.2e  x %.2e" % ((len(idx),) + e))
    assert max(e) < 1e-5, e


def test_configs_2_and_5_a_thousand_problems_against_the_oracle(setup):
    """Config 2 (10 000 voxels) and config 5 (50 000 voxels x 20 lambdas = logspace(-2, 0, 20) lambda_max,v, partitioned on
    the device): 1 024 random problems each against the C oracle on diff_z, z, x."""
    solver, hrf, Y = setup
    step = 1.0 / LIP
    Y2 = Y[:10000].contiguous()
    W2, _, _ = solver.fista_solve(Y2, hrf, 1.0, step, 500)
    idx = np.random.RandomState(7).choice(10000, 1024, replace=False)
    e2 = _errs_vs_oracle(solver, W2, Y2, idx, hrf, 1.0, step, 500)
    print("config 2: 1024 voxels vs the C oracle: diff_z %.2e  z %.2e  x %.2e" % e2)
    assert max(e2) < 1e-5, e2
    Y5 = Y[:50000].contiguous()
    lmax = solver.lambda_max(Y5, hrf)
    lam = (lmax[:, None] * torch.logspace(-2.0, 0.0, 20, dtype=torch.float64, device="cuda")[None, :]).reshape(-1)
    W5, _, nd5 = solver.fista_solve(Y5, hrf, lam, step, 500, y_rep=20, lmax=lmax)
    assert int(nd5.min()) == 500
    idx5 = np.random.RandomState(8).choice(50000 * 20, 1024, replace=False)
    e5 = _errs_vs_oracle(solver, W5, Y5, idx5, hrf, lam.cpu().numpy(), step, 500, y_rep=20)
    print("config 5: 1024 problems vs the C oracle: diff_z %.2e  z %.2e  x %.2e" % e5)
    assert max(e5) < 1e-5, e5
