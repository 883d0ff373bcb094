"""GPU tests of the round-3 additions, all through the C ABI: the deconv window rule carried on
the two-problems-per-row kernel as a no-fire certificate with an exact re-solve of the problems
it cannot clear (pybold/bold_signal.py:82-95; include/pybold_hip.h, pb_fista_solve)."""
import numpy as np
import pytest
import torch

from oracle import pybold_oracle as orc

pytestmark = pytest.mark.gpu

EPS = 1.0e-5


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


def synthetic(n_vox, seed, n=300):
    from pybold_amd import data
    from pybold_amd.hrf_model import spm_hrf
    hrf = spm_hrf(1.0, t_r=1.0, dur=30.0)[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(n_vox, dur=n / 60.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0,
                                           std_dur=1.0, snr=1.0, seed=seed, device=torch.device("cuda"))
    return Y, hrf


def oracle_window(Yh, hrf, lbda, lip, n_iter, tol, w0=None):
    out = [orc.deconv_fixed_lbda(Yh[v], hrf, lbda, nb_iter=n_iter, tol=tol, lipschitz=lip, dense=False,
                                 w0=None if w0 is None else w0[v]) for v in range(len(Yh))]
    return np.stack([o[2] for o in out]), np.array([o[4] for o in out]), [o[3] for o in out]


@pytest.mark.parametrize("force", ["cert", "cert2"])
def test_certificate_half_fire_half_never(solver, golden, force):
    """A batch where about half of the voxels meet the window rule before n_iter and the others
    never do, forced through the certificate path: stop iterations, iterates and cost traces
    equal the float64 oracle's; the rows whose rule never fires are bit-identical to the plain
    solve of the same kernel form (the certificate only watches)."""
    g = golden("early_stop")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    Y, _ = synthetic(37, seed=5)                      # odd count: a lone problem in the last row
    Y[3] = 0.0                                        # y = 0: criterion 0/0 -> fires at the first test
    Y[20] = torch.from_numpy(g["y"]).float().cuda()   # golden series: fires at 191 for tol = 0.01
    tol = 0.01
    Yh = Y.cpu().numpy().astype(np.float64)
    _, n_fire, _ = oracle_window(Yh, hrf, 1.0, lip, 400, tol)      # where each voxel's rule fires
    assert n_fire[20] == 191 and n_fire[3] == 8
    n_iter = int(np.median(n_fire))                                # about half fire before n_iter
    Wr, nr, Jr = oracle_window(Yh, hrf, 1.0, lip, n_iter, tol)
    fired = n_fire < n_iter
    assert 8 <= fired.sum() <= 29, fired.sum()                     # a real mix
    assert (nr == np.minimum(n_fire, n_iter)).all()
    W, J, nd = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, n_iter, want_J=True, stop="window", tol=tol,
                                  wind=6, force=force)
    nd = nd.cpu().numpy()
    assert (nd == nr).all(), np.flatnonzero(nd != nr)
    Wn, Jn = W.cpu().numpy(), J.cpu().numpy()
    live = np.linalg.norm(Wr, axis=1) > 0
    assert rel_rows(Wn[live], Wr[live]) < EPS
    assert np.abs(Wn[~live]).max() == 0.0
    for v in range(len(Yh)):
        assert np.isfinite(Jn[v, :nd[v]]).all() and np.isnan(Jn[v, nd[v]:]).all(), v
        if live[v]:
            # the oracle's trace is normalised by J[0] like the reference's (bold_signal.py:97)
            np.testing.assert_allclose(Jn[v, :nd[v]] / Jn[v, 0], Jr[v], rtol=5e-5)
    # the never-firing rows against the plain solve (so close to firing, some of them were flagged
    # and re-solved on the single-row form, whose direct FIRs round differently: not bitwise here;
    # test_default_tolerance_runs_on_the_pair_form holds the bitwise claim)
    Wp, Jp, _ = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, n_iter, want_J=True, force="fast2")
    keep = np.flatnonzero(~fired)
    assert rel_rows(Wn[keep], Wp.cpu().numpy()[keep]) < 1e-6


def test_certificate_warm_start_and_per_problem_lambda(solver, golden):
    """Flagged problems restart from the caller's warm start (the certificate kernel leaves
    their iterate untouched), with y_rep > 1 and one lambda per problem."""
    g = golden("early_stop")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    Y, _ = synthetic(6, seed=11)
    y_rep = 3
    lbda = np.tile(np.array([0.3, 1.0, 3.0]), 6)
    rng = np.random.RandomState(3)
    W0 = 0.01 * rng.randn(18, 300)
    tol = 0.012
    Yh = np.repeat(Y.cpu().numpy().astype(np.float64), y_rep, axis=0)
    n_fire = np.array([orc.deconv_fixed_lbda(Yh[p], hrf, lbda[p], nb_iter=400, tol=tol, lipschitz=lip, dense=False,
                                             w0=W0[p])[4] for p in range(18)])
    n_iter = int(np.median(n_fire))
    out = [orc.deconv_fixed_lbda(Yh[p], hrf, lbda[p], nb_iter=n_iter, tol=tol, lipschitz=lip, dense=False,
                                 w0=W0[p]) for p in range(18)]
    Wr, nr = np.stack([o[2] for o in out]), np.array([o[4] for o in out])
    assert 4 <= (n_fire < n_iter).sum() <= 14
    W, J, nd = solver.fista_solve(Y, hrf, lbda, 1.0 / lip, n_iter, W0=dev64(W0), want_J=True, stop="window",
                                  tol=tol, wind=6, y_rep=y_rep, force="cert")
    assert (nd.cpu().numpy() == nr).all()
    assert rel_rows(W.cpu().numpy(), Wr) < EPS


def test_default_tolerance_runs_on_the_pair_form(solver, golden):
    """The reference defaults (tol = 1e-6, wind = 6, 1000 iterations): the library picks the
    certificate path by itself, nothing fires, and the result is bit-identical to the plain
    solve with cost trace; n_done = n_iter everywhere.  Larger batch: several pieces (whole
    rounds on the pair form, a remainder on other forms)."""
    g = golden("early_stop")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    assert "matrix pipe" in solver.which_kernel(300, 30, 20000, stop="window", wind=6)
    Y, _ = synthetic(20000, seed=2)
    n_iter = 300
    # (force="valu": the vector forms -- whole rounds on the pair form with the certificate; the
    # matrix-pipe form carries the same certificate: tests/test_gpu_mfma.py)
    W, J, nd = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, n_iter, want_J=True, stop="window", tol=1e-6, wind=6, force="valu")
    assert int(nd.min()) == n_iter and int(nd.max()) == n_iter
    Wp, Jp, _ = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, n_iter, want_J=True, force="valu")   # same kernel form
    assert torch.equal(W, Wp) and torch.equal(J, Jp)
    # and against the full rule evaluated on the single-row form
    Wn, Jn, ndn = solver.fista_solve(Y[:4096], hrf, 1.0, 1.0 / lip, n_iter, want_J=True, stop="window",
                                     tol=1e-6, wind=6, force="nocert")
    assert int(ndn.min()) == n_iter
    assert rel_rows(W[:4096].cpu().numpy(), Wn.cpu().numpy()) < 1e-6
    # the golden default run (1000 iterations, the rule never fires): pinned to the reference
    Wg, _, ndg = solver.fista_solve(dev32(np.stack([g["y"]] * 4)), hrf, 1.0, 1.0 / lip, 1000, want_J=True,
                                    stop="window", tol=1e-6, wind=6, force="valu")
    assert (ndg.cpu().numpy() == int(g["n_default"])).all()
    assert rel_rows(Wg.cpu().numpy(), np.stack([g["dz_default"]] * 4)) < EPS


def test_certificate_is_tight_enough_to_be_useful(solver, golden):
    """At tol = 1e-4 with 1000 iterations (criterion ~ 9e-4 at the end) the certificate must
    still clear every problem of a realistic batch: same result as the full rule, and -- the
    point -- no slower path taken silently (n_done = n_iter for all is necessary for that)."""
    g = golden("early_stop")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    Y, _ = synthetic(512, seed=7)
    W, _, nd = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, 1000, want_J=True, stop="window", tol=1e-4, wind=6,
                                  force="cert2")
    Wn, _, ndn = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, 1000, want_J=True, stop="window", tol=1e-4, wind=6,
                                    force="nocert")
    assert (nd == ndn).all() and int(nd.min()) == 1000
    assert rel_rows(W.cpu().numpy(), Wn.cpu().numpy()) < 1e-6
    # diagnostic form without the re-solve: flagged problems keep n_done = -1
    _, _, ndc = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, 1000, want_J=True, stop="window", tol=1e-4, wind=6,
                                   force="certonly")
    assert int((ndc < 0).sum()) == 0
    _, _, ndc = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, 1000, want_J=True, stop="window", tol=2e-3, wind=6,
                                   force="certonly")
    assert int((ndc < 0).sum()) == 512               # the rule does fire here (~0.9/k): everything is flagged


# ---- series of 16 S < N <= 32 S scans: the two halves of ONE series in the slots of a row ----------
def long_problem(n_vox, n, k_taps, seed):
    rng = np.random.RandomState(seed)
    hrf = orc.spm_hrf(1.0, 1.0, float(k_taps), False)[0][:k_taps]
    Z = np.zeros((n_vox, n))
    for v in range(n_vox):
        for _ in range(6):
            o = rng.randint(0, n - 40)
            Z[v, o:o + rng.randint(8, 30)] += rng.choice([-1.0, 1.0])
    Y = orc.causal_conv(hrf, Z) + 0.3 * rng.randn(n_vox, n)
    lip = orc.gram_lipschitz(hrf, n)
    return Y, hrf, lip


@pytest.mark.parametrize("n,k", [(600, 30), (608, 30), (305, 30), (400, 27), (480, 30), (520, 32), (330, 16)])
def test_split_pair_form_matches_oracle(solver, n, k):
    """N in (16 S, 32 S]: plain solve, cost trace and the iterate against the float64 oracle;
    the kernel is the pair form (asked for by force='fast2'; the library picks it from 1 024
    series on)."""
    Y, hrf, lip = long_problem(9, n, k, seed=n)
    assert len(hrf) == k
    n_iter = 300
    Wr = orc.fista_batch(Y.astype(np.float32).astype(np.float64), hrf, 0.7, 1.0 / lip, n_iter)
    W, J, nd = solver.fista_solve(dev32(Y), hrf, 0.7, 1.0 / lip, n_iter, want_J=True, force="fast2")
    assert "matrix pipe" in solver.which_kernel(n, k, 6000)      # (the library's own choice since round 4; the pair form is forced here)
    assert rel_rows(W.cpu().numpy(), Wr) < EPS
    Xr, Zr = orc.fista_outputs(Wr, hrf)
    Jr = 0.5 * np.sum(np.square(Xr - Y.astype(np.float32)), axis=1) + 0.7 * np.abs(Wr).sum(axis=1)
    np.testing.assert_allclose(J.cpu().numpy()[:, -1], Jr, rtol=2e-5)
    # without the cost trace: same iterate (another instantiation: not bitwise), any batch position
    W2, _, _ = solver.fista_solve(dev32(np.concatenate([Y[4:], Y[:4]])), hrf, 0.7, 1.0 / lip, n_iter, force="fast2")
    assert rel_rows(W2.cpu().numpy(), np.concatenate([Wr[4:], Wr[:4]])) < EPS
    W3, _, _ = solver.fista_solve(dev32(Y), hrf, 0.7, 1.0 / lip, n_iter, force="fast2")
    assert torch.equal(W3[4:], W2[:5]) and torch.equal(W3[:4], W2[5:])


def test_split_pair_form_window_rule(solver):
    """600-scan series with the window rule through the certificate path: stop iterations and
    iterates of the oracle, about half of the series firing before n_iter."""
    Y, hrf, lip = long_problem(12, 600, 30, seed=1)
    Y32 = Y.astype(np.float32).astype(np.float64)
    tol = 0.01
    n_fire = np.array([orc.deconv_fixed_lbda(Y32[v], hrf, 0.7, nb_iter=500, tol=tol, lipschitz=lip, dense=False)[4]
                       for v in range(12)])
    n_iter = int(np.median(n_fire))
    out = [orc.deconv_fixed_lbda(Y32[v], hrf, 0.7, nb_iter=n_iter, tol=tol, lipschitz=lip, dense=False)
           for v in range(12)]
    Wr, nr = np.stack([o[2] for o in out]), np.array([o[4] for o in out])
    assert 3 <= (n_fire < n_iter).sum() <= 9
    W, J, nd = solver.fista_solve(dev32(Y), hrf, 0.7, 1.0 / lip, n_iter, want_J=True, stop="window", tol=tol,
                                  wind=6, force="cert2")
    assert (nd.cpu().numpy() == nr).all()
    assert rel_rows(W.cpu().numpy(), Wr) < EPS
    # default tolerance on a full-size batch: nothing fires, same result as the plain solve
    Yb = dev32(np.tile(Y, (200, 1)))
    Wd, _, ndd = solver.fista_solve(Yb, hrf, 0.7, 1.0 / lip, 200, want_J=True, stop="window", tol=1e-6, wind=6)
    Wp, _, _ = solver.fista_solve(Yb, hrf, 0.7, 1.0 / lip, 200, want_J=True)
    assert int(ndd.min()) == 200 and torch.equal(Wd, Wp)


# ---- window rule at wind = 4 and 8, and float32 ring vs float64 kernel ------------------------------
@pytest.mark.parametrize("wind", [4, 8])
@pytest.mark.parametrize("force", ["fast1", "wide", None])
def test_window_rule_other_window_lengths(solver, golden, wind, force):
    """wind = 4 / 8 stay register-resident (increment ring of wind - 2 slots): stop iterations and
    iterates equal the float64 oracle's (pybold/bold_signal.py:82-95 with that `wind`)."""
    g = golden("early_stop")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    assert "LDS" not in solver.which_kernel(300, 30, 5000, stop="window", wind=wind)
    Y, _ = synthetic(11, seed=20 + wind)
    Y[4] = torch.from_numpy(g["y"]).float().cuda()
    Yh = Y.cpu().numpy().astype(np.float64)
    fired = 0
    for tol in (0.1, 0.02, 0.005):
        out = [orc.deconv_fixed_lbda(Yh[v], hrf, 1.0, nb_iter=400, tol=tol, wind=wind, lipschitz=lip, dense=False)
               for v in range(len(Yh))]
        Wr, nr = np.stack([o[2] for o in out]), np.array([o[4] for o in out])
        fired += int((nr < 400).sum())
        W, J, nd = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, 400, want_J=True, stop="window", tol=tol, wind=wind,
                                      force=force)
        assert (nd.cpu().numpy() == nr).all(), (wind, force, tol, nd.cpu().numpy(), nr)
        assert rel_rows(W.cpu().numpy(), Wr) < EPS
    assert fired >= 11, fired                           # the rule did fire in a good part of the cases


def test_window_rule_float32_ring_agrees_with_float64_kernel(solver, golden):
    """Consecutive increments alternate in sign, so the window combination cancels: the batch kernels
    (float32 FIRs, float32 increment ring) must still stop where the all-float64 kernel does.
    Allowed mismatch: none of 3 x 3 x 64 cases may differ by more than one iteration, at most
    1 % may differ at all."""
    g = golden("early_stop")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    Y, _ = synthetic(64, seed=99)
    total, differ = 0, 0
    for tol in (1e-2, 3e-3, 1e-3):
        for lbda in (0.3, 1.0, 3.0):
            _, _, n64 = solver.fista_solve(Y.double(), hrf, lbda, 1.0 / lip, 2000, stop="window", tol=tol, wind=6)
            for force in ("fast1", "wide"):
                _, _, n32 = solver.fista_solve(Y, hrf, lbda, 1.0 / lip, 2000, stop="window", tol=tol, wind=6, force=force)
                d = (n32 - n64).abs()
                assert int(d.max()) <= 1, (tol, lbda, force, int(d.max()))
                total += d.numel()
                differ += int((d > 0).sum())
            assert int(n64.min()) < 2000
    assert differ <= 0.01 * total, (differ, total)


# ---- config 4: an outer iteration of the shared-HRF loop in three launches --------------------------
def test_fused_outer_iteration_pieces(solver):
    """pb_hrf_normal_eq_w == pb_hrf_normal_eq(pb_integ_op(w)) bit for bit, plus sum ||w||_1;
    pb_theta_fit_step == pb_theta_fit, plus 1 / pb_gram_frobenius of the new HRF and the
    normalised cost; an empty shard gives zeros."""
    t_r, dur, n = 0.75, 20.0, 300
    rng = np.random.RandomState(0)
    for V in (1, 7, 3000):
        W = dev64(rng.randn(V, n) * (rng.rand(V, n) < 0.1))
        Y = dev32(rng.randn(V, n))
        K = 27
        ne = solver.hrf_normal_eq(solver.integ_op(W), Y, K)
        msg = solver.hrf_normal_eq_w(W, Y, K)
        assert msg.shape == (K * K + K + 2,)
        assert torch.equal(msg[:-1], ne)
        np.testing.assert_allclose(float(msg[-1]), float(W.abs().sum()), rtol=1e-13)
        theta, f, taps = solver.theta_fit(ne, t_r, dur, (0.6, 1.9))
        th2, f2, taps2, step, jc = solver.theta_fit_step(msg, t_r, dur, (0.6, 1.9), n, 1.7)
        assert torch.equal(theta, th2) and torch.equal(f, f2) and torch.equal(taps[0], taps2)
        fro = solver.gram_frobenius_batch(taps, n)
        assert torch.equal(step, 1.0 / fro)
        np.testing.assert_allclose(float(jc), float((2.0 * f + 1.7 * msg[-1]) / msg[-2]), rtol=1e-14)
    empty = solver.hrf_normal_eq_w(dev64(np.zeros((0, n))), dev32(np.zeros((0, n))), 27)
    assert float(empty.abs().sum()) == 0.0


def test_bd_shared_fused_equals_unfused():
    """The three-launch outer iteration changes no number: theta after every outer iteration, the
    cost trace and the iterates equal those of the round-2 sequence of launches."""
    from pybold_amd import data, distributed
    from pybold_amd.hrf_model import spm_hrf
    t_r, dur = 0.75, 20.0
    h_true = spm_hrf(0.7, t_r, dur, False)[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(3000, dur=3.75, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0,
                                           std_dur=1.0, snr=10.0, seed=3)
    W1, h1, d1 = distributed.bd_shared(Y, t_r, lbda=1.7, theta_0=2.0, hrf_dur=dur, nb_iter=6, nb_inner=50)

    class Unfused(distributed.HipOps):
        fused = False
    W0, h0, d0 = distributed.bd_shared(Y, t_r, lbda=1.7, theta_0=2.0, hrf_dur=dur, nb_iter=6, nb_inner=50,
                                       ops=Unfused(t_r, dur, Y.shape[1]))
    assert np.array_equal(d1["theta"], d0["theta"])
    np.testing.assert_allclose(d1["J"], d0["J"], rtol=1e-13)
    assert torch.equal(W1, W0) and np.array_equal(h1, h0)


# ---- normal equations, one voxel per wave: every code path of normal_eq_wave_kernel ------------------
@pytest.mark.parametrize("n,k", [(300, 27), (5, 8), (1, 1), (33, 32), (64, 3), (513, 20), (1000, 30), (2000, 16),
                                 (2400, 16), (300, 33)])
def test_normal_equations_wave_form_shapes(solver, n, k):
    """Series shorter than the HRF, one sample, K = 32 (the form's limit; 33 taps and series whose four
    staging areas exceed the LDS fall back to the workgroup-per-voxel kernel), lengths beyond the
    register prefetch (> 512) and beyond the in-register cumulative sum (chunks of more than two
    samples), voxel counts below / across the number of waves; float32 and float64 series; the
    fused form (innovation in, cumulative sum inside) -- all against the float64 oracle."""
    rng = np.random.RandomState(1000 * n + k)
    for V in (1, 3, 9, 1030):
        if V * n > 600000:
            continue
        Wn = (rng.rand(V, n) < 0.1) * rng.randn(V, n)
        Z = np.cumsum(Wn, axis=1)
        Y = rng.randn(V, n)
        Y32 = Y.astype(np.float32).astype(np.float64)
        G, b, yy = orc.hrf_normal_eq(Z, Y, k)
        ref = np.concatenate([G.ravel(), b, [yy]])
        tol = dict(rtol=1e-11, atol=1e-11 * max(np.abs(ref).max(), 1e-300))
        ne = solver.hrf_normal_eq(dev64(Z), dev64(Y), k).cpu().numpy()
        np.testing.assert_allclose(ne, ref, **tol)
        G32, b32, yy32 = orc.hrf_normal_eq(Z, Y32, k)
        ref32 = np.concatenate([G32.ravel(), b32, [yy32]])
        ne32 = solver.hrf_normal_eq(dev64(Z), dev32(Y), k).cpu().numpy()
        np.testing.assert_allclose(ne32, ref32, **tol)
        msg = solver.hrf_normal_eq_w(dev64(Wn), dev32(Y), k).cpu().numpy()
        np.testing.assert_allclose(msg[:-1], ref32, **tol)
        np.testing.assert_allclose(msg[-1], np.abs(Wn).sum(), rtol=1e-12)
        # the fused form integrates with pb_integ_op's summation tree: bit-identical normal equations
        ne_unfused = solver.hrf_normal_eq(solver.integ_op(dev64(Wn)), dev32(Y), k)
        assert torch.equal(torch.from_numpy(msg[:-1]).cuda(), ne_unfused)
