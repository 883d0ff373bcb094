"""Round 4: the parity edges of the matrix-pipe kernel.

  * config 4's WHOLE loop (21 z-steps, normal equations, theta fits) on the path the 50k-voxel
    benchmark runs -- `fista_mfma_kernel<..., TAPS_DEV>` with PB_FLAG_NO_RHO_GUARD on the intermediate
    z-steps -- against an oracle loop on the SAME batch of 17 000 voxels: the dilation after every one
    of the 20 outer iterations, and diff_z / z / x of every voxel at the end;
  * (test_adversarial_*) data unlike the generator's through the default dispatch at one matrix-pipe
    round and more.

Reference: pybold/bold_signal.py:281-382 (bd), :217-222 (hrf_fit_err), :62-72 (the recurrence).
"""
import numpy as np
import pytest
import torch

from oracle import pybold_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    from pybold_amd import solver as s
    return s


def rel_rows(a, b):
    return np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-300)


class _OneProcess:                                   # the oracle loop never talks to other ranks
    world_size, rank = 1, 0

    @staticmethod
    def allreduce_(t):
        return t


def test_config4_whole_loop_on_the_matrix_pipe_against_the_oracle(solver):
    """BASELINE config 4 (TR 0.75 s, N = 300, 20 s HRF = 27 taps, true dilation 0.7, theta_0 = 2.0,
    lambda = 1.7, 20 outer x 100 inner iterations + closing z-step, Frobenius step) on 17 000 voxels:
    one matrix-pipe round (16 384 voxels on `fista_mfma_kernel<10, ..., TAPS_DEV>`, the accuracy guard
    off on the 20 intermediate z-steps exactly as in the 50k-voxel run) + 616 voxels on the vector
    forms.  The oracle runs the same loop on the same batch from its own state (C float64 z-steps,
    NumPy normal equations and 1-D search): theta after EVERY outer iteration within 1e-6, and
    diff_z, z, x of EVERY voxel within 1e-5 at the end.  The same run with the guard on says what
    "the next warm-started z-step forgets such errors" is worth in numbers."""
    from oracle.shared_ops import FastOracleOps
    from pybold_amd import data, distributed
    t_r, dur, lbda, V4, N = 0.75, 20.0, 1.7, 17000, 300
    h_true = orc.spm_hrf(0.7, t_r, dur, False)[0]
    K = len(h_true)
    Y4, _, _ = data.gen_rnd_bloc_bold_batch(V4, dur=3.75, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0,
                                            std_dur=1.0, snr=10.0, seed=44)
    assert Y4.shape == (V4, N) and K == 27
    # the z-steps of this batch do run on the matrix pipe (whole round) + a vector remainder
    n_main, main, tail = solver.launch_plan(N, K, V4)
    assert n_main == 16384 and "matrix pipe" in main and "matrix pipe" not in tail

    Wg, hg, dg = distributed.bd_shared(Y4, t_r, lbda=lbda, theta_0=2.0, hrf_dur=dur, nb_iter=20, nb_inner=100)

    class Guarded(distributed.HipOps):               # every z-step with the accuracy guard (and its re-solve)
        def z_step(self, Y, taps, lbda, nb_inner, W, step=None, last=True):
            return super().z_step(Y, taps, lbda, nb_inner, W, step, last=True)
    Wq, hq, dq = distributed.bd_shared(Y4, t_r, lbda=lbda, theta_0=2.0, hrf_dur=dur, nb_iter=20, nb_inner=100,
                                       ops=Guarded(t_r, dur, N))
    # how many voxels the guard hands back in an intermediate z-step (the first one: iterate tiny against the threshold)
    taps0 = torch.from_numpy(orc.spm_hrf(2.0, t_r, dur, False)[0].copy()).cuda()
    step0 = 1.0 / solver.gram_frobenius_batch(taps0.reshape(1, -1), N)
    _, nd = solver.fista_solve_pp(Y4, taps0, step0, lbda, 100, force="noresolve")
    back_first = float((nd[:16384] < 0).double().mean())

    Wo, ho, do = distributed.bd_shared(Y4.cpu(), t_r, lbda=lbda, theta_0=2.0, hrf_dur=dur, nb_iter=20, nb_inner=100,
                                       ops=FastOracleOps(N, t_r, dur), comm=_OneProcess())
    tg, tq, to = (np.asarray(d["theta"], dtype=np.float64) for d in (dg, dq, do))
    assert len(to) == 21
    dth, dth_q = np.abs(tg - to), np.abs(tq - to)
    Wo = Wo.numpy()
    Zo = np.cumsum(Wo, axis=1)
    Xo = orc.causal_conv(ho, Zo)
    errs = {}
    for tag, W, h in (("guard off (as benchmarked)", Wg, hg), ("guard on", Wq, hq)):
        X, Z = solver.fista_outputs(W, h)
        errs[tag] = [float(rel_rows(a.cpu().numpy(), b).max()) for a, b in ((W, Wo), (Z, Zo), (X, Xo))]
    print("config 4, 17 000 voxels, whole loop vs oracle: max |dtheta| over 20 outer iterations %.2e (guard on: %.2e); "
          "diff_z / z / x %s; guard on %s; first z-step hands back %.1f %% of the round with the guard on"
          % (dth.max(), dth_q.max(), ["%.1e" % e for e in errs["guard off (as benchmarked)"]],
             ["%.1e" % e for e in errs["guard on"]], 100 * back_first))
    assert dth.max() < 1e-6, dth
    assert dth_q.max() < 1e-6, dth_q
    np.testing.assert_allclose(dg["J"], do["J"], rtol=1e-6)
    for tag in errs:
        assert max(errs[tag]) < 1e-5, (tag, errs[tag])
    np.testing.assert_allclose(hg, orc.spm_hrf(float(tg[-1]), t_r, dur, False)[0], rtol=1e-10, atol=1e-14)   # h = h(theta)
    np.testing.assert_allclose(hg, ho, rtol=0, atol=1e-6)


def test_oracle_ops_for_large_batches_match_the_pinned_ones():
    """The C z-step + partial-autocorrelation normal equations used above == the NumPy OracleOps (pinned
    through tests/test_oracle_golden.py) on a small batch, whole loop."""
    from oracle.shared_ops import FastOracleOps, OracleOps
    from pybold_amd import data, distributed
    t_r, dur = 0.75, 20.0
    h_true = orc.spm_hrf(0.7, t_r, dur, False)[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(24, dur=3.75, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0, std_dur=1.0,
                                           snr=10.0, seed=3)
    out = [distributed.bd_shared(Y.cpu(), t_r, lbda=1.7, theta_0=2.0, hrf_dur=dur, nb_iter=3, nb_inner=60, ops=ops,
                                 comm=_OneProcess()) for ops in (OracleOps(300, t_r, dur), FastOracleOps(300, t_r, dur))]
    np.testing.assert_allclose(out[0][2]["theta"], out[1][2]["theta"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(out[0][0].numpy(), out[1][0].numpy(), rtol=1e-7, atol=1e-12)


@pytest.mark.parametrize("n,k", [(300, 30), (129, 1), (160, 2), (320, 33), (300, 48)])
def test_adversarial_series_through_the_default_dispatch(solver, n, k):
    """Data unlike the generator's through the DEFAULT dispatch at one matrix-pipe round + a vector remainder
    (18 432 problems per call): DC baselines 10x / 100x / 1000x the fluctuation, Student-t noise, SNR -10 and
    +30 dB, constant / single-spike / all-zero series (tests/adversarial_data.py), lambda / lambda_max in
    {1e-3, 1e-1, 1}, 500 iterations.  A random sample of every family against the C float64 oracle: diff_z, z
    and x within eps = 1e-5 (diff_z[1:] too for the DC families, whose sample 0 carries the baseline) -- for
    the problems the matrix-pipe kernel kept AND for those its guards handed back to the float32 operators;
    all-zero solutions are all-zero; away from lambda_max the guards hand back (almost) nothing.  The full table
    (4 lambdas, 192 samples per family) is profiles/r4_adversarial_sweep.txt (tools/r4_adversarial_sweep.py)."""
    from oracle import c_oracle
    from tests.adversarial_data import DC_FAMILIES, FAMILIES, hrf_for, make_batch
    dev = torch.device("cuda")
    hrf = hrf_for(k)
    A = orc.toeplitz_from_kernel(hrf, n, n).dot(np.tril(np.ones((n, n))))
    step = 1.0 / (0.9 * np.linalg.norm(A, 2) ** 2)
    Y, fam = make_batch(n, hrf, 100 + n + k, dev)
    P = Y.shape[0]
    n_main, main, _ = solver.launch_plan(n, k, P)
    assert n_main == 16384 and "matrix pipe" in main            # the call does run one matrix-pipe round
    rng = np.random.RandomState(n * 100 + k)
    samp = np.sort(np.concatenate([rng.choice(np.nonzero(fam == f)[0], 48, replace=False) for f in range(len(FAMILIES))]))
    sel = torch.from_numpy(samp).to(dev)
    Ys = Y[sel].cpu().numpy().astype(np.float64)
    for c in (1e-3, 1e-1, 1.0):
        W, _, nd = solver.fista_solve(Y, hrf, c, step, 500)
        assert bool(torch.isfinite(W).all()) and int(nd.min()) == 500
        _, _, nd0 = solver.fista_solve(Y, hrf, c, step, 500, force="noresolve")
        back = (nd0 < 0).cpu().numpy()
        if c < 1.0:
            assert back.mean() < 0.03, (c, back.mean())
        Wo, _, _ = c_oracle.fista_batch(Ys, hrf, c, step, 500, threads=0)
        Xg, Zg = solver.fista_outputs(W[sel].contiguous(), hrf)
        Wg, Xg, Zg = W[sel].cpu().numpy(), Xg.cpu().numpy(), Zg.cpu().numpy()
        Xo, Zo = orc.fista_outputs(Wo, hrf)
        zero = np.linalg.norm(Wo, axis=1) == 0
        if zero.any():
            assert np.abs(Wg[zero]).max() == 0.0
        ok = ~zero
        for name, a, b in (("diff_z", Wg, Wo), ("z", Zg, Zo), ("x", Xg, Xo)):
            e = rel_rows(a[ok], b[ok])
            assert e.max() < 1e-5, (n, k, c, name, float(e.max()), FAMILIES[int(fam[samp][ok][int(e.argmax())])],
                                    "handed back" if back[samp][ok][int(e.argmax())] else "kept")
        dc = ok & np.isin(fam[samp], DC_FAMILIES)
        e = rel_rows(Wg[dc][:, 1:], Wo[dc][:, 1:])
        assert e.max() < 1e-5, (n, k, c, "diff_z[1:] of the DC families", float(e.max()))


# ---- the matrix-pipe form with every series split over two waves (fista_mfma2.h) ---------------------------------
@pytest.mark.parametrize("n,k", [(300, 30), (320, 33), (160, 27), (129, 16), (225, 27), (289, 2), (321, 30), (330, 16),
                                 (400, 27), (480, 30), (577, 30), (600, 30), (608, 30), (640, 33),
                                 (330, 34), (400, 40), (600, 42), (640, 48)])      # (round 5: 34+ taps, three near tiles)
def test_split_matrix_pipe_form_matches_oracle(solver, n, k):
    """`fista_mfma2_kernel`: 16 problems per workgroup of two waves, the left wave owning the first
    ceil(nb / 2) blocks of 32 samples and the right wave the rest (5 <= nb <= 20: 129 .. 640 scans -- the
    reference's shipped demo is 600, examples/synth_data/deconv.py:46); warm and cold start, against the C float64
    oracle; nothing handed back on ordinary data; any batch position gives the same bits; for N <= 320 equal to
    the one-wave form within the split-operand rounding."""
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
    Wc, _, ndc = solver.fista_solve(Yd, hrf, 0.3, 1.0 / lip, 200, force="mfma2only")       # cold start
    Woc, _, _ = c_oracle.fista_batch(Yh, hrf, 0.3, 1.0 / lip, 200, threads=4)
    assert int(ndc.min()) == 200 and rel_rows(Wc.cpu().numpy(), Woc).max() < 3e-6
    # another batch position, another workgroup: the same bits
    perm = np.r_[np.arange(23, 40), np.arange(23)]
    W2, _, _ = solver.fista_solve(Yd[torch.from_numpy(perm).cuda()].contiguous(), hrf, 0.3, 1.0 / lip, 200,
                                  W0=W0d[torch.from_numpy(perm).cuda()].contiguous(), force="mfma2only")
    assert torch.equal(W2, W[torch.from_numpy(perm).cuda()])
    if n <= 320:
        W1, _, _ = solver.fista_solve(Yd, hrf, 0.3, 1.0 / lip, 200, W0=W0d, force="mfmaonly")
        assert rel_rows(W.cpu().numpy(), W1.cpu().numpy()).max() < 3e-6
    # the library's own dispatch for this shape: long series take the split form from 5 120 problems on
    if n > 320:
        assert "split over two" in solver.which_kernel(n, k, 6000)
        Yb = torch.from_numpy(np.tile(Yv, (150, 1)).astype(np.float32)).cuda()             # 6 000 series
        # (white noise: since round 5 the conditioning guard keeps about half of such series off the matrix pipe -- "noill"
        # = the partitioned dispatch without that guard, to see the split form's own bits; the guarded default against the oracle)
        Wl, _, ndl = solver.fista_solve(Yb, hrf, 0.3, 1.0 / lip, 200, force="noill")
        assert int(ndl.min()) == 200 and torch.equal(Wl[:40], Wc) and torch.equal(Wl[-40:], Wc)
        Wg, _, ndg = solver.fista_solve(Yb, hrf, 0.3, 1.0 / lip, 200)
        assert int(ndg.min()) == 200 and rel_rows(Wg[:40].cpu().numpy(), Woc).max() < 3e-6 and torch.equal(Wg[:40], Wg[-40:])


def test_split_matrix_pipe_form_guards_and_shared_hrf(solver, golden):
    """The guards of the split form see the WHOLE series (both waves agree): a warm start outside the float16
    range in the left half only / the right half only, NaN, sparse solutions, an all-zero series -- handed back
    and re-solved to the oracle's answer; the shared-HRF z-step (taps and step read from device memory) through
    `pb_fista_solve_pp` on a batch between a quarter and half a round."""
    from oracle import c_oracle
    from pybold_amd import data
    g = golden("case1")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    Y, _, _ = data.gen_rnd_bloc_bold_batch(48, dur=5.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0,
                                           snr=1.0, seed=6)
    Y[7] = 0.0
    Yh = Y.cpu().numpy().astype(np.float64)
    rng = np.random.RandomState(2)
    W0 = 1e-3 * rng.randn(48, 300)
    W0[3, :100] *= 1e9                                      # out of range in the left wave's half only
    W0[5, 200:] *= 1e9                                      # ... in the right wave's half only
    W0[21, 250] = np.nan
    lmax = solver.lambda_max(Y, hrf).cpu().numpy()
    lam = np.where(np.arange(48) % 2 == 0, 1.0, 0.7 * lmax)
    lam[[3, 5, 7, 21]] = 1.0
    Wo, _, _ = c_oracle.fista_batch(Yh, hrf, lam, 1.0 / lip, 120, W0=W0, threads=8)
    W0d = torch.from_numpy(W0).cuda()
    W, _, nd = solver.fista_solve(Y, hrf, lam, 1.0 / lip, 120, W0=W0d, force="mfma2")
    Wn = W.cpu().numpy()
    ok = (np.arange(48) != 21) & (np.linalg.norm(Wo, axis=1) > 0)
    assert (nd.cpu().numpy() == 120).all() and rel_rows(Wn[ok], Wo[ok]).max() < 1e-5 and np.isnan(Wn[21]).any()
    _, _, ndc = solver.fista_solve(Y, hrf, lam, 1.0 / lip, 120, W0=W0d, force="mfma2only")
    caught = np.flatnonzero(ndc.cpu().numpy() == -1)
    assert {3, 5, 21} <= set(caught) and (ndc.cpu().numpy()[[0, 2, 6, 8]] == 120).all()
    assert np.isin(caught, np.r_[3, 5, 21, np.arange(1, 48, 2)]).all() and len(caught) >= 10
    # shared HRF from device memory: 6 000 voxels = one pass of the split form (the config-4 shard of one of 8 GPUs)
    t_r, dur, n = 0.75, 20.0, 300
    h = orc.spm_hrf(0.9, t_r, dur, False)[0]
    lip4 = orc.gram_lipschitz(h, n)
    Yb = torch.from_numpy(rng.randn(6000, n).astype(np.float32)).cuda()
    taps, stepc = torch.from_numpy(h.copy()).cuda(), torch.from_numpy(np.array([1.0 / lip4])).cuda()
    Wp, ndp = solver.fista_solve_pp(Yb, taps, stepc, 1.7, 60)
    assert int(ndp.min()) == 60 and "split over two" in solver.launch_plan(n, len(h), 6000)[2]
    idx = np.r_[0, 15, 16, 5999, rng.choice(6000, 12, replace=False)]
    Wop = orc.fista_batch(Yb.cpu().numpy()[idx].astype(np.float64), h, 1.7, 1.0 / lip4, 60)
    assert rel_rows(Wp.cpu().numpy()[idx], Wop).max() < 1e-5
    Wv, _ = solver.fista_solve_pp(Yb, taps, stepc, 1.7, 60, force="valu")
    assert float(((Wp - Wv).norm(dim=1) / Wv.norm(dim=1)).max()) < 4e-6


# ---- regularisation path partitioned on the device (pb_fista_solve_path) -------------------------------------------
def test_regularisation_path_partition(solver, golden):
    """BASELINE config 5's workload (voxel x 20 lambdas = logspace(-2, 0, 20) x lambda_max,v, y shared) through
    `pb_fista_solve_path`: the dense class on the matrix-pipe form, the sparse class on the pair form, the lists built
    on the device.  Every problem equals the plain per-problem-lambda dispatch (vector forms) within the split-operand
    rounding and the C float64 oracle within eps, across all 20 lambda indices; the top of the path is exactly 0;
    partition ratios from "everything sparse" to "everything dense" (the guards then hand the sparse third back),
    ragged sizes, one lambda per series, an all-zero series."""
    from oracle import c_oracle
    from pybold_amd import data
    g = golden("case1")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    V, L, N = 2100, 20, 300
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=5.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=9)
    Y[17] = 0.0
    lmax = solver.lambda_max(Y, hrf)
    grid = torch.logspace(-2.0, 0.0, L, dtype=torch.float64, device="cuda")
    lam = (lmax[:, None] * grid[None, :]).reshape(-1)
    lam[17 * L:18 * L] = 1.0
    P = V * L
    Wv, _, ndv = solver.fista_solve(Y, hrf, lam, 1.0 / lip, 300, y_rep=L, force="valu")       # (the vector dispatch)
    W, _, nd = solver.fista_solve(Y, hrf, lam, 1.0 / lip, 300, y_rep=L, lmax=lmax)
    # without the caller's lambda_max the library makes its own (float32 pass) and partitions the same way (round 5)
    W2, _, _ = solver.fista_solve(Y, hrf, lam, 1.0 / lip, 300, y_rep=L)
    assert float(((W2 - W).norm(dim=1) / W.norm(dim=1).clamp_min(1e-300)).max()) < 4e-6
    assert int(nd.min()) == 300 and int(nd.max()) == 300 and bool(torch.isfinite(W).all())
    nrm = Wv.norm(dim=1)
    assert bool((W[nrm == 0] == 0).all())                      # lambda_max and the all-zero series: exactly 0
    assert float(((W - Wv).norm(dim=1)[nrm > 0] / nrm[nrm > 0]).max()) < 4e-6
    # the dense class did run on the matrix pipe: different bits from the vector forms there, equal bits in the sparse class
    dense = (lam < 0.13 * lmax.repeat_interleave(L))
    assert 0.45 < float(dense.double().mean()) < 0.65
    assert not bool((W[dense] == Wv[dense]).all())
    _, _, nd0 = solver.fista_solve(Y, hrf, lam, 1.0 / lip, 300, y_rep=L, lmax=lmax, force="noresolve")
    assert float((nd0[dense] < 0).double().mean()) < 0.03 and int((nd0[~dense] < 0).sum()) == 0
    rng = np.random.RandomState(3)
    idx = np.sort(np.concatenate([rng.choice(V, 6, replace=False) * L + i for i in range(L)]))     # every lambda index
    Ys = Y[torch.from_numpy(idx // L).cuda()].cpu().numpy().astype(np.float64)
    Wo, _, _ = c_oracle.fista_batch(Ys, hrf, lam.cpu().numpy()[idx], 1.0 / lip, 300, threads=0)
    ok = np.linalg.norm(Wo, axis=1) > 0
    assert rel_rows(W.cpu().numpy()[idx][ok], Wo[ok]).max() < 1e-5
    # other ratios: all sparse (bitwise the vector dispatch), all dense (the guards + re-solve take care of the sparse third)
    Ws, _, _ = solver.fista_solve(Y, hrf, lam, 1.0 / lip, 300, y_rep=L, lmax=lmax, dense_ratio=1e-9)
    assert float(((Ws - Wv).norm(dim=1)[nrm > 0] / nrm[nrm > 0]).max()) < 2e-6     # (pair form vs the plan of the default dispatch)
    Wd, _, ndd = solver.fista_solve(Y, hrf, lam, 1.0 / lip, 300, y_rep=L, lmax=lmax, dense_ratio=1e9)
    assert int(ndd.min()) == 300 and float(((Wd - Wv).norm(dim=1)[nrm > 0] / nrm[nrm > 0]).max()) < 4e-6
    # ragged: one lambda per series (y_rep = 1), a count that fills no block, warm start
    lam1 = lmax * torch.from_numpy(rng.choice([0.01, 0.05, 0.3, 0.9], V)).cuda()
    W0 = torch.from_numpy(1e-3 * rng.randn(V - 3, N)).cuda()
    Wa, _, nda = solver.fista_solve(Y[:V - 3], hrf, lam1[:V - 3], 1.0 / lip, 120, W0=W0, lmax=lmax[:V - 3])
    Wb, _, _ = solver.fista_solve(Y[:V - 3], hrf, lam1[:V - 3], 1.0 / lip, 120, W0=W0)
    nb = Wb.norm(dim=1)
    assert int(nda.min()) == 120 and float(((Wa - Wb).norm(dim=1)[nb > 0] / nb[nb > 0]).max()) < 4e-6


# ---- the _loops_deconv stop rule inside the matrix-pipe kernel ------------------------------------------------------
def test_loops_rule_inside_the_matrix_pipe_kernel(solver, golden):
    """`_loops_deconv`'s criterion ||w_{k+1} - u_k|| / (||w_{k+1}|| + 1e-10) < tol (pybold/bold_signal.py:267-273)
    evaluated in float64 inside `fista_mfma_kernel<..., LOOPS>` (two float64 sums per sample beside the update) ON THE
    ITERATE OF THE 22-BIT OPERATORS (relative error ~1e-6: a criterion within that of `tol` may cross one iteration
    earlier or later than in the float64 reference -- allowed for, below): the
    iteration every problem stops at and its iterate equal the float64 oracle's -- on the golden series, and on a
    1 000-case sweep (series x lambda x tolerance: about half of the problems stop early, spread over the run);
    the library's dispatch puts whole rounds of such solves on that kernel."""
    from pybold_amd import data
    g = golden("loops_deconv")
    y, h = g["y"], g["h"]
    n = len(y)
    lip = orc.gram_lipschitz(h, n)
    Yb = np.stack([y, 0.3 * y, 3.0 * y, -y, 0.05 * y])
    Yd = torch.from_numpy(Yb.astype(np.float32)).cuda()
    Y32 = Yb.astype(np.float32).astype(np.float64)
    for tol in (1e-2, 1e-3, 2e-2):
        Wo, ndo = orc.loops_batch(Y32, h, 1.7, 1.0 / lip, 100, tol)
        W, _, nd = solver.fista_solve(Yd, h, 1.7, 1.0 / lip, 100, stop="loops", tol=tol, force="mfma")
        same = nd.cpu().numpy() == ndo
        assert np.abs(nd.cpu().numpy() - ndo).max() <= 1, (tol, nd.cpu().numpy(), ndo)
        assert rel_rows(W.cpu().numpy()[same], Wo[same]).max() < 1e-5
    assert solver.launch_plan(n, len(h), 100000, stop="loops")[1].startswith("fista_mfma_kernel")
    # sweep: 250 series x 4 (lambda, tolerance) pairs, N = 300, K = 30: stops spread over iterations 40 .. 400,
    # a quarter of the problems never stop
    g1 = golden("case1")
    hrf, lip1 = g1["hrf"], float(g1["lipschitz"])
    rng = np.random.RandomState(12)
    Z = np.zeros((250, 300))
    for v in range(250):
        for _ in range(5):
            o = rng.randint(0, 280)
            Z[v, o:o + rng.randint(8, 16)] = 1.0
    X = orc.causal_conv(hrf, Z)
    noise = rng.randn(250, 300)
    noise *= (np.linalg.norm(X, axis=1) / np.linalg.norm(noise, axis=1) / np.sqrt(10 ** (np.linspace(-5, 20, 250) / 10)))[:, None]
    Y = torch.from_numpy((X + noise).astype(np.float32)).cuda()
    Yh = Y.cpu().numpy().astype(np.float64)
    cases = early = 0
    for lbda, tol in ((0.5, 1e-4), (0.5, 2e-4), (2.0, 2.5e-4), (2.0, 2.7e-4)):
        Wo, ndo = orc.loops_batch(Yh, hrf, lbda, 1.0 / lip1, 400, tol)
        W, _, nd = solver.fista_solve(Y, hrf, lbda, 1.0 / lip1, 400, stop="loops", tol=tol, force="mfma")
        same = nd.cpu().numpy() == ndo                      # (measured: 1 000 / 1 000 equal; +-1 allowed on 1 % at most)
        assert np.abs(nd.cpu().numpy() - ndo).max() <= 1 and same.mean() >= 0.99, (lbda, tol, int((~same).sum()))
        assert rel_rows(W.cpu().numpy()[same], Wo[same]).max() < 1e-5
        cases += len(ndo)
        early += int((ndo < 400).sum())
    assert cases == 1000 and 0.5 < early / cases < 0.95 and len(set(ndo.tolist())) > 20, early
    # through the default dispatch: whole round on the matrix pipe + remainder on the vector forms, same rule
    Yl = Y.repeat(70, 1)[:16384 + 500].contiguous()
    Wl, _, ndl = solver.fista_solve(Yl, hrf, 2.0, 1.0 / lip1, 400, stop="loops", tol=2.7e-4)
    _, ndo = orc.loops_batch(Yh, hrf, 2.0, 1.0 / lip1, 400, 2.7e-4)
    assert (ndl.cpu().numpy() == ndo[np.arange(16384 + 500) % 250]).all()      # (the remainder: single-row / one-per-wave forms)
    assert torch.equal(Wl[:250], Wl[250:500])


@pytest.mark.parametrize("n,k", [(600, 30), (400, 27), (640, 33), (300, 30), (161, 16), (600, 42), (400, 48), (300, 42), (240, 48)])
def test_split_form_cost_trace_and_window_rule(solver, n, k):
    """`fista_mfma2_kernel<..., WITH_J, CERT>`: the cost trace (each wave adds up its half of ||r||^2 and ||w||_1, the
    halves meet at the barrier of the forward pass) against the float64 oracle's; the window rule (wind = 6) as a
    no-fire certificate with eight tracked samples per problem -- stop iterations and iterates of the exact rule, about
    half of the series firing before n_iter (the uncleared ones are re-solved exactly by the library); at the default
    tolerance nothing fires and the result is the plain + cost-trace solve, bit for bit."""
    from oracle import c_oracle
    rng = np.random.RandomState(n * 7 + k)
    hrf = orc.spm_hrf(1.0, 1.0, float(k), False)[0][:k] if k >= 20 else np.hanning(k + 2)[1:-1] * 0.3
    lip = orc.gram_lipschitz(hrf, n)
    Yv = rng.randn(24, n)
    Y32 = Yv.astype(np.float32).astype(np.float64)
    Yd = torch.from_numpy(Yv.astype(np.float32)).cuda()
    Wo, Jo, _ = c_oracle.fista_batch(Y32, hrf, 0.7, 1.0 / lip, 200, want_J=True, threads=4)
    W, J, nd = solver.fista_solve(Yd, hrf, 0.7, 1.0 / lip, 200, want_J=True, force="mfma2only")
    assert int(nd.min()) == 200 and rel_rows(W.cpu().numpy(), Wo).max() < 3e-6
    np.testing.assert_allclose(J.cpu().numpy(), Jo, rtol=3e-5)
    # the window rule: oracle stop iterations for a tolerance at which some series fire
    tol = 0.01
    n_fire = np.array([orc.deconv_fixed_lbda(Y32[v], hrf, 0.7, nb_iter=400, tol=tol, lipschitz=lip, dense=False)[4] for v in range(24)])
    n_iter = int(np.median(n_fire))
    out = [orc.deconv_fixed_lbda(Y32[v], hrf, 0.7, nb_iter=n_iter, tol=tol, lipschitz=lip, dense=False) for v in range(24)]
    Wr, nr = np.stack([o[2] for o in out]), np.array([o[4] for o in out])
    Ww, Jw, ndw = solver.fista_solve(Yd, hrf, 0.7, 1.0 / lip, n_iter, want_J=True, stop="window", tol=tol, wind=6, force="mfma2cert")
    assert (ndw.cpu().numpy() == nr).all(), (ndw.cpu().numpy(), nr)
    assert rel_rows(Ww.cpu().numpy(), Wr).max() < 1e-5
    # what the certificate alone clears: every series that does not fire must not be flagged wrongly as "fired"
    _, _, ndc = solver.fista_solve(Yd, hrf, 0.7, 1.0 / lip, n_iter, want_J=True, stop="window", tol=tol, wind=6, force="mfma2certonly")
    ndc = ndc.cpu().numpy()
    assert ((ndc == -1) | (ndc == n_iter)).all() and (ndc[nr < n_iter] == -1).all()
    # the default tolerance: nothing fires, nothing is handed back, the plain + J result bit for bit
    Wd, Jd, ndd = solver.fista_solve(Yd, hrf, 0.7, 1.0 / lip, 200, want_J=True, stop="window", tol=1e-6, wind=6, force="mfma2certonly")
    assert int(ndd.min()) == 200 and torch.equal(Wd, W) and torch.equal(Jd, J)
    # (round 5: also 225 .. 310 scans with 34+ taps -- the one-wave form has no certificate beside three near tiles, the split form has)
    if n > 320 or k > 33:                           # the library's own dispatch for the reference-default call on long series
        assert "split over two" in solver.which_kernel(n, k, 6000, want_J=True, stop="window")
        Yb = torch.from_numpy(np.tile(Yv, (250, 1)).astype(np.float32)).cuda()
        Wl, Jl, ndl = solver.fista_solve(Yb, hrf, 0.7, 1.0 / lip, 200, want_J=True, stop="window", tol=1e-6, wind=6, force="noill")
        assert int(ndl.min()) == 200 and torch.equal(Wl[:24], W) and torch.equal(Wl[-24:], W)      # (see above: white noise)
        Wg, Jg, ndg = solver.fista_solve(Yb, hrf, 0.7, 1.0 / lip, 200, want_J=True, stop="window", tol=1e-6, wind=6)
        assert int(ndg.min()) == 200 and rel_rows(Wg[:24].cpu().numpy(), Wo).max() < 3e-6
