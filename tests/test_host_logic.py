"""CPU-only tests of the host-side logic of pybold_amd (no kernel launches):
scalar sequences, Toeplitz helpers, Lipschitz constants, the HRF model, the
generic power iteration, sharding, and the "no CPU fallback" rule."""
import numpy as np
import pytest
import torch

import pybold_amd
from oracle import pybold_oracle as orc
from pybold_amd import distributed, solver


def test_momentum_sequence_matches_oracle():
    b = solver.momentum_betas(600)
    np.testing.assert_array_equal(b, orc.momentum_sequence(600))
    assert b[0] == 0.0 and 0.99 < b[-1] < 1.0
    # restart from an explicit t (chunked launches)
    t = 1.0
    for _ in range(7):
        t = 0.5 * (1.0 + np.sqrt(1.0 + 4.0 * t * t))
    np.testing.assert_array_equal(solver.momentum_betas(5, t0=t), b[7:12])


def test_toeplitz_helpers(golden):
    g = golden("operators")
    np.testing.assert_array_equal(pybold_amd.toeplitz_from_kernel(np.arange(1., 5.), 6, 6),
                                  g["toep_small"])
    np.testing.assert_array_equal(
        pybold_amd.toeplitz_from_kernel(g["rect_sig"], len(g["rect_k"]), len(g["rect_sig"])),
        g["rect_T"])
    k = np.random.RandomState(0).randn(27)
    H = pybold_amd.toeplitz_from_kernel(k, 240, 240)
    np.testing.assert_array_equal(pybold_amd.kernel_from_toeplitz(H), k)
    H[5, 100] = 1.0
    with pytest.raises(ValueError, match="Toeplitz"):
        pybold_amd.kernel_from_toeplitz(H)


def test_gram_frobenius_matches_dense_definition(golden):
    g = golden("loops_deconv")
    h, n = g["h"], len(g["y"])
    assert pybold_amd.gram_frobenius(h, n) == pytest.approx(orc.gram_lipschitz(h, n), rel=1e-13)
    assert pybold_amd.gram_frobenius(h[:3], 5) == pytest.approx(orc.gram_lipschitz(h[:3], 5), rel=1e-13)


def test_spm_hrf_matches_reference_values(golden):
    g = golden("spm_hrf")
    for i in range(6):
        delta, t_r, dur, norm = g["p%d" % i]
        h, t = pybold_amd.spm_hrf(delta, t_r=t_r, dur=dur, normalized_hrf=bool(norm))
        np.testing.assert_allclose(h, g["h%d" % i], rtol=1e-12, atol=1e-15)
        np.testing.assert_array_equal(t, g["t%d" % i])
    for bad in (0.49, 2.01):
        with pytest.raises(ValueError):
            pybold_amd.spm_hrf(bad)


def test_spectral_radius_est_is_duck_typed(golden):
    """Any object with .op/.adj works (reference contract, pybold/utils.py:94-109)."""
    g = golden("case1")
    H = orc.DenseH(g["hrf"], 300, 300)          # CPU stand-in for an operator object
    np.random.seed(0)
    rho = pybold_amd.spectral_radius_est(H, (300,))
    assert 0.9 * rho == pytest.approx(float(g["lipschitz"]), rel=1e-13)


def test_mad_daub_noise_est_properties():
    """db3 level-1 detail band + MAD (pybold/utils.py:10-25).  Unpinned against
    PyWavelets; checked against the filter's defining properties instead."""
    from pybold_amd.utils import _DB3_DEC_HI, mad, mad_daub_noise_est
    g = _DB3_DEC_HI
    assert abs(g.sum()) < 1e-10 and abs((g ** 2).sum() - 1.0) < 1e-10     # orthonormal high-pass
    assert abs((g * np.arange(6)).sum()) < 1e-9                           # vanishing moments
    assert abs((g * np.arange(6) ** 2).sum()) < 1e-9
    rng = np.random.RandomState(0)
    X = rng.randn(64, 300) * 2.5
    est = mad_daub_noise_est(X)
    np.testing.assert_allclose(est, [orc.mad_daub_noise_est(x) for x in X], rtol=1e-12)
    assert abs(est.mean() - 2.5) < 0.15                                   # consistent for white noise
    t = np.linspace(0, 1, 300)
    assert mad_daub_noise_est(3.0 + 2.0 * t + 0.5 * t ** 2) < 1e-8        # blind to quadratics
    assert mad(np.array([1., 2., 3., 4., 100.]), c=1.0) == 1.0
    assert mad_daub_noise_est(X[0]) == pytest.approx(est[0])              # 1-D form


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 100000, 100001):
        for world in (1, 2, 3, 8):
            spans = [distributed.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d
            assert max(b - a for a, b in spans) == -(-n // world)


def test_no_cpu_fallback_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    y = np.zeros(300)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pybold_amd.deconv(y, 1.0, np.ones(30), lbda=1.0, nb_iter=2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pybold_amd.DiscretInteg().op(y)
    with pytest.raises(TypeError):
        solver.fista_solve(torch.zeros(2, 300), np.ones(30), 1.0, 1.0, 2)


def test_unsupported_modes_raise():
    with pytest.raises(NotImplementedError):
        pybold_amd.ConvAndLinear(pybold_amd.DiscretInteg(), np.ones(3), 10, spectral_conv=True)


def test_product_never_imports_the_oracle():
    """The shipped package must not route through oracle/ (checked textually)."""
    import os
    pkg = os.path.dirname(pybold_amd.__file__)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                text = open(os.path.join(root, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
                assert "pybold_oracle" not in text and "fista_oracle" not in text, f


def test_bench_job_layout_and_self_launch_guard():
    """bench.py: strong scaling shards BASELINE config 3's 100k voxels over the ranks
    (12 500 per GPU at 8), weak gives every rank its own batch; `--gpus N` from a bare shell
    on a machine with fewer GPUs fails cleanly before anything touches a GPU."""
    import os
    import subprocess
    import sys
    import bench
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for world in (1, 2, 4, 8, 3, 7):
        spans = [bench.job_layout(100000, "strong", world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 100000 and all(s[2] == 100000 for s in spans)
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))          # contiguous, disjoint
        assert max(s[1] - s[0] for s in spans) == -(-100000 // world)
        weak = [bench.job_layout(100000, "weak", world, r) for r in range(world)]
        assert all(w[1] - w[0] == 100000 and w[2] == 100000 * world for w in weak)
    assert bench.job_layout(100000, "strong", 8, 3)[1] - bench.job_layout(100000, "strong", 8, 3)[0] == 12500
    assert bench.flops_per_voxel_iter(300, 30) == 39600.0
    import torch
    if torch.cuda.device_count() < 2:
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 2 and "needs 2 GPUs" in r.stderr and r.stdout.strip() == ""


def test_launch_plan_of_a_plain_solve():
    """pb_fista_plan is a host-only query: whole rounds of pair waves (16 384 problems on 256
    CUs; the same default when no device is visible), half a round + a concurrent remainder
    between 8 192 and 14 336 problems and after whole rounds, the cheapest single form below."""
    from pybold_amd import solver
    pair, row, wave = (solver.KERNEL_NAMES[k] for k in (2, 1, 3))
    expect = {1: (0, None, wave), 1000: (0, None, wave), 4096: (0, None, row), 5000: (4096, row, wave),
              6144: (4096, row, wave), 6200: (0, None, pair), 8192: (0, None, pair), 21000: (16384, pair, row),
              10000: (8192, pair, wave), 12500: (8192, pair, row), 13312: (8192, pair, row),
              14000: (8192, pair, row), 14400: (0, None, pair), 16384: (0, None, pair), 25000: (24576, pair, wave),
              29000: (24576, pair, row), 50000: (49152, pair, wave), 100000: (98304, pair, wave)}
    for P, want in expect.items():
        assert solver.launch_plan(300, 30, P, force="valu") == want, P
    # with the matrix-pipe forms (129..310 scans, up to 33 taps): whole rounds of 16 384 problems on the one-wave form
    # (round 3); what they leave (round 4): up to 4 608 problems on the vector forms' plan, up to half a round as ONE
    # pass of the split form (every series over two waves: half the latency), up to half a round + 2 048 as that pass
    # with one-problem waves beside and behind it, anything larger as one more pass of the one-wave form
    mm, m2 = solver.KERNEL_NAMES[4], solver.KERNEL_NAMES[5]
    expect = {1: (0, None, wave), 4096: (0, None, row), 4608: (4096, row, wave), 4609: (0, None, m2), 8192: (0, None, m2),
              8193: (8192, m2, wave), 10000: (8192, m2, wave), 10240: (8192, m2, wave), 10241: (0, None, mm),
              12500: (0, None, mm), 16384: (0, None, mm), 20000: (16384, mm, row), 21000: (16384, mm, m2),
              25000: (16384, mm, m2), 26624: (16384, mm, m2), 26625: (0, None, mm), 50000: (49152, mm, wave),
              100000: (98304, mm, wave)}
    for P, want in expect.items():
        assert solver.launch_plan(300, 30, P) == want, P
    # the window rule (as a certificate) rides the split form too
    assert solver.launch_plan(300, 30, 10000, stop="window") == (8192, m2, wave)
    # series of 321..640 scans: the split form from 5 120 problems on (whole passes of 8 192, remainders above 2 560)
    assert solver.launch_plan(600, 30, 50000) == (49152, m2, wave) and solver.launch_plan(600, 30, 8192) == (0, None, m2)
    assert solver.launch_plan(600, 30, 4096)[2] == pair and solver.launch_plan(600, 30, 11000) == (0, None, m2)
    assert solver.launch_plan(128, 16, 100000)[1] == pair and row in solver.launch_plan(300, 40, 100000)[1:]
    # the window rule at the reference's wind = 6 rides the pair form (no-fire certificate +
    # re-solve); the _loops_deconv rule does not; shapes outside the tables go to the LDS kernel
    assert solver.launch_plan(300, 30, 100000, stop="window", force="valu") == (98304, pair, wave)
    assert solver.launch_plan(300, 30, 100000, stop="window") == (98304, solver.KERNEL_NAMES[4], wave)
    n_main, main, tail = solver.launch_plan(300, 30, 100000, stop="loops")      # the exact rule inside the matrix-pipe form
    assert (n_main, main) == (98304, solver.KERNEL_NAMES[4]) and tail in (row, wave)
    assert solver.launch_plan(5000, 30, 100) == (0, None, solver.KERNEL_NAMES[0])


def test_large_batch_oracle_ops_match_the_pinned_ones():
    """`FastOracleOps` (C z-steps, normal equations from partial autocorrelations: the checker of the
    full-size config-4 test) == `OracleOps` (NumPy, pinned by the goldens)."""
    from oracle.shared_ops import FastOracleOps, OracleOps, hrf_normal_eq_blas
    rng = np.random.RandomState(0)
    for (V, n, K) in [(37, 300, 27), (3, 40, 30), (5, 20, 27), (1, 64, 1)]:
        Z, Y = rng.randn(V, n), rng.randn(V, n)
        G, b, yy = orc.hrf_normal_eq(Z, Y, K)
        G2, b2, yy2 = hrf_normal_eq_blas(Z, Y, K)
        np.testing.assert_allclose(G2, G, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(b2, b, rtol=1e-12, atol=1e-12)
        assert yy2 == pytest.approx(yy, rel=1e-14)
    t_r, dur, n = 0.75, 20.0, 300
    slow, fast = OracleOps(n, t_r, dur), FastOracleOps(n, t_r, dur, threads=2)
    taps = slow.hrf(torch.tensor([1.3]))
    Y = torch.from_numpy(rng.randn(6, n).astype(np.float32))
    W0 = torch.from_numpy(0.01 * rng.randn(6, n))
    Ws, Wf = slow.z_step(Y, taps, 0.4, 25, W0.clone()), fast.z_step(Y, taps, 0.4, 25, W0.clone())
    np.testing.assert_allclose(Wf.numpy(), Ws.numpy(), rtol=1e-10, atol=1e-14)
    np.testing.assert_allclose(fast.normal_eq(Wf, Y, 27).numpy(), slow.normal_eq(Ws, Y, 27).numpy(), rtol=1e-10)


def test_device_side_plans_tile_every_list_length():
    """Round 5: a call partitioned on the device lays each of its lists (dense class, sparse class, handed-back problems)
    over a STATIC sequence of candidate launches whose slot ranges a one-thread kernel computes from the list length
    (csrc/plan.h; the same functions, host build, behind pb_fista_list_plan).  For every list length: the candidates'
    ranges tile [0, n) exactly once, every range fits the grid the host sizes for that candidate, nothing lands on a
    candidate the host would not launch (no pair form / no one-problem-per-wave form / one stream), and a list of n problems
    gets the forms pb_fista_plan_ex gives a call of n problems."""
    import ctypes
    from pybold_amd import _lib
    lib = _lib.load()
    NC = 10
    MFMA, PAIR0, FAST0, MFMA2, PAIR1, FAST1, WIDE, SW0, SW1, SF = range(NC)
    rg, bd = (ctypes.c_int32 * (2 * NC))(), (ctypes.c_int32 * NC)()
    lengths = sorted(set(list(range(0, 20000, 7)) + list(range(16384 - 40, 16384 + 40)) + list(range(32768 - 40, 32768 + 40)) +
                         [4608, 4609, 8192, 8193, 9216, 9217, 10240, 10241, 12288, 14336, 98304, 100000, 123457, 1000000]))
    for kind in (1, 2):
        for has_pair in (0, 1):
            for has_wide in (0, 1):
                for one_stream in (0, 1):
                    for has_mfma2 in ((0, 1) if kind == 1 else (0,)):
                        for n in lengths:
                            rc = lib.pb_fista_list_plan(kind, n, 1000000, has_pair, has_wide, one_stream, has_mfma2, 2, rg, bd)
                            assert rc == 0, (kind, has_pair, has_wide, one_stream, has_mfma2, n, lib.pb_last_error())
                            r = [(rg[2 * c], rg[2 * c + 1]) for c in range(NC)]
                            live = sorted((a, b) for a, b in r if b > a)
                            assert (live[0][0] if live else 0) == 0 and (live[-1][1] if live else 0) == n
                            assert all(live[i][1] == live[i + 1][0] for i in range(len(live) - 1)), (n, r)
                            for c, (a, b) in enumerate(r):
                                assert b - a <= bd[c], (kind, n, c, r, list(bd))
                            if kind == 2 or not has_mfma2:
                                assert r[MFMA2] == (0, 0)
                            if kind == 2:
                                assert r[MFMA] == (0, 0)
                            if not has_pair:
                                assert r[PAIR0] == (0, 0) and r[PAIR1] == (0, 0)
                            if not has_wide:
                                assert r[WIDE] == (0, 0) and r[SW0] == (0, 0) and r[SW1] == (0, 0)
                            if one_stream:     # no side pieces, and no second piece of a form (capi.hip does not launch those then)
                                assert r[SW0] == (0, 0) and r[SW1] == (0, 0) and r[SF] == (0, 0)
                                assert r[PAIR1] == (0, 0) and r[FAST1] == (0, 0), (kind, n, r)
    # a partitioned call (kind 3): n_max problems, n of them dense -- the candidates tile the positions [0, n_max) of the call's
    # list array; the matrix-pipe forms stay inside the dense head
    for P in (4096, 10000, 20000, 100000):
        for n_d in sorted(set([0, 1, 4608, 4609, 8192, 8193, 9999, 16384, 16385, 30000, 98304, P] + list(range(0, P + 1, 977)))):
            if n_d > P:
                continue
            for one_stream in (0, 1):
                rc = lib.pb_fista_list_plan(3, n_d, P, 1, 1, one_stream, 1, 2, rg, bd)
                assert rc == 0, (P, n_d, lib.pb_last_error())
                r = [(rg[2 * c], rg[2 * c + 1]) for c in range(NC)]
                live = sorted((a, b) for a, b in r if b > a)
                assert live[0][0] == 0 and live[-1][1] == P and all(live[i][1] == live[i + 1][0] for i in range(len(live) - 1)), (P, n_d, r)
                assert all(b - a <= bd[c] for c, (a, b) in enumerate(r)), (P, n_d, r, list(bd))
                assert r[MFMA][1] <= n_d and r[MFMA2][1] <= n_d, (P, n_d, r)
                if one_stream:
                    assert r[PAIR1] == (0, 0) and r[FAST1] == (0, 0) and r[SW0] == (0, 0) and r[SF] == (0, 0), (P, n_d, r)
    # the plan of a list = the plan of a call of that many problems (N = 300, K = 30: pair form, one-problem waves, split form)
    nm, mf, tf = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    for n in (5000, 8192, 10000, 12500, 16384, 25000, 50000, 98304, 100000):
        lib.pb_fista_plan_ex(300, 30, n, 0, 6, _lib.PB_FLAG_NO_PARTITION, ctypes.byref(nm), ctypes.byref(mf), ctypes.byref(tf))
        lib.pb_fista_list_plan(1, n, n, 1, 1, 0, 1, 2, rg, bd)
        whole = rg[2 * MFMA + 1] - rg[2 * MFMA]
        if mf.value == 4:
            assert whole == nm.value, (n, whole, nm.value)
        else:
            assert nm.value == 0 or whole == 0, (n, whole, nm.value, mf.value)


def test_backtracking_statement_reduces_to_the_constant_step_recurrence(golden):
    """The opt-in backtracking mode's NumPy statement (its only checker; the reference has no backtracking): with a
    start step <= 1 / L every acceptance test passes at once and the iterates are those of the reference's recurrence; from
    a start step 16x too large it halves the step until the quadratic upper bound holds and the cost still decreases."""
    g = golden("case1")
    y, hrf, lip = g["y"], g["hrf"], float(g["lipschitz"])
    rho = lip / 0.9
    ref = orc.fista_batch(y[None, :], hrf, 1.0, 1.0 / rho, 60)
    W, step, halv, margin = orc.fista_backtrack_batch(y[None, :], hrf, 1.0, 1.0 / rho, 60)
    assert halv[0] == 0 and step[0] == 1.0 / rho
    np.testing.assert_allclose(W, ref, rtol=0, atol=1e-15)
    W2, step2, halv2, _ = orc.fista_backtrack_batch(y[None, :], hrf, 1.0, 16.0 / rho, 60)
    assert 3 <= halv2[0] <= 6 and step2[0] == 16.0 / rho * 0.5 ** halv2[0] and step2[0] <= 2.0 / rho

    def cost(w):
        r = orc.causal_conv(hrf, np.cumsum(w)) - y
        return 0.5 * r.dot(r) + np.abs(w).sum()
    assert cost(W2[0]) < cost(np.zeros_like(y)) and np.isfinite(W2).all()
