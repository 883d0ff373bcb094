"""torch.ops.pybold_hip (pybold_amd/csrc/torch_ops.cpp): the TORCH_LIBRARY shim over the C ABI.
CPU: the shim loads and registers every operator; GPU: results are bit-identical to the ctypes path."""
import numpy as np
import pytest
import torch


def test_shim_loads_and_registers_every_operator():
    from pybold_amd import torch_ops
    ops = torch_ops.load()
    for name in ("fista_solve", "fista_outputs", "op_forward", "op_adjoint", "hrf_normal_eq", "theta_fit"):
        assert hasattr(ops, name), name
    # a CPU tensor has no kernel behind these operators: the dispatcher says so, nothing computes on the host
    with pytest.raises((RuntimeError, NotImplementedError)):
        ops.op_forward(torch.zeros(2, 8, dtype=torch.float64), torch.ones(3, dtype=torch.float64), 8)


@pytest.mark.gpu
def test_torch_ops_equal_the_ctypes_path(golden):
    from pybold_amd import solver, torch_ops
    from pybold_amd.hrf_model import spm_hrf
    ops = torch_ops.load()
    g = golden("case1")
    hrf, lip = g["hrf"], float(g["lipschitz"])
    rng = np.random.RandomState(0)
    Y = torch.from_numpy(rng.randn(20000, 300).astype(np.float32)).cuda()
    for kw in (dict(), dict(want_J=True), dict(want_J=True, stop="window", tol=1e-6)):
        W, J, nd = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, 40, **kw)
        W2, J2, nd2 = torch_ops.fista_solve(Y, hrf, 1.0, 1.0 / lip, 40, want_J=kw.get("want_J", False),
                                            stop_mode=solver._STOP[kw.get("stop")], tol=kw.get("tol", 0.0))
        assert torch.equal(W, W2) and torch.equal(nd, nd2) and (J is None or torch.equal(J, J2))
    taps = torch.from_numpy(hrf).cuda()
    X, Z = solver.fista_outputs(W, hrf)
    X2, Z2 = ops.fista_outputs(W, taps)
    assert torch.equal(X, X2) and torch.equal(Z, Z2)
    A = torch.from_numpy(rng.randn(64, 300)).cuda()
    assert torch.equal(solver.op_forward(A, hrf), ops.op_forward(A, taps, 300))
    assert torch.equal(solver.op_adjoint(A, hrf), ops.op_adjoint(A, taps, 300))
    t_r, dur = 0.75, 20.0
    ne = solver.hrf_normal_eq(Z[:4096], Y[:4096], 27)
    assert torch.equal(ne, ops.hrf_normal_eq(Z[:4096], Y[:4096], 27))
    th, f, tp = solver.theta_fit(ne, t_r, dur, (0.6, 1.9))
    th2, f2, tp2 = ops.theta_fit(ne.reshape(1, -1), solver._sample_times_on(ne.device, t_r, dur), 6.0, 0.001, 16.0, 0.001,
                                 0.167, 0.6, 1.9, 3)
    assert torch.equal(th, th2) and torch.equal(f, f2) and torch.equal(tp, tp2)
    # errors surface as RuntimeError with the library's message
    with pytest.raises(RuntimeError, match="pb_fista_solve"):
        torch_ops.fista_solve(Y[:4], hrf, 1.0, -1.0, 5)


@pytest.mark.gpu
def test_deconv_batch_through_torch_ops_on_the_golden_grid(golden, monkeypatch):
    """`pybold_amd.deconv` with a 2-D batch routed through torch.ops.pybold_hip (DISPATCH = "torch_ops",
    or PYBOLD_AMD_DISPATCH=torch_ops in the environment): the lambda x iterations grid of the reference
    goldens (SURVEY 8c case 3), iterate and normalised cost trace, equal to the ctypes route bit for bit."""
    import pybold_amd
    from pybold_amd import bold_signal
    g = golden("grid")
    hrf = g["hrf"]
    Y = np.stack([g["y_s%d" % s] for s in range(4)])
    for lbda in (0.1, 1.0, 10.0):
        for nit in (1, 3, 500):
            np.random.seed(0)
            ref = pybold_amd.deconv(Y, 1.0, hrf, lbda=lbda, nb_iter=nit, early_stopping=False)
            monkeypatch.setattr(bold_signal, "DISPATCH", "torch_ops")
            np.random.seed(0)
            got = pybold_amd.deconv(Y, 1.0, hrf, lbda=lbda, nb_iter=nit, early_stopping=False)
            monkeypatch.setattr(bold_signal, "DISPATCH", "ctypes")
            for a, b in zip(ref[:4], got[:4]):
                np.testing.assert_array_equal(a, b)
            gold = np.stack([g["dz_s%d_l%g_n%d" % (s, lbda, nit)] for s in range(4)])
            assert (np.linalg.norm(got[2] - gold, axis=1) / np.linalg.norm(gold, axis=1)).max() < 1e-5
            Jg = np.stack([g["J_s%d_l%g_n%d" % (s, lbda, nit)] for s in range(4)])
            np.testing.assert_allclose(got[3], Jg, rtol=3e-5)


@pytest.mark.gpu
def test_torch_ops_refuse_foreign_pointers(golden):
    """A CPU (or wrong-dtype) taps / sample-time tensor, a short cost-trace buffer: TORCH_CHECK errors,
    not a device fault."""
    from pybold_amd import solver, torch_ops
    ops = torch_ops.load()
    hrf = golden("case1")["hrf"]
    W = torch.zeros((8, 300), dtype=torch.float64, device="cuda")
    Y = torch.zeros((8, 300), dtype=torch.float32, device="cuda")
    taps_cpu = torch.from_numpy(np.ascontiguousarray(hrf))
    for bad in (taps_cpu, taps_cpu.cuda().float()):
        with pytest.raises(RuntimeError, match="taps_dev"):
            ops.fista_outputs(W, bad)
        with pytest.raises(RuntimeError, match="taps_dev"):
            ops.op_forward(W, bad, 300)
        with pytest.raises(RuntimeError, match="taps_dev"):
            ops.op_adjoint(W, bad, 300)
    nd = torch.empty((8,), dtype=torch.int32, device="cuda")
    betas = solver._betas_on(W.device, 10)
    Jshort = torch.empty((8, 5), dtype=torch.float32, device="cuda")
    with pytest.raises(RuntimeError, match="J must be"):
        ops.fista_solve(Y, W, taps_cpu, taps_cpu.cuda(), 1e-6, 1.0, None, betas, 10, Jshort, 0, 0.0, 6, nd, 1, 0)
    with pytest.raises(RuntimeError, match="taps_dev"):
        ops.fista_solve(Y, W, taps_cpu, taps_cpu, 1e-6, 1.0, None, betas, 10, None, 0, 0.0, 6, nd, 1, 0)
    ne = torch.zeros((1, 27 * 27 + 27 + 1), dtype=torch.float64, device="cuda")
    t_cpu = torch.from_numpy(solver.hrf_sample_times(0.75, 20.0))
    with pytest.raises(RuntimeError, match="sample times"):
        ops.theta_fit(ne, t_cpu, 6.0, 0.001, 16.0, 0.001, 0.167, 0.6, 1.9, 3)
    with pytest.raises(RuntimeError, match="K\\*K"):
        ops.theta_fit(ne[:, :-3].contiguous(), t_cpu.cuda(), 6.0, 0.001, 16.0, 0.001, 0.167, 0.6, 1.9, 3)
