"""Randomised parity sweep over call shapes, wider than the test-suite (development aid):
every kernel form that accepts the shape against the C float64 oracle -- iterate and cost trace.
usage: python tools/parity_sweep.py [n_cases] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle, pybold_oracle as orc
from pybold_amd import solver

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {}
t0 = time.time()
for case in range(n_cases):
    N = int(rng.choice([rng.randint(1, 305), rng.randint(129, 311), rng.randint(1, 700), rng.randint(700, 2433)], p=[0.3, 0.4, 0.2, 0.1]))
    K = int(min(rng.randint(1, 49), max(1, N)))
    V = int(rng.randint(1, 120))
    y_rep = int(rng.choice([1, 1, 1, 3, 20]))
    n_iter = int(rng.randint(0, 120))
    hrf = rng.randn(K) * 0.3
    if rng.rand() < 0.3:
        hrf[0] = 0.0                                              # the SKIP0 builds
    Y = rng.randn(V, N)
    Y32 = Y.astype(np.float32).astype(np.float64)
    P = V * y_rep
    lam = (rng.choice([0.0, 1e-3, 0.05, 1.0, 30.0]) * (0.5 + rng.rand(P))) if rng.rand() < 0.5 else float(rng.choice([0.0, 0.05, 1.0]))
    W0 = 0.01 * rng.randn(P, N) if rng.rand() < 0.4 else None
    want_J = bool(rng.rand() < 0.5)
    A = orc.toeplitz_from_kernel(hrf, N, N).dot(np.tril(np.ones((N, N)))) if N <= 400 else None
    if A is not None:
        lip = 1.05 * np.linalg.norm(A, 2) ** 2 + 1e-9
    else:                                                          # cheap upper bound: (sum|h| * N)^2
        lip = (np.abs(hrf).sum() * N) ** 2 + 1e-9
    step = 1.0 / lip
    Yrep = np.repeat(Y32, y_rep, axis=0)
    ref, Jref, _ = c_oracle.fista_batch(Yrep, hrf, lam, step, n_iter, W0=W0, want_J=want_J, threads=8)
    scale = np.abs(ref).max() + 1e-30
    Yd = torch.from_numpy(Y.astype(np.float32)).cuda()
    W0d = torch.from_numpy(W0).cuda() if W0 is not None else None
    forms = ["generic"] if 3 * N + K + 8 <= 20000 else []
    if solver.has_fast_path(N, K):
        forms += [None, "fast1", "fast2", "fast2d", "seq", "one"]
    if N <= 2432 and K <= 48:
        forms += ["wide"]
    if 128 < N <= 310 and solver.has_fast_path(N, K):              # round 3: the matrix-pipe form (guards + re-solve
        forms += ["mfma", "valu"]                                  # included), and the dispatch without it
    for force in forms:
        try:
            W, J, nd = solver.fista_solve(Yd, hrf, lam, step, n_iter, W0=W0d, want_J=want_J, y_rep=y_rep, force=force)
        except Exception as e:                                     # a form that does not take this shape says so
            if force in ("wide", "fast2", "fast2d") and "no " in str(e).lower() or "rejected" in str(e).lower() or "wide" in str(e).lower():
                continue
            raise
        err = float(np.abs(W.cpu().numpy() - ref).max() / scale)
        assert (nd.cpu().numpy() == n_iter).all()
        jerr = 0.0
        if want_J and n_iter > 0:
            Jg = J.cpu().numpy().astype(np.float64)
            jerr = float(np.abs(Jg - Jref).max() / (np.abs(Jref).max() + 1e-30))
        w = worst.setdefault(str(force), [0.0, 0.0, 0, None])
        w[2] += 1
        if err > w[0]:
            w[0], w[3] = err, (V, N, K, y_rep, n_iter)
        w[1] = max(w[1], jerr)
        if err > 1e-5 or jerr > 1e-4:
            print("FAIL", force, (V, N, K, y_rep, n_iter, want_J), err, jerr, flush=True)
    if case % 50 == 49:
        print("... %d cases, %.0f s" % (case + 1, time.time() - t0), flush=True)
print("%-8s %6s %12s %12s   worst case (V, N, K, y_rep, n_iter)" % ("form", "calls", "max err W", "max err J"))
for k, (e, j, n, c) in sorted(worst.items()):
    print("%-8s %6d %12.3e %12.3e   %s" % (k, n, e, j, c))
