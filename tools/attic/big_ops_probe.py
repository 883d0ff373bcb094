import sys
sys.path.insert(0, ".")
import numpy as np, torch
from pybold_amd import solver
import pybold_amd
from oracle import pybold_oracle as orc
rng = np.random.RandomState(0)
for n, K in ((6000, 600), (3000, 600), (9000, 30)):
    k = rng.randn(K) * 0.1; x = rng.randn(n)
    try:
        H = pybold_amd.ConvAndLinear(pybold_amd.DiscretInteg(), k, n, n)
        a = H.op(x); b = H.adj(x)
        Ho = orc._MatrixFreeH(k)
        print(n, K, "op err", np.abs(a - Ho.op(x)).max() / np.abs(a).max(), "adj err", np.abs(b - Ho.adj(x)).max() / np.abs(b).max())
    except Exception as e:
        print(n, K, "FAILED:", e)
for n, K in ((2000, 40), (5000, 100)):
    k = rng.randn(K) * 0.1; Y = rng.randn(3, n)
    try:
        W, J, _ = solver.fista_solve(torch.from_numpy(Y.astype(np.float32)).cuda(), k, 0.1, 1e-7, 20, want_J=True, stop="window", tol=0.0, wind=6)
        Wo = orc.fista_batch(Y.astype(np.float32).astype(np.float64), k, 0.1, 1e-7, 20)
        print(n, K, "fista generic err", np.abs(W.cpu().numpy() - Wo).max() / np.abs(Wo).max())
    except Exception as e:
        print(n, K, "FAILED:", e)
