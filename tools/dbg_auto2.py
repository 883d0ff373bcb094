import sys
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
import pybold_amd
from oracle import c_oracle, pybold_oracle as orc
g = np.load("tests/golden/grid.npz")
hrf, lip = g["hrf"], float(g["lip_s0"])
Y = np.stack([g["y_s%d" % (s % 4)] * (1.0 + 0.05 * (s // 4)) for s in range(16)])
step = 1.0 / lip
v = 8
y = Y[v]
sig = orc.mad_daub_noise_est(y)
# replicate the outer loop, comparing each inner solve with the oracle's
H = orc._MatrixFreeH(hrf)
Hty = H.adj(y)
w_o = np.zeros(300); alpha_o = 1.0
Yd = torch.from_numpy(y[None].copy()).cuda()            # float64 -> generic f64 kernel
W = torch.zeros((1, 300), dtype=torch.float64, device="cuda"); alpha_g = 1.0
for i in range(12):
    lb_o, lb_g = 1 / (2 * alpha_o), 1 / (2 * alpha_g)
    # oracle inner solve, counting iterations
    cnt = [0]
    w_o = orc._inner_fista(w_o, H, Hty, step, lb_o / lip, 1000, True, 6, 1e-6)
    Wn, _, nd = solver.fista_solve(Yd, hrf, np.array([lb_g]), step, 1000, W0=W, stop="window", tol=1e-6, wind=6)
    W = Wn
    wg = W.cpu().numpy()[0]
    r_o = np.sum((orc.causal_conv(hrf, np.cumsum(w_o)) - y) ** 2); r_g = np.sum((orc.causal_conv(hrf, np.cumsum(wg)) - y) ** 2)
    alpha_o += 1e-4 * (r_o - 300 * sig ** 2); alpha_g += 1e-4 * (r_g - 300 * sig ** 2)
    print(i, "n_done gpu", int(nd[0]), "rel err w %.2e" % (np.linalg.norm(wg - w_o) / np.linalg.norm(w_o)), "lbda", lb_o, lb_g)
