import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import pybold_oracle as orc
from pybold_amd import solver
g = np.load("tests/golden/case1.npz"); hrf, lip = g["hrf"], float(g["lipschitz"]); rho = lip/0.9; N = 300
t = np.arange(N); rng = np.random.RandomState(0)
names = ["golden","alt","spike_end","spike0","step","noise","ramp","dc+noise"]
fams = [g["y"], np.where(t % 2 == 0, 1.0, -1.0), np.r_[np.zeros(N - 1), 1.0], np.r_[1.0, np.zeros(N - 1)], (t > 150).astype(float), rng.randn(N), t / float(N), 1e3 * (t > 150) + rng.randn(N)]
Y = np.stack(fams); Yd = torch.from_numpy(Y.astype(np.float32)).cuda(); Yo = Yd.cpu().numpy().astype(np.float64)
def rel(a,b): return np.linalg.norm(a-b,axis=1)/np.linalg.norm(b,axis=1)
for n_it in (40, 500):
  for lb in (0.0, 1.0):
    ref = orc.fista_batch(Yo, hrf, lb, 1.0/rho, n_it)
    Wm,_,nd = solver.fista_solve(Yd, hrf, lb, 1.0/rho, n_it, force="mfmaonly")
    Wv,_,_ = solver.fista_solve(Yd, hrf, lb, 1.0/rho, n_it, force="fast1")
    em, ev = rel(Wm.cpu().numpy(), ref), rel(Wv.cpu().numpy(), ref)
    X,Z = solver.fista_outputs(Wm, hrf); xr, zr = orc.fista_outputs(ref, hrf)
    ez = rel(Z.cpu().numpy(), zr); ex = rel(X.cpu().numpy(), xr)
    print("n_it %d lbda %.0f" % (n_it, lb))
    for i,nm in enumerate(names):
        print("   %-10s mfma dz %.1e z %.1e x %.1e (back %d) | fast1 dz %.1e | |w| %.2e" % (nm, em[i], ez[i], ex[i], int(nd[i] < 0), ev[i], np.linalg.norm(ref[i])))
