"""Window stop rule (pybold/bold_signal.py:82-95) at random tolerances: iterations executed by
the batch kernels (float32 FIRs, float32 increment ring) and by the float64 kernel against the
float64 oracle, voxel by voxel (development aid).  usage: python tools/stop_sweep.py [voxels] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pybold_oracle as orc
from pybold_amd import solver, data
from pybold_amd.hrf_model import spm_hrf

V = int(sys.argv[1]) if len(sys.argv) > 1 else 96
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.RandomState(seed)
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
lip = 723876.2744579345
Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=5, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=seed)
Yh = Y.cpu().numpy().astype(np.float64)
print("%8s %6s | %-28s | %-28s | %-28s" % ("tol", "lbda", "fast1 (fp32 FIRs)", "wide (fp32 FIRs)", "float64 kernel"))
for tol in (3e-2, 1e-2, 3e-3, 1e-3, 3e-4):
    for lbda in (0.3, 1.0, 3.0):
        ref_n = np.empty(V, dtype=int)
        ref_w = np.empty((V, 300))
        for v in range(V):
            _, _, w, _, n, _ = orc.deconv_fixed_lbda(Yh[v], hrf, lbda, nb_iter=600, early_stopping=True, tol=tol,
                                                      lipschitz=lip, dense=False)
            ref_n[v], ref_w[v] = n, w
        cols = []
        for force, Yin in (("fast1", Y), ("wide", Y), (None, Y.double())):
            W, _, nd = solver.fista_solve(Yin, hrf, lbda, 1.0 / lip, 600, stop="window", tol=tol, force=force)
            nd = nd.cpu().numpy()
            same = int((nd == ref_n).sum())
            off = np.abs(nd - ref_n).max()
            ok = nd == ref_n
            err = float((np.abs(W.cpu().numpy()[ok] - ref_w[ok]).max(axis=1) / (np.abs(ref_w[ok]).max(axis=1) + 1e-30)).max()) if ok.any() else float("nan")
            cols.append("%3d/%d same, max off %2d, %.1e" % (same, V, off, err))
        print("%8.0e %6.1f | %-28s | %-28s | %-28s   (stop its %d..%d)" % (tol, lbda, cols[0], cols[1], cols[2], ref_n.min(), ref_n.max()), flush=True)
