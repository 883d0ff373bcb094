"""Round 5: the library's default dispatch against its own two extremes along lambda / lambda_max.

100 000 voxels x 300 scans x 500 iterations (BASELINE config 3's batch), per-voxel lambda = c * lambda_max,v for
c in {0.01, 0.05, 0.13, 0.2, 0.3, 0.6, 1} (every problem of a point in the same class), then ONE scalar lambda for the
batch placed at the batch's quantiles of 0.13 * lambda_max (a MIXED batch).  Per point, ms per solve of
  default   pb_fista_solve_ex: partition on the device, dense class on the matrix pipe, sparse class on the vector forms
  nopart    round 4's dispatch (PB_FLAG_NO_PARTITION): everything on the matrix pipe, then the handed-back problems
  valu      the vector forms only (PB_FLAG_NO_MFMA)
for the plain solve and for the reference-default call shape (cost trace + window rule, tol 1e-6).
Asked: default <= 1.05 x min(nopart, valu) at every point.

    python tools/r5_lambda_sweep.py [--voxels 100000] > profiles/r5_lambda_sweep.txt
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pybold_oracle as orc  # noqa: E402  (HRF and step only)
from pybold_amd import data, solver  # noqa: E402


def timed(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=100000)
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--scans", type=int, default=300)
    ap.add_argument("--shapes", default="plain,default")
    ap.add_argument("--ratio", type=float, default=0.0, help="dense_ratio of the partition (0: the library's default)")
    args = ap.parse_args()
    V, n_it = args.voxels, args.iters
    hrf = orc.spm_hrf(1.0, 1.0, 30.0)[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=args.scans / 60.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=0)
    N = Y.shape[1]
    lip = 0.9 * orc.spectral_radius_est(orc._MatrixFreeH(hrf), np.random.RandomState(0).randn(N))
    step = 1.0 / lip
    lmax = solver.lambda_max(Y, hrf)
    print("# %d voxels x %d scans, K = %d, %d iterations; lambda_max: median %.3g, 5%% %.3g, 95%% %.3g; dense_ratio %s"
          % (V, N, len(hrf), n_it, float(lmax.median()), float(lmax.quantile(0.05)), float(lmax.quantile(0.95)), args.ratio or "default"))
    worst = 0.0
    shapes = {"plain": ("plain solve", dict()),
              "default": ("cost trace + window rule, tol 1e-6 (the reference-default call)", dict(want_J=True, stop="window", tol=1e-6, wind=6)),
              "loops": ("_loops_deconv rule, tol 1e-4", dict(stop="loops", tol=1e-4))}
    for key in args.shapes.split(","):
        shape, kw = shapes[key]
        print("\n## %s\n%-36s %9s %9s %9s %8s   %s" % (shape, "point", "default", "nopart", "valu", "ratio", "handed back by nopart's matrix-pipe pass"))
        points = [("lambda = %.2f lambda_max,v" % c, lmax * c) for c in (0.01, 0.05, 0.13, 0.2, 0.3, 0.6, 1.0)]
        points += [("scalar: q%02d of 0.13 lambda_max,v" % int(100 * q), float((0.13 * lmax).quantile(q))) for q in (0.1, 0.5, 0.9)]
        points += [("scalar: lambda = 1 (config 3)", 1.0)]
        for name, lam in points:
            t = {}
            for force in (None, "nopart", "valu"):
                t[force] = timed(lambda: solver.fista_solve(Y, hrf, lam, step, n_it, force=force, dense_ratio=args.ratio, **kw))
            hb = ""
            if not kw:
                _, _, nd = solver.fista_solve(Y, hrf, lam, step, n_it, force="mfmaonly")
                hb = "%.1f %%" % (100.0 * float((nd < 0).float().mean()))
            ratio = t[None] / min(t["nopart"], t["valu"])
            worst = max(worst, ratio)
            print("%-36s %9.2f %9.2f %9.2f %8.3f   %s" % (name, t[None], t["nopart"], t["valu"], ratio, hb), flush=True)
    print("\n# worst default / min(nopart, valu): %.3f" % worst)


if __name__ == "__main__":
    main()
