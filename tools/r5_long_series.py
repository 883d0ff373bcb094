"""Round 5: series of 641 .. 1 280 scans -- the four-wave matrix-pipe form (fista_mfma4.h) against the one-problem-per-wave
vector form it replaces, by batch size.  N = 1 200, K = 28 (VERDICT r4 item 6b asks >= 1.0e9 voxel-iterations/s).

    python tools/r5_long_series.py [scans [taps]] > profiles/r5_long_series_1200_scans.txt
"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 28
    n_it = 500
    hrf = orc.spm_hrf(1.0, 1.0, float(k), False)[0][:k]
    lip = orc.gram_lipschitz(hrf, n)
    print("# %d scans, K = %d, %d iterations, lambda = 1 (block signals, SNR 1 dB); ms per solve and 1e9 voxel-iterations/s" % (n, len(hrf), n_it))
    print("%-8s %22s %22s %22s   %s" % ("problems", "default", "four waves (forced)", "vector form (valu)", "handed back"))
    sizes = (512, 1024, 1536, 2048, 3072, 4096, 5120, 6144, 8192, 12288, 16384, 20000, 32768) if len(sys.argv) <= 2 else (4096, 8192, 16384, 32768)
    for P in sizes:
        Y, _, _ = data.gen_rnd_bloc_bold_batch(P, dur=n / 60.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=0)
        Y = Y[:, :n].contiguous()
        row = []
        for force in (None, "mfma2", "valu"):
            t = timed(lambda: solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, n_it, force=force))
            row.append("%8.3f ms %6.3f" % (t, P * n_it / t / 1e6))
        _, _, nd = solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, n_it, force="mfma2only")
        print("%-8d %22s %22s %22s   %.2f %%" % (P, row[0], row[1], row[2], 100.0 * float((nd < 0).float().mean())), flush=True)
    shapes = ((dict(want_J=True), "cost trace"), (dict(want_J=True, stop="window", tol=1e-6, wind=6), "cost trace + window rule, tol 1e-6"),
              (dict(stop="loops", tol=1e-6), "_loops_deconv rule, tol 1e-6"))
    for kw, name in shapes:
        P = 16384
        Y, _, _ = data.gen_rnd_bloc_bold_batch(P, dur=n / 60.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=0)
        Y = Y[:, :n].contiguous()
        ts = [timed(lambda: solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, n_it, force=f, **kw)) for f in (None, "valu")]
        print("%s, %d problems: default %.3f ms (%.3f), vector form %.3f ms (%.3f)" % (name, P, ts[0], P * n_it / ts[0] / 1e6, ts[1], P * n_it / ts[1] / 1e6))


if __name__ == "__main__":
    main()
