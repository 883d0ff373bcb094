"""Round 5: the reference-default call (cost trace + window rule) at 300 scans with a 42-tap HRF (TR 0.72 s, 30 s HRF): the split
form's certificate (three near tiles) against the vector dispatch.   python tools/r5_short_long_hrf.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pybold_oracle as orc  # noqa: E402
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


n, k, n_it = 300, 42, 500
hrf = orc.spm_hrf(1.0, 1.0, float(k), False)[0][:k]
lip = orc.gram_lipschitz(hrf, n)
print("# %d scans, K = %d, %d iterations, lambda = 1; ms per solve (1e9 voxel-iterations/s)" % (n, k, n_it))
for P in (10000, 50000, 100000):
    Y, _, _ = data.gen_rnd_bloc_bold_batch(P, dur=n / 60.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=0)
    for name, kw in (("plain", {}), ("cost trace + window rule, tol 1e-6", dict(want_J=True, stop="window", tol=1e-6, wind=6))):
        td = timed(lambda: solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, n_it, **kw))
        tv = timed(lambda: solver.fista_solve(Y, hrf, 1.0, 1.0 / lip, n_it, force="valu", **kw))
        print("%-7d %-36s default %8.3f ms (%.3f)   vector forms %8.3f ms (%.3f)" % (P, name, td, P * n_it / td / 1e6, tv, P * n_it / tv / 1e6), flush=True)
