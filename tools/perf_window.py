"""Throughput of the stop-rule paths (voxel-iterations/s), realistic synthetic series.
window+J = the reference-default deconv call (bold_signal.py:13-14).  `lib` = library dispatch
(tol = 1e-6: no-fire certificate on the pair form + re-solve of what it cannot clear),
`full` = the rule evaluated in full on the single-row form (round-2 path)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import data, solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
for V in (10000, 100000):
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=5.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0,
                                           snr=1.0, seed=1, device=torch.device("cuda"))
    for name, kw in (("plain", dict()), ("plain+J", dict(want_J=True)),
                     ("loops", dict(stop="loops", tol=0.0)),
                     ("window lib", dict(stop="window", tol=1e-6, wind=6)),
                     ("window+J lib", dict(stop="window", tol=1e-6, wind=6, want_J=True)),
                     ("window+J lib tol=1e-4", dict(stop="window", tol=1e-4, wind=6, want_J=True)),
                     ("window full", dict(stop="window", tol=1e-6, wind=6, force="nocert")),
                     ("window+J full", dict(stop="window", tol=1e-6, wind=6, want_J=True, force="nocert")),
                     ("generic window", dict(force="generic", stop="window", tol=0.0, wind=6))):
        nit = 100 if "generic" in name else 500
        best = 1e9
        solver.fista_solve(Y, hrf, 1.0, step, nit, **kw); torch.cuda.synchronize()
        for _ in range(3):
            t0 = time.perf_counter(); _, _, nd = solver.fista_solve(Y, hrf, 1.0, step, nit, **kw); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%-24s V=%6d it=%d: %8.2f ms  %.3e voxel-iter/s  (n_done min %d)" %
              (name, V, nit, best * 1e3, V * nit / best, int(nd.min())), flush=True)
