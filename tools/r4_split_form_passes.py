"""ms per 500-iteration plain solve by batch size and kernel form (round 4): the split matrix-pipe form
(`fista_mfma2_kernel`: every series over two waves / two SIMDs) against the one-wave matrix-pipe form, the vector
plan of round 2 and the library's dispatch.  Calibrates plan_pieces_mfma (capi.hip) and records the long-series rates.

usage: python tools/r4_split_form_passes.py > profiles/r4_split_form_passes.txt
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pybold_oracle as orc          # noqa: E402
from pybold_amd import data, solver              # noqa: E402

dev = torch.device("cuda")


def ms(plan, reps=10):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        plan.run()
        torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps)
    return min(best)


def table(N, K, sizes, forms, n_iter=500):
    hrf = orc.spm_hrf(1.0, 1.0, float(K), False)[0][:K]
    step = 1.0 / (0.9 * orc.gram_lipschitz(hrf, N))
    Y, _, _ = data.gen_rnd_bloc_bold_batch(max(sizes), dur=(N + .5) / 60., tr=1.0, hrf=hrf, nb_events=5 if N >= 250 else 2,
                                           avg_dur=10.0, std_dur=1.0, snr=1.0, seed=1, device=dev)
    print("\n## N = %d, K = %d, %d iterations: ms per solve (G voxel-iterations/s)" % (N, K, n_iter))
    print("%8s " % "problems" + " ".join("%22s" % f for f in forms) + "   plan of the dispatch")
    for P in sizes:
        row = []
        for f in forms:
            try:
                plan = solver.FistaPlan(Y[:P], hrf, 1.0, step, n_iter, force=None if f == "dispatch" else f)
                t = ms(plan)
                bad = int((plan.n_done != n_iter).sum())
                row.append("%9.3f (%5.2f)%s" % (t, P * n_iter / t / 1e6, "" if bad == 0 else " !%d" % bad))
            except Exception as e:
                row.append("n/a")
        n_main, main, tail = solver.launch_plan(N, K, P)
        print("%8d " % P + " ".join("%22s" % r for r in row) + "   %s%s" % (
            ("%d %s + " % (n_main, main.split(" ")[0])) if n_main else "", tail.split(" ")[0]), flush=True)


table(300, 30, [1024, 2048, 3072, 4096, 5000, 6250, 8192, 9000, 10000, 12500, 16384, 20000, 25000, 50000, 100000],
      ["dispatch", "mfma", "mfma2", "valu"])
table(600, 30, [1024, 2048, 4096, 8192, 10000, 25000, 50000], ["dispatch", "mfma2", "valu"])
table(400, 27, [8192, 50000], ["dispatch", "mfma2", "valu"])
table(640, 33, [8192, 50000], ["dispatch", "mfma2", "valu"])
table(240, 27, [4096, 8192, 10000], ["dispatch", "mfma", "mfma2", "valu"])
