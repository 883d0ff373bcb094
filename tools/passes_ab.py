"""A/B of single-pass times: one-wave and split matrix-pipe forms at a few shapes (development aid)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pybold_oracle as orc
from pybold_amd import data, solver
dev = torch.device("cuda")
def ms(plan, reps=10):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        plan.run(); torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): plan.run()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps)
    return min(best)
for N, K, P, f in [(300, 30, 16384, "mfmaonly"), (300, 30, 8192, "mfma2only"), (280, 30, 8192, "mfma2only"), (310, 30, 8192, "mfma2only"),
                   (558, 30, 8192, "mfma2only"), (589, 30, 8192, "mfma2only"), (600, 30, 8192, "mfma2only"), (620, 30, 8192, "mfma2only"), (400, 27, 8192, "mfma2only")]:
    hrf = orc.spm_hrf(1.0, 1.0, float(K), False)[0][:K]
    step = 1.0 / (0.9 * orc.gram_lipschitz(hrf, N))
    Y, _, _ = data.gen_rnd_bloc_bold_batch(P, dur=(N + .5) / 60., tr=1.0, hrf=hrf, nb_events=5, avg_dur=10.0, std_dur=1.0, snr=1.0, seed=1, device=dev)
    try:
        plan = solver.FistaPlan(Y, hrf, 1.0, step, 500, force=f)
        t = ms(plan)
        print("N=%d K=%d P=%d %s: %.3f ms  (%.2f G/s) bad=%d" % (N, K, P, f, t, P * 500 / t / 1e6, int((plan.n_done != 500).sum())), flush=True)
    except Exception as e:
        print("N=%d %s: n/a %s" % (N, f, e))
