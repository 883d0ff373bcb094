import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import data, blind, spm_hrf
t_r, dur = 0.75, 20.0
h_true = spm_hrf(0.7, t_r, dur, False)[0]
for V, nb in ((1000, 20), (10000, 20), (50000, 20)):
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=3.75, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=10.0, seed=0)
    blind.bd_batch(Y[:64], t_r, lbda=1.7, hrf_dur=dur, nb_iter=2); torch.cuda.synchronize()
    t0 = time.perf_counter()
    X, Z, W, H, d = blind.bd_batch(Y, t_r, lbda=1.7, hrf_dur=dur, nb_iter=nb)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    th = d["theta"]
    print("bd_batch V=%d nb_iter=%d (inner %d each): %.3f s; theta median %.3f [%.3f, %.3f] (generated 0.70); J median %.4f"
          % (V, nb, nb, dt, np.median(th), np.percentile(th, 10), np.percentile(th, 90), np.median(d["J"][-1])), flush=True)
