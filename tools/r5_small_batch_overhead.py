import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import pybold_oracle as orc
from pybold_amd import data, solver
hrf = orc.spm_hrf(1.0, 1.0, 30.0)[0]
def timed(fn, reps=7, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))
step = 1.0/723876.2744579345
for V in (5000, 10000, 12500, 20000, 25000):
    Y,_,_ = data.gen_rnd_bloc_bold_batch(V, dur=5, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=0)
    p1 = solver.FistaPlan(Y, hrf, 1.0, step, 500, force=None)
    p2 = solver.FistaPlan(Y, hrf, 1.0, step, 500, force="nopart")
    a = timed(lambda: p1.launch(cold=True)); b = timed(lambda: p2.launch(cold=True)); a2 = timed(lambda: p1.launch(cold=True))
    print("%6d voxels: default %.3f / %.3f ms   nopart %.3f ms   (+%.0f us)" % (V, a, a2, b, 1000*(min(a,a2)-b)))
