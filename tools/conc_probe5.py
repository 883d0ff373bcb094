"""12 500 problems: left-overs BEFORE the single-row waves on the side stream?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
s2 = torch.cuda.Stream()

def clock(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for P in (12500, 12800, 13312):
    Y = torch.randn(P, 300, device="cuda", dtype=torch.float32)
    pa = solver.FistaPlan(Y[:8192], hrf, 1.0, step, 500, force="fast2")
    pb = solver.FistaPlan(Y[8192:12288], hrf, 1.0, step, 500, force="fast1")
    pc = solver.FistaPlan(Y[12288:], hrf, 1.0, step, 500, force="wide")
    auto = solver.FistaPlan(Y, hrf, 1.0, step, 500, force=None)
    def two(order):
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        pa.run()
        with torch.cuda.stream(s2):
            for k in order:
                (pb if k == "b" else pc).run()
        cur.wait_stream(s2)
    for _ in range(2):
        print("P=%d library %.3f ms | side stream: single-row then left-overs %.3f ms | left-overs then single-row %.3f ms"
              % (P, clock(auto.run), clock(lambda: two("bc")), clock(lambda: two("cb"))), flush=True)
