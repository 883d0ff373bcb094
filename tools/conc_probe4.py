"""12 500 problems: the 212 left-overs first (own high-priority stream) instead of last?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
P = 12500
Y = torch.randn(P, 300, device="cuda", dtype=torch.float32)
pa = solver.FistaPlan(Y[:8192], hrf, 1.0, step, 500, force="fast2")
pb = solver.FistaPlan(Y[8192:12288], hrf, 1.0, step, 500, force="fast1")
pc = solver.FistaPlan(Y[12288:], hrf, 1.0, step, 500, force="wide")
auto = solver.FistaPlan(Y, hrf, 1.0, step, 500, force=None)
s1 = torch.cuda.Stream(priority=-1)
s2 = torch.cuda.Stream()

def three(order):
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    for k in order:
        if k == "c":
            with torch.cuda.stream(s1): pc.run()
        elif k == "b":
            with torch.cuda.stream(s2): pb.run()
        else:
            pa.run()
    cur.wait_stream(s1); cur.wait_stream(s2)

def clock(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for _ in range(2):
    print("library dispatch      %.3f ms" % clock(auto.run))
    for order in ("cab", "cba", "acb", "abc"):
        print("three streams %s     %.3f ms" % (order, clock(lambda: three(order))), flush=True)
