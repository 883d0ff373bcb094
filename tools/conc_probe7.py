"""4 096 < P <= 6 144: one single-row wave per SIMD with one-per-wave left-overs beside it, against the pair form alone?"""
import os, sys, time
import torch
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
for P in (4500, 5000, 5500, 6000, 6144, 6500, 7000):
    Y = torch.randn(P, 300, device="cuda")
    pa = solver.FistaPlan(Y[:4096], hrf, 1.0, step, 500, force="fast1")
    pc = solver.FistaPlan(Y[4096:], hrf, 1.0, step, 500, force="wide")
    auto = solver.FistaPlan(Y, hrf, 1.0, step, 500, force=None)
    def two(first):
        cur = torch.cuda.current_stream()
        s2.wait_stream(cur)
        if first == "a": pa.run()
        with torch.cuda.stream(s2): pc.run()
        if first != "a": pa.run()
        cur.wait_stream(s2)
    print("P=%d library %.3f ms | single-row(4096) || one-per-wave(%d): %.3f ms (left-overs launched first: %.3f ms)"
          % (P, clock(auto.run), P - 4096, clock(lambda: two("a")), clock(lambda: two("c"))), flush=True)
