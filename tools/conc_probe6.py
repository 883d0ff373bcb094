"""Does the side stream still overlap after the application has created other streams?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
Y = torch.randn(12500, 300, device="cuda")
mode = sys.argv[1]
keep = []
if mode == "streams_before":
    keep = [torch.cuda.Stream() for _ in range(6)]
    for s in keep:
        with torch.cuda.stream(s):
            torch.zeros(8, device="cuda").add_(1)
    torch.cuda.synchronize()
def clock(plan, reps=20):
    for _ in range(5): plan.run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): plan.run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
if mode == "pipeline_first":
    Yh = torch.randn(100000, 300).pin_memory()
    p = solver.HostPipeline(100000, 300, hrf, 1.0, step, 500, out_dtype=torch.float32)
    p.run(Yh); p.run(Yh)
    del p, Yh
if mode == "bigsolve_first":
    Yb = torch.randn(100000, 300, device="cuda")
    pb_ = solver.FistaPlan(Yb, hrf, 1.0, step, 500, force=None)
    for _ in range(10): pb_.run()
    torch.cuda.synchronize()
auto = solver.FistaPlan(Y, hrf, 1.0, step, 500, force=None)
seq = solver.FistaPlan(Y, hrf, 1.0, step, 500, force="seq")
print(mode, "first: auto %.3f ms, seq %.3f ms" % (clock(auto), clock(seq)), flush=True)
if mode == "streams_after":
    keep = [torch.cuda.Stream() for _ in range(6)]
    for s in keep:
        with torch.cuda.stream(s):
            torch.zeros(8, device="cuda").add_(1)
    torch.cuda.synchronize()
    print(mode, "after 6 more streams: auto %.3f ms, seq %.3f ms" % (clock(auto), clock(seq)), flush=True)
if mode == "pipeline_between":
    Yh = torch.randn(100000, 300).pin_memory()
    p = solver.HostPipeline(100000, 300, hrf, 1.0, step, 500, out_dtype=torch.float32)
    p.run(Yh); p.run(Yh)
    print(mode, "after a HostPipeline run: auto %.3f ms, seq %.3f ms" % (clock(auto), clock(seq)), flush=True)
    del p, Yh
    torch.cuda.empty_cache()
    print(mode, "after deleting it: auto %.3f ms, seq %.3f ms" % (clock(auto), clock(seq)), flush=True)
