"""H2D / solve / D2H overlap of solver.HostPipeline by chunk size (development aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
V, N, it = 100000, 300, 500
Y = torch.randn(V, N, device="cuda")
Yh = Y.cpu().pin_memory()

def clock(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

plan = solver.FistaPlan(Y, hrf, 1.0, step, it, force=None)
print("resident solve           %.3f ms" % clock(plan.run))
Yd = torch.empty_like(Y)
print("H2D alone (120 MB)       %.3f ms" % clock(lambda: Yd.copy_(Yh, non_blocking=True)))
Wh = torch.empty((V, N), dtype=torch.float32).pin_memory()
W32 = plan.W.float()
print("D2H alone (120 MB f32)   %.3f ms" % clock(lambda: Wh.copy_(W32, non_blocking=True)))
print("f64->f32 on device       %.3f ms" % clock(lambda: W32.copy_(plan.W)))
for chunk in (16384, 32768, 49152, 65536, 100000):
    for od in (None, torch.float32):
        p = solver.HostPipeline(V, N, hrf, 1.0, step, it, chunk=chunk, out_dtype=od)
        print("chunk %6d out=%-8s %.3f ms" % (chunk, "HBM" if od is None else "host f32", clock(lambda: p.run(Yh))), flush=True)
        del p

# where does the exposed time go?  the same choreography with pieces removed
class NoCopy(solver.HostPipeline):
    pass
p = solver.HostPipeline(V, N, hrf, 1.0, step, it, chunk=16384, out_dtype=None)
def only_plans():
    for pl in p.plans: pl.run()
print("7 chunk plans back to back on one stream   %.3f ms" % clock(only_plans))
def plans_on_side():
    cur = torch.cuda.current_stream()
    p.s_cmp.wait_stream(cur)
    with torch.cuda.stream(p.s_cmp):
        for pl in p.plans: pl.run()
    cur.wait_stream(p.s_cmp)
print("... on a side stream                        %.3f ms" % clock(plans_on_side))
def copies_only():
    cur = torch.cuda.current_stream()
    p.s_in.wait_stream(cur)
    with torch.cuda.stream(p.s_in):
        for c, (lo, hi) in enumerate(p.bounds):
            p.Yd[c % 2][:hi - lo].copy_(Yh[lo:hi], non_blocking=True)
    cur.wait_stream(p.s_in)
print("7 chunk H2D copies on the copy stream       %.3f ms" % clock(copies_only))
def first_copy_then_plans():
    cur = torch.cuda.current_stream()
    p.s_in.wait_stream(cur); p.s_cmp.wait_stream(cur)
    with torch.cuda.stream(p.s_in):
        p.Yd[0].copy_(Yh[:16384], non_blocking=True); p.ev_in[0].record(p.s_in)
    with torch.cuda.stream(p.s_cmp):
        p.s_cmp.wait_event(p.ev_in[0])
        for pl in p.plans: pl.run()
    cur.wait_stream(p.s_cmp)
print("one chunk copy, then all plans               %.3f ms" % clock(first_copy_then_plans))
