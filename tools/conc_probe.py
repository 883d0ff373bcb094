"""Does running the remainder launch on a second stream overlap with the main launch?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.27
s1 = torch.cuda.Stream()

def bench(pieces, P, concurrent, reps=9):
    Y = torch.randn(P, 300, device="cuda", dtype=torch.float32)
    plans = []
    lo = 0
    for n, force in pieces:
        plans.append(solver.FistaPlan(Y[lo:lo + n], hrf, 1.0, step, 500, force=force))
        lo += n
    assert lo == P
    def run():
        s0 = torch.cuda.current_stream()
        if concurrent:
            ev = torch.cuda.Event(); ev.record(s0)
            plans[0].run()
            with torch.cuda.stream(s1):
                s1.wait_event(ev)
                for p in plans[1:]:
                    p.run()
                ev2 = torch.cuda.Event(); ev2.record(s1)
            s0.wait_event(ev2)
        else:
            for p in plans:
                p.run()
    for _ in range(3): run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3

cases = {
    10000: [[(8192, "fast2"), (1808, "wide")], [(8192, "fast2"), (1808, "fast1")]],
    12500: [[(8192, "fast2"), (4096, "fast1"), (212, "wide")], [(8192, "fast2"), (4308, "wide")], [(8192, "fast2"), (4308, "fast1")], [(12500, "fast2")]],
    12288: [[(8192, "fast2"), (4096, "fast1")]],
    16000: [[(8192, "fast2"), (4096, "fast1"), (3712, "wide")], [(16000, "fast2")]],
    25000: [[(24576, "fast2"), (424, "wide")], [(16384, "fast2"), (8192, "fast2"), (424, "wide")]],
    100000: [[(98304, "fast2"), (1696, "wide")]],
}
for P, variants in cases.items():
    for pieces in variants:
        a = bench(pieces, P, False)
        b = bench(pieces, P, True) if len(pieces) > 1 else float("nan")
        print("P=%6d %-60s sequential %7.3f ms  concurrent %7.3f ms  (%.3f G)" % (P, str(pieces), a, b, P * 500 / (min(a, b if b == b else a)) / 1e6), flush=True)
