"""A/B timing of alternative builds of libpybold_hip.so (PYBOLD_HIP_LIB), interleaved
rounds in separate processes are avoided: each build is timed in its own process but
the script reports min and median over several launches of 100k x 300 x 500."""
import os, subprocess, sys
libs = sys.argv[1:]
code = r'''
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
torch.manual_seed(0)
Y = torch.randn(100000, 300, device="cuda", dtype=torch.float32)
import os
plan = solver.FistaPlan(Y, hrf, 1.0, 1.0 / 723876.27, 500, force=os.environ.get("PYBOLD_AB_FORCE", "fast1"))
for _ in range(3): plan.run()
torch.cuda.synchronize()
ts = []
for _ in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    plan.W.zero_(); e0.record(); plan.launch(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("min %.3f ms  median %.3f ms  checksum %.10e" % (min(ts), float(np.median(ts)), float(plan.W.abs().sum())))
'''
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != "default":
            env["PYBOLD_HIP_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("%-28s round %d: %s" % (lib, rnd, out.stdout.strip() or out.stderr.strip()[-300:]), flush=True)
