"""A/B of library builds on the matrix-pipe kernel (one process per build): ms per 500-iteration solve."""
import os, subprocess, sys
code = r'''
import sys, time, torch
sys.path.insert(0, ".")
from pybold_amd import data, solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
step = 1.0 / 723876.2744579345
Y, _, _ = data.gen_rnd_bloc_bold_batch(98304, dur=5.0, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0, std_dur=1.0, snr=1.0, seed=1, device=torch.device("cuda"))
plan = solver.FistaPlan(Y, hrf, 1.0, step, 500, force="mfma")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    plan.run(); torch.cuda.synchronize()
best = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): plan.run()
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 10)
print("%.3f %.3f %.3f" % tuple(best))
'''
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default":
        env["PYBOLD_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("%-32s %s" % (lib, out.stdout.strip() or out.stderr.strip()[-300:]), flush=True)
