"""Clocks and power while the config-3 solve runs back to back (development aid)."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pybold_amd import solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
Y = torch.randn(98304, 300, device="cuda")
plan = solver.FistaPlan(Y, hrf, 1.0, 1.0 / 723876.27, 500, force=None)
stop = False
def sample():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--showperflevel"], capture_output=True, text=True, timeout=20).stdout
            keep = [l for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "Power", "Temperature (Sensor junction)", "fclk"))]
            print(" | ".join(l.split(":", 1)[-1].strip()[:60] for l in keep[:8]), flush=True)
        except Exception as e:
            print("rocm-smi:", e, flush=True)
        time.sleep(1.0)
print("idle:"); 
t = threading.Thread(target=sample); t.start(); time.sleep(2.5)
print("busy:", flush=True)
t0 = time.time()
n = 0
while time.time() - t0 < 8.0:
    for _ in range(20): plan.run()
    torch.cuda.synchronize(); n += 20
dt = time.time() - t0
stop = True; t.join()
print("%.3f ms per solve (98 304 voxels), %.3f G voxel-iterations/s" % (dt / n * 1e3, 98304 * 500 * n / dt / 1e9))
