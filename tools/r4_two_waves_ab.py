"""Occupancy A/B of the matrix-pipe kernel (round 4): the NB = 5 instantiation (129 <= N <= 160) built
for ONE wave per SIMD (the shipped build: 237 VGPRs + 112 AGPRs) against the same source built with
`amdgpu_waves_per_eu(2,2)` (128 + 128 registers -- the compiler's fixed half/half split of the budget when
accumulator registers are used -- 188 B of scratch per lane).  Same problems, same iterations: ms per
500-iteration solve of 32 768 problems (two rounds at one wave per SIMD, one round at two).

usage: python tools/r4_two_waves_ab.py default pybold_amd/libpybold_hip_w2.so
"""
import os
import subprocess
import sys

code = r'''
import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from pybold_amd import data, solver
from pybold_amd.hrf_model import spm_hrf
hrf = spm_hrf(1.0, t_r=1.0, dur=30.)[0]
N = 160
from oracle import pybold_oracle as orc
A = orc.toeplitz_from_kernel(hrf, N, N).dot(np.tril(np.ones((N, N))))
step = 1.0 / (0.9 * np.linalg.norm(A, 2) ** 2)
for P in (16384, 32768, 65536):
    Y, _, _ = data.gen_rnd_bloc_bold_batch(P, dur=(N + .5) / 60., tr=1.0, hrf=hrf, nb_events=3, avg_dur=10.0, std_dur=1.0, snr=1.0, seed=1, device=torch.device("cuda"))
    plan = solver.FistaPlan(Y, hrf, 1.0, step, 500, force="mfma")
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        plan.run(); torch.cuda.synchronize()
    best = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): plan.run()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / 10)
    bad = int((plan.n_done < 0).sum())
    print("P=%d: %.3f %.3f %.3f ms  (%.2f G voxel-iterations/s; handed back %d)" % ((P,) + tuple(best) + (P * 500 / min(best) / 1e6, bad)), end=";  ")
'''
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default":
        env["PYBOLD_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("%-36s %s" % (lib, out.stdout.strip() or out.stderr.strip()[-400:]), flush=True)
