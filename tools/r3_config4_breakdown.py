"""Config 4 (shared-HRF blind loop) split per outer iteration: z-step, normal equations + message,
theta step -- HIP events around each part of the loop `distributed.bd_shared` runs.
Usage (GPU): python tools/r3_config4_breakdown.py [voxels] [force]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pybold_amd import data, distributed, solver  # noqa: E402
from pybold_amd.hrf_model import spm_hrf  # noqa: E402


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    force = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device("cuda:0")
    t_r, hrf_dur, lbda, nb_outer, nb_inner, N = 0.75, 20.0, 1.7, 20, 100, 300
    h_true = spm_hrf(0.7, t_r, hrf_dur, False)[0]
    K = len(h_true)
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=N * t_r / 60.0, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0,
                                           std_dur=1.0, snr=10.0, seed=4000, device=dev)
    ops = distributed.HipOps(t_r, hrf_dur, N)
    rows = []
    for rep in range(2):                      # second pass is the one reported (first warms everything up)
        theta = torch.full((1,), 2.0, dtype=torch.float64, device=dev)
        taps = ops.hrf(theta)
        W = torch.zeros((V, N), dtype=torch.float64, device=dev)
        msg = torch.empty((K * K + K + 2,), dtype=torch.float64, device=dev)
        step = None
        rows = []
        for it in range(nb_outer + 1):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            if step is None:
                step = 1.0 / solver.gram_frobenius_batch(taps.reshape(1, -1), N)
            W, n_done = solver.fista_solve_pp(Y, taps, step, lbda, nb_inner, W0=W, inplace=True,
                                              force=force if force else (None if it == nb_outer else "intermediate"))
            ev[1].record()
            ops.normal_eq_msg(W, Y, K, msg)
            ev[2].record()
            if it < nb_outer:
                theta, f, taps, step, jc = ops.theta_step(msg, (0.3, 1.9), lbda)
            ev[3].record()
            torch.cuda.synchronize()
            rows.append({"outer": it, "z_step_ms": ev[0].elapsed_time(ev[1]), "normal_eq_ms": ev[1].elapsed_time(ev[2]),
                         "theta_step_ms": ev[2].elapsed_time(ev[3]), "theta": float(theta),
                         "max_abs_w": float(W.abs().max()), "handed_back": int((n_done < 0).sum())})
    tot = {k: sum(r[k] for r in rows) for k in ("z_step_ms", "normal_eq_ms", "theta_step_ms")}
    print(json.dumps({"voxels": V, "force": force, "totals_ms": tot, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
