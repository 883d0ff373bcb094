#!/usr/bin/env python3
"""Benchmark of the deconvolution hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* = one full solve of the per-GPU voxel batch: 500 iterations of the
reference's FISTA-like recurrence (pybold/bold_signal.py:62-72) on V voxels x
300 scans, L1/TV-regularised, fixed canonical HRF -- BASELINE.json config 3's
problem (100k voxels x 300 scans x 500 iterations), which is the size the
north_star target is quoted on and fits one GPU.  Voxels are independent, so
with N > 1 every rank solves its own shard of V voxels (weak scaling, no
data-path collective); the only collectives are the timing barrier and the
MAX-reduction of the elapsed time.

Prints ONE JSON line (rank 0): metric = voxel-iterations/s (whole job), plus
`roofline` (dominant kernel against the HBM roofline, algorithmic bytes =
12*N B per voxel-iteration) and `cpu_baseline` (the C/OpenMP
float64 oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_VOXEL_ITER_PER_SCAN = 12   # read w, read y, write w in fp32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--voxels", type=int, default=100000, help="voxels per GPU")
    ap.add_argument("--scans", type=int, default=300)
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--lbda", type=float, default=1.0)
    ap.add_argument("--extras", action="store_true",
                    help="also time BASELINE config 2 (10k voxels) and the PCIe-inclusive solve "
                         "(extra launches of the same kernel; off by default so that a rocprof "
                         "summary of the default command holds only the timed launches)")
    ap.add_argument("--kernel", choices=["auto", "fast1", "generic"], default="auto",
                    help="auto = library dispatch; fast1 = one problem per DPP row "
                         "(fista_fast_kernel) even where the pair kernel applies")
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="target CPU time of the cpu_baseline sample (0 = skip)")
    return ap.parse_args()


def main():
    args = parse()
    # Everything but the final JSON line goes to stderr, including what native
    # libraries (RCCL's version banner) write to file descriptor 1.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = run(args)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
    if line is not None:
        print(line, flush=True)


def run(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from pybold_amd import data, solver
    from pybold_amd.hrf_model import spm_hrf
    from pybold_amd.linear import ConvAndLinear, DiscretInteg
    from pybold_amd.utils import spectral_radius_est

    V, N, n_iter = args.voxels, args.scans, args.iters
    tr = 1.0
    hrf = spm_hrf(1.0, t_r=tr, dur=30.0)[0]                    # canonical HRF, K = 30
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=N * tr / 60.0, tr=tr, hrf=hrf, nb_events=5,
                                           avg_dur=12.0, std_dur=1.0, snr=1.0, seed=1000 + rank,
                                           device=dev)
    np.random.seed(0)                                           # same constant on every rank
    H = ConvAndLinear(DiscretInteg(), hrf, dim_in=N, dim_out=N)
    lipschitz = 0.9 * spectral_radius_est(H, (N,))              # pybold/bold_signal.py:52
    step = 1.0 / lipschitz
    plan = solver.FistaPlan(Y, hrf, args.lbda, step, n_iter,
                            force=None if args.kernel == "auto" else args.kernel)
    kernel_name = (solver.which_kernel(N, len(hrf), V) if args.kernel == "auto" else
                   {"fast1": solver.KERNEL_NAMES[1], "generic": solver.KERNEL_NAMES[0]}[args.kernel])

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        plan.run()
    torch.cuda.synchronize(dev)
    barrier()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(args.steps):
        plan.W.zero_()
        ev[k][0].record()          # same stream the kernel is launched on
        plan.launch()
        ev[k][1].record()
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    total_units = float(world) * V * n_iter * args.steps
    value = total_units / elapsed
    alg_bytes = BYTES_PER_VOXEL_ITER_PER_SCAN * N * float(V) * n_iter      # per launch
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if tj.get("voxels") == V and tj.get("iters") == n_iter and tj.get("scans") == N:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "voxel-iterations/sec", "value": value, "unit": "voxel-iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "wall_clock_to_eps_ms": elapsed / args.steps * 1e3,   # one full solve meeting eps <= 1e-5
        "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "dtype_note": "FIR/scans/residual fp32 (packed), iterate and update fp64 on chip; "
                      "y fp32 in HBM; outputs fp64",
        "data": "synthetic",
        "config": {"workload": "BASELINE config 3 per GPU: %d voxels x %d scans, L1/TV block-signal "
                               "deconv, fixed canonical HRF (K=%d), lambda=%g, %d FISTA iterations "
                               "per step" % (V, N, len(hrf), args.lbda, n_iter),
                   "voxels_per_gpu": V, "scans": N, "taps": int(len(hrf)), "iters_per_step": n_iter,
                   "kernel": kernel_name,
                   "parallelism": "voxel-shard x%d, no data-path collective" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "state-resident multi-iteration kernel: HBM is touched once per "
                             "solve, so the algorithmic-byte rate may exceed the HBM peak; the "
                             "kernel is VALU-issue bound (see DESIGN.md)"},
    }

    if rank == 0 and world == 1 and args.extras:
        # BASELINE config 2 (10k voxels x 300 scans x 500 iterations) for reference
        V2 = min(10000, V)
        plan2 = solver.FistaPlan(Y[:V2].contiguous(), hrf, args.lbda, step, n_iter, force=None)
        plan2.run()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(5):
            plan2.run()
        torch.cuda.synchronize(dev)
        dt2 = (time.perf_counter() - t2) / 5
        out["other_configs"] = {"config2_%d_voxels" % V2: {
            "value": V2 * n_iter / dt2, "unit": "voxel-iterations/s", "ms_per_solve": dt2 * 1e3}}
        # host-buffer variant of the boundary: y starts in pinned host memory and the
        # iterate is copied back (never the headline value)
        Yh = Y.cpu().pin_memory()
        Wh = torch.empty(plan.W.shape, dtype=torch.float64).pin_memory()
        Yd = torch.empty_like(Y)
        planp = solver.FistaPlan(Yd, hrf, args.lbda, step, n_iter, force=None)
        def pcie_step():
            Yd.copy_(Yh, non_blocking=True)
            planp.run()
            Wh.copy_(planp.W, non_blocking=True)
        pcie_step()
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        for _ in range(3):
            pcie_step()
        torch.cuda.synchronize(dev)
        dt3 = (time.perf_counter() - t3) / 3
        out["pcie_inclusive"] = {"value": V * n_iter / dt3, "unit": "voxel-iterations/s",
                                 "ms_per_solve": dt3 * 1e3,
                                 "note": "H2D of y (fp32) + solve + D2H of diff_z (fp64), pinned host buffers"}
        del Yh, Wh, Yd, planp

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"], out["parity"] = cpu_baseline(Y, plan.W, hrf, args.lbda, step, n_iter,
                                                          args.cpu_seconds)
    if dist is not None:
        dist.destroy_process_group()
    return json.dumps(out) if rank == 0 else None


def cpu_baseline(Y, W_gpu, hrf, lbda, step, n_iter, target_s):
    """C/OpenMP float64 port of the reference loop (oracle/fista_oracle.c) on the
    host cores of this box, on the first voxels of the same workload; also the
    parity of the GPU result on that sample."""
    from oracle import c_oracle
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, int(os.environ.get("PYBOLD_BENCH_CPU_THREADS", "16")))   # 1-GPU CPU share
    Yh = Y[:max(4 * cores, 64)].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    c_oracle.fista_batch(Yh[:2 * cores], hrf, lbda, step, n_iter, threads=cores)   # calibrate
    per_voxel = (time.perf_counter() - t0) / (2 * cores)
    n_sample = int(min(Y.shape[0], max(2 * cores, target_s / max(per_voxel, 1e-9))))
    Yh = Y[:n_sample].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    Wc, _, used = c_oracle.fista_batch(Yh, hrf, lbda, step, n_iter, threads=cores)
    dt = time.perf_counter() - t0
    Wg = W_gpu[:n_sample].cpu().numpy()
    err = float((np.linalg.norm(Wg - Wc, axis=1) / (np.linalg.norm(Wc, axis=1) + 1e-300)).max())
    # the reference's own formulation (dense Toeplitz mat-vecs, NumPy float64, as
    # pybold/linear.py:73-113) on ONE core, for context next to the matrix-free C port
    from oracle import pybold_oracle as orc
    t1 = time.perf_counter()
    orc.fista_batch(Yh[:4], hrf, lbda, step, n_iter, dense=True)
    dense_rate = 4 * n_iter / (time.perf_counter() - t1)
    base = {"value": n_sample * n_iter / dt, "unit": "voxel-iterations/s", "cores": int(used),
            "kind": "port", "numpy_dense_toeplitz_1core": dense_rate,
            "sample": "first %d voxels of the same batch x %d iterations, C/OpenMP float64 "
                      "matrix-free port (oracle/fista_oracle.c), %.1f s" % (n_sample, n_iter, dt)}
    parity = {"max_rel_l2_diff_z_vs_cpu_oracle": err, "voxels_checked": n_sample, "tolerance": 1e-5}
    return base, parity


if __name__ == "__main__":
    main()
