#!/usr/bin/env python3
"""Benchmark of the deconvolution hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One *step* = one full solve of the voxel batch: 500 iterations of the reference's
FISTA-like recurrence (pybold/bold_signal.py:62-72) on 300-scan series, L1/TV-regularised,
fixed canonical HRF -- BASELINE.json config 3 (100k voxels x 300 scans x 500 iterations),
the size the north_star target is quoted on; it fits one GPU.

N > 1: one process per GPU.  Started WITHOUT a torch.distributed environment, this script
launches its own ranks (python -m torch.distributed.run ... bench.py) as a child process
before it touches the GPU; under torchrun (the driver's form) it is one of the ranks.
Voxels are independent, so the ranks share the 100k voxels of config 3 in contiguous shards
(`distributed.shard_bounds`, the reference's joblib axis,
examples/icassp_2019/simulation.py:62-72) with NO data-path collective: strong scaling, the
axis of the north_star's ">= 6x at 8 GPUs".  The only collectives are the timing barrier and
the MAX-reduction of the elapsed time.  `--scaling weak` (or the `weak_scaling` key of the
default line for N > 1) gives every rank its own 100k voxels instead.

`--config {2,3,4,5}` picks the BASELINE.json configuration (default 3; the default command is
unchanged): 2 = 10k voxels (L1 deconv), 3 = 100k voxels, 4 = semi-blind deconvolution with ONE
shared HRF dilation (50k voxels, 20 outer x 100 inner iterations + the closing z-step, one
all-reduce of K^2+K+2 float64 per outer iteration), 5 = regularisation path (50k voxels x 20
lambdas = 10^6 problems sharing each voxel's series).  One JSON line each, same blocks.

Prints ONE JSON line (rank 0): metric = voxel-iterations/s (whole job), plus `roofline`
(dominant kernel against the fp32 vector-ALU peak that binds it; the HBM-algorithmic figure
of SURVEY.md 8d and the measured HBM traffic as context) and `cpu_baseline` (the reference's
dense-Toeplitz NumPy formulation fanned over the host cores, and the matrix-free C/OpenMP
port, both from oracle/, on bounded samples of the same workload).
"""
import argparse
import math
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector), needs v_pk_fma_f32
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense BF16/FP16 MFMA peak
BYTES_PER_VOXEL_ITER_PER_SCAN = 12   # read w, read y, write w in fp32


CONFIG_VOXELS = {2: 10000, 3: 100000, 4: 50000, 5: 50000}
N_LAMBDA = 20                        # config 5: lambdas per voxel


DENSE_RATIO = 0.19        # include/pybold_hip.h: PB_PATH_DENSE_RATIO


def executed_flops_per_voxel_iter(n, k):
    """What the pair kernel with 2-parallel fast FIRs executes: three half-length sub-filters
    per FIR instead of four (3/4 of the multiply-adds), same scans and update."""
    return 3.0 * n * k + 12.0 * n


def mfma_per_wave_iteration(n, k, split=False):
    """v_mfma_f32_16x16x32_f16 instructions per iteration and 16 problems: one wave of fista_mfma_kernel
    (NB = ceil(N/31) blocks of 31 samples + one sum slot, NT near tiles (2 for K <= 33, 3 for K <= 64), three
    split products per tile and row half, both passes; the far field rides in the sum slot and costs no
    instruction of its own; pybold_amd/csrc/fista_mfma.h -- 228 at N = 300, K = 30, the count SQ_INSTS_MFMA
    reports, profiles/r4_pmc_mfma.json; round 3's carry-tile form: 276) or, `split`, the two waves of
    fista_mfma2_kernel together (fista_mfma2.h: floor(NB/2) + ceil(NB/2) blocks of 32 samples with the
    carry tile; the waves' carry chains restart at the cut: 270 at N = 300)."""
    nt = 2 if k <= 33 else 3
    if split == 4:
        # fista_mfma4_kernel: four waves of A = ceil(N / 128) blocks of 32 samples; per wave and pass 15 A matrix
        # instructions less what the ends of the series lack (fista_mfma4.h; 576 at N = 1 200: 147 x 7 passes + 138)
        a = (n + 127) // 128
        mid = 2 * (3 * (a - 1) + 12 * a)                       # a middle wave, both passes (neighbours on both sides)
        first = (3 * (a - 2) + 6 * a + 6 * (a - 1)) + (3 * (a - 1) + 12 * a)
        last = (3 * (a - 1) + 12 * a) + (3 * (a - 2) + 6 * a + 6 * (a - 1))
        return 2 * mid + first + last
    if split:
        nb = (n + 31) // 32
        a, b = nb // 2, nb - nb // 2
        return (3 * (a - 2) + 6 * a + 6 * (a - 1)) + (3 * (b - 1) + 12 * b) + (3 * (b - 2) + 6 * b + 6 * (b - 1)) + (3 * (a - 1) + 12 * a)
    nb = (n + 30) // 31
    return 2 * 6 * sum(nb - o for o in range(nt))


MFMA_FLOP = 2.0 * 16 * 16 * 32       # one v_mfma_f32_16x16x32_f16


def flops_per_voxel_iter(n, k):
    """Algorithmic flops of one iteration: two K-tap FIRs 4NK, two scans 2N,
    update/prox/momentum ~10N."""
    return 4.0 * n * k + 12.0 * n


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=[2, 3, 4, 5], default=3,
                    help="BASELINE.json configuration (see the module docstring)")
    ap.add_argument("--voxels", type=int, default=None,
                    help="voxels of the whole job (strong scaling) / per GPU (--scaling weak); "
                         "default: the configuration's own count (10k / 100k / 50k / 50k)")
    ap.add_argument("--busy-seconds", type=float, default=6.0,
                    help="N = 1: untimed back-to-back solves after the timed region (sustained rate; also "
                         "keeps the GPU visibly busy between the two CPU legs)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: shard --voxels over the ranks (strong, default) or give every "
                         "rank --voxels of its own (weak)")
    ap.add_argument("--scans", type=int, default=300)
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--lbda", type=float, default=1.0)
    ap.add_argument("--extras", action="store_true",
                    help="also time BASELINE config 2 (10k voxels), the 12 500-voxel shard of "
                         "config 3 on 8 GPUs and the chunked (pipelined) host-buffer solve (extra "
                         "launches; off by default so that a rocprof summary of the default command "
                         "holds only the config-3 launches)")
    ap.add_argument("--kernel", choices=["auto", "seq", "fast1", "generic"], default="auto",
                    help="auto = library dispatch; seq = the same without the internal side "
                         "stream; fast1 = one problem per DPP row (fista_fast_kernel) even "
                         "where the pair kernel applies")
    ap.add_argument("--spin-seconds", type=float, default=0.25,
                    help="untimed busy time before the warm-up steps (GPU clock ramp)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0,
                    help="target time of each cpu_baseline sample (0 = skip)")
    ap.add_argument("--master-port", type=int, default=0, help="self-launched ranks only")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="development aid: all N ranks share GPU 0 and synchronise over gloo (RCCL "
                         "refuses two ranks on one device); rehearses launch, sharding, barrier and "
                         "MAX-reduction of the N > 1 path on a one-GPU box. The line is marked "
                         "\"rehearsal\": true and is not a measurement")
    return ap.parse_args()


def job_layout(voxels, scaling, world, rank):
    """Rows `[lo, hi)` of the job this rank solves and the job's total voxel count: strong =
    contiguous shards of `voxels` (the reference's joblib axis), weak = `voxels` per rank."""
    from pybold_amd.distributed import shard_bounds
    if scaling == "strong":
        lo, hi = shard_bounds(voxels, world, rank)
        return lo, hi, voxels
    return rank * voxels, (rank + 1) * voxels, voxels * world


def self_launch(args):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as ONE child process
    (torch.distributed.run) and pass its exit code on.  Nothing here touches the GPU:
    `torch.cuda.device_count()` does not initialise it."""
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and not (args.rehearse_on_one_gpu and have >= 1):
        sys.stderr.write("bench.py --gpus %d needs %d GPUs; this machine shows %d\n"
                         % (args.gpus, args.gpus, have))
        return 2
    port = args.master_port
    if not port:
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.voxels is None:
        args.voxels = CONFIG_VOXELS[args.config]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
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
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    cfg = args.config

    # CPU legs first (rank 0, one GPU): their worker processes are started before this
    # process initialises the GPU
    cpu_dense = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        cpu_dense = cpu_dense_baseline(args.scans, args.iters, args.lbda, args.cpu_seconds)

    import numpy as np
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    red_dev = torch.device("cpu") if args.rehearse_on_one_gpu else dev
    collective = None
    if dist is not None:
        # what the communicator saw: one all-gather of every rank's PCI address (every rank takes part), so that a
        # multi-GPU line shows RCCL ran over N DIFFERENT devices (a rehearsal on one GPU shows 1)
        pr = torch.cuda.get_device_properties(dev)
        addr = (int(getattr(pr, "pci_domain_id", 0)) << 16) | (int(getattr(pr, "pci_bus_id", 0)) << 8) | int(getattr(pr, "pci_device_id", 0))
        mine = torch.tensor([addr], dtype=torch.int64, device=red_dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        ids = [int(t.item()) for t in every]
        collective = {"backend": dist.get_backend(), "world_size": world, "distinct_devices": len(set(ids)),
                      "pci_addresses": ["%04x:%02x:%02x" % (i >> 16, (i >> 8) & 0xff, i & 0xff) for i in ids]}

    from pybold_amd import data, solver
    from pybold_amd.hrf_model import spm_hrf
    from pybold_amd.linear import ConvAndLinear, DiscretInteg
    from pybold_amd.utils import spectral_radius_est

    lo, hi, V_total = job_layout(args.voxels, args.scaling, world, rank)
    V = hi - lo                                                 # this rank's voxels

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed_local(fn, steps, warmup, spin_s=0.0):
        """This rank only (no collective): `warmup` untimed calls, then `steps` calls bracketed by
        events on the launch stream.  Returns (elapsed s by the host clock, mean ms by events)."""
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < spin_s:
            fn()
            torch.cuda.synchronize(dev)
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()          # same stream the kernel is launched on
            fn()
            ev[k][1].record()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, float(np.mean([a.elapsed_time(b) for a, b in ev]))

    def timed(fn, steps, warmup, spin_s=0.0):
        """The contract's timed region, on EVERY rank: `warmup` untimed steps, then exactly `steps`
        steps between barrier + synchronize on both sides; MAX over ranks.  `spin_s` > 0 first
        keeps the device busy with the same (untimed) step for that long: a step of a 12 500-voxel
        shard lasts 2.7 ms, and three warm-up steps end before the GPU has left its idle clocks
        (measured: 2.80 ms/step after 3 warm-up steps, 2.63 after 30 or more)."""
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < spin_s:
            fn()
            torch.cuda.synchronize(dev)
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(steps):
            ev[k][0].record()
            fn()
            ev[k][1].record()
        torch.cuda.synchronize(dev)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if cfg == 4:
        out = run_config4(args, world, rank, dev, dist, V, V_total, lo, timed, timed_local, barrier)
        if collective:
            out["collective"] = collective
        if rank == 0 and world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline_block(cpu_dense, None, args.iters)
        if dist is not None:
            dist.destroy_process_group()
        return json.dumps(out) if rank == 0 else None

    N, n_iter = args.scans, args.iters
    tr = 1.0
    hrf = spm_hrf(1.0, t_r=tr, dur=30.0)[0]                    # canonical HRF, K = 30
    K = len(hrf)
    y_rep = N_LAMBDA if cfg == 5 else 1

    def make_batch(n_vox, seed):
        Y, _, _ = data.gen_rnd_bloc_bold_batch(n_vox, dur=N * tr / 60.0, tr=tr, hrf=hrf, nb_events=5,
                                               avg_dur=12.0, std_dur=1.0, snr=1.0, seed=seed, device=dev)
        return Y
    Y = make_batch(V, 1000 + rank)
    np.random.seed(0)                                           # same constant on every rank
    H = ConvAndLinear(DiscretInteg(), hrf, dim_in=N, dim_out=N)
    lipschitz = 0.9 * spectral_radius_est(H, (N,))              # pybold/bold_signal.py:52
    step = 1.0 / lipschitz
    force = None if args.kernel == "auto" else args.kernel
    if cfg == 5:      # lambda_v,i = logspace(-2, 0, 20)_i * ||H^T y_v||_inf, one problem per (voxel, lambda)
        lam = (solver.lambda_max(Y, hrf)[:, None] *
               torch.logspace(-2.0, 0.0, N_LAMBDA, dtype=torch.float64, device=dev)[None, :]).reshape(-1)
    else:
        lam = args.lbda
    P = V * y_rep                                               # problems of this rank
    lmax5 = None
    if cfg == 5 and args.kernel == "auto":
        lmax5 = solver.lambda_max(Y, hrf)      # (the path's own top: what the caller built its lambdas from)
    plan = solver.FistaPlan(Y, hrf, lam, step, n_iter, y_rep=y_rep, force=force, lmax=lmax5)
    plan0 = solver.launch_plan(N, K, max(P, 1), force="valu" if torch.is_tensor(lam) else None)
    kernel_name = ((plan0[1] if plan0[0] > 0 else plan0[2]) if args.kernel in ("auto", "seq") else
                   {"fast1": solver.KERNEL_NAMES[1], "generic": solver.KERNEL_NAMES[0]}[args.kernel])

    def one_step():
        plan.launch(cold=True)     # deconv's cold start (w = 0) without a memset: PB_FLAG_COLD_START

    elapsed, kern_ms = timed(one_step, args.steps, args.warmup, args.spin_seconds)
    value = float(V_total) * y_rep * n_iter * args.steps / elapsed

    # One step = ONE pb_fista_solve call = up to two kernels (solver.launch_plan): the whole
    # rounds of waves on the dominant kernel, the remainder on the cheapest form.  The
    # dominant kernel is timed on its own here (same grid, same data, a few launches after
    # the timed region, THIS RANK ONLY: no collective, ranks may differ in their plans) so that
    # its duration can be set against its rocprofv3 average.
    # (per-problem lambdas -- config 5's regularisation path, sparse solutions by construction -- stay
    # on the vector forms: the library does not put them on the matrix pipe)
    plan_force = "valu" if torch.is_tensor(lam) else None
    n_main, main_kernel, tail_kernel = solver.launch_plan(N, K, max(P, 1), force=plan_force)
    P_dom = n_main if (args.kernel in ("auto", "seq") and n_main > 0) else P
    matrix_pipe = args.kernel in ("auto", "seq") and "matrix pipe" in (main_kernel if n_main else tail_kernel)
    split_form = matrix_pipe and "split over two" in (main_kernel if n_main else tail_kernel)
    if matrix_pipe and "four waves" in (main_kernel if n_main else tail_kernel):
        split_form = 4                                          # (641 .. 1 280 scans: one series over the four waves of a workgroup)
    if P_dom != P and P_dom % y_rep == 0:
        lam_dom = lam[:P_dom] if torch.is_tensor(lam) else lam
        plan_dom = solver.FistaPlan(Y[:P_dom // y_rep], hrf, lam_dom, step, n_iter, y_rep=y_rep,
                                    force=("mfma2" if split_form else "mfma") if matrix_pipe else "fast2")
        _, dom_ms = timed_local(lambda: plan_dom.launch(cold=True), max(3, min(args.steps, 5)), 1, 0.05)
        del plan_dom
    else:
        P_dom, dom_ms = P, kern_ms
    pair_form = args.kernel in ("auto", "seq") and "two problems per row" in (main_kernel if n_main else tail_kernel)
    path_info = None
    if lmax5 is not None:
        # regularisation path partitioned on the device (pb_fista_solve_path): two big launches per step -- the dense
        # class on the matrix-pipe form, the sparse class on the pair form; each is timed on its own (same lists), the
        # longer one is the step's dominant kernel
        n_dense = int((lam < DENSE_RATIO * lmax5.repeat_interleave(y_rep)).sum().item())     # (the library's own test, path.h)
        # (the dense class alone, without the re-solve: what its guards leave at n_done = -1)
        _, _, nd0 = solver.fista_solve(Y, hrf, lam, step, n_iter, y_rep=y_rep, lmax=lmax5, force="path_dense")
        n_back = int((nd0[lam < DENSE_RATIO * lmax5.repeat_interleave(y_rep)] < 0).sum())
        t_cls = {}
        for tag, frc in (("dense", "path_dense"), ("sparse", "path_sparse")):
            pl = solver.FistaPlan(Y, hrf, lam, step, n_iter, y_rep=y_rep, force=frc, lmax=lmax5, W=plan.W)
            _, t_cls[tag] = timed_local(lambda: pl.launch(cold=True), 3, 1, 0.0)
        plan.launch(cold=True)                                  # the complete result again (the aids leave a class unsolved)
        torch.cuda.synchronize(dev)
        matrix_pipe = t_cls["dense"] >= t_cls["sparse"]
        pair_form, split_form = not matrix_pipe, False
        P_dom, dom_ms = (n_dense, t_cls["dense"]) if matrix_pipe else (P - n_dense, t_cls["sparse"])
        main_kernel, tail_kernel, n_main = solver.KERNEL_NAMES[4], solver.KERNEL_NAMES[2], n_dense
        kernel_name = main_kernel if matrix_pipe else tail_kernel
        path_info = {"dense_ratio": DENSE_RATIO, "dense_problems": n_dense, "sparse_problems": P - n_dense,
                     "handed_back_by_the_guards": n_back, "handed_back_fraction_of_dense": n_back / max(n_dense, 1),
                     "dense_launch_ms": t_cls["dense"], "sparse_launch_ms": t_cls["sparse"],
                     "note": "each launch includes the three partition launches (lists built on the device, no host "
                             "synchronisation) and covers P slots: waves beyond a list's length leave at once"}
    flops_launch = flops_per_voxel_iter(N, K) * float(P_dom) * n_iter      # dominant kernel, one launch
    exec_launch = (executed_flops_per_voxel_iter(N, K) if pair_form else flops_per_voxel_iter(N, K)) * float(P_dom) * n_iter
    mfma_block = None
    if matrix_pipe:
        # what the matrix-pipe kernel EXECUTES: mfma_per_wave_iteration() matrix instructions per wave of 16 problems
        n_mfma = mfma_per_wave_iteration(N, K, split_form)
        mfma_flop = n_mfma * MFMA_FLOP / 16.0
        exec_launch = mfma_flop * float(P_dom) * n_iter
        mfma_block = {"mfma_instructions_per_16_problems_and_iteration": n_mfma,
                      "waves_per_16_problems": 4 if split_form == 4 else (2 if split_form else 1),
                      "f16_flops_per_voxel_iteration": mfma_flop,
                      "matrix_pipe_busy_estimate": n_mfma * 16.0 / 16.0 * float(P_dom) * n_iter /
                                                   (1024.0 * 2.1e9 * dom_ms * 1e-3),
                      "note": "three float16 split products per tile (hi.hi, hi.lo, lo.hi), scans folded into the "
                              "tiles (a third of the tile entries are structural zeros); one-wave form: the far field rides "
                              "in a sum slot of the near tiles (no product of its own), split form: one carry tile per block.  "
                              "The kernel is bound by the issue of one wave per SIMD (the matrix instructions hold the "
                              "vector issue port for 8 of their 16 cycles) at the package power limit, not by the matrix pipe"}
    alg_bytes = BYTES_PER_VOXEL_ITER_PER_SCAN * N * float(P_dom) * n_iter
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tfile) and cfg == 3:
        try:
            tj = json.load(open(tfile))
            if tj.get("voxels") == P_dom and tj.get("iters") == n_iter and tj.get("scans") == N:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    ms_step = elapsed / args.steps * 1e3
    what = {2: "BASELINE config 2: %d voxels x %d scans, L1 deconv" % (V_total, N),
            3: "BASELINE config 3: %d voxels x %d scans, L1/TV block-signal deconv" % (V_total, N),
            5: "BASELINE config 5: regularisation path, %d voxels x %d lambdas (logspace(-2,0,%d) x "
               "lambda_max of each voxel) = %d problems x %d scans sharing each voxel's series"
               % (V_total, N_LAMBDA, N_LAMBDA, V_total * N_LAMBDA, N)}[cfg]
    if world == 1:
        workload = ("%s, fixed canonical HRF (K=%d), %s%d FISTA iterations per step"
                    % (what, K, "" if cfg == 5 else "lambda=%g, " % args.lbda, n_iter))
    else:
        workload = ("%s on %d GPUs (%s scaling): %s voxels per GPU, fixed canonical HRF (K=%d), %s%d FISTA "
                    "iterations per step" % (what, world, args.scaling,
                                             "%d" % V if args.scaling == "weak" else "ceil(%d/%d)" % (V_total, world),
                                             K, "" if cfg == 5 else "lambda=%g, " % args.lbda, n_iter))
    out = {
        "metric": "voxel-iterations/sec", "value": value, "unit": "voxel-iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "untimed_spin_s": args.spin_seconds,
        **({"rehearsal": True} if args.rehearse_on_one_gpu else {}),
        "ms_per_step": ms_step,
        "wall_clock_to_eps_ms": ms_step,       # one full solve meeting eps <= 1e-5, y resident in HBM
        "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, **dtype_fields(matrix_pipe),
        "data": "synthetic",
        "config": {"workload": workload, "baseline_config": cfg,
                   "voxels_total": V_total, "voxels_this_rank": V, "problems_this_rank": P,
                   "scans": N, "taps": int(K),
                   "iters_per_step": n_iter, "kernel": kernel_name,
                   "launches_per_step": ([{"kernel": main_kernel, "problems": n_main}] if n_main else []) +
                                        [{"kernel": tail_kernel, "problems": P - n_main}]
                   if args.kernel in ("auto", "seq") else [{"kernel": kernel_name, "problems": P}],
                   **({"regularisation_path": path_info} if path_info else {}),
                   "parallelism": "contiguous voxel shards x%d, no data-path collective" % world},
        "roofline": roofline_block(matrix_pipe, kernel_name if path_info else (main_kernel if n_main else tail_kernel), dom_ms, P_dom, n_iter, N, K,
                                   flops_launch, exec_launch, alg_bytes, traffic, mfma_block,
                                   {"step_kernels_ms": kern_ms,
                                    **({} if path_info else {
                                        "step_frac": (exec_launch if matrix_pipe else flops_launch) * (float(P) / P_dom) /
                                                     (kern_ms * 1e-3) / 1e12 /
                                                     (MFMA_F16_PEAK_TFLOPS if matrix_pipe else VALU_FP32_PEAK_TFLOPS)})}),
    }

    if collective:
        out["collective"] = collective
    if world > 1 and args.scaling == "strong":
        # secondary figure: weak scaling (every rank its own `--voxels` voxels); every rank takes
        # part (the decision depends on `world` and the flags only, never on a rank's own plan)
        Yw = make_batch(args.voxels, 2000 + rank)
        if cfg == 5:
            lamw = (solver.lambda_max(Yw, hrf)[:, None] *
                    torch.logspace(-2.0, 0.0, N_LAMBDA, dtype=torch.float64, device=dev)[None, :]).reshape(-1)
        else:
            lamw = args.lbda
        planw = solver.FistaPlan(Yw, hrf, lamw, step, n_iter, y_rep=y_rep, force=force)
        steps_w = max(3, min(args.steps, 5))
        el_w, k_w = timed(lambda: planw.launch(cold=True), steps_w, 1)
        out["weak_scaling"] = {"value": float(args.voxels) * y_rep * world * n_iter * steps_w / el_w,
                               "unit": "voxel-iterations/s", "voxels_per_gpu": args.voxels,
                               "ms_per_step": el_w / steps_w * 1e3, "steps": steps_w}
        del Yw, planw

    if rank == 0 and world == 1 and args.busy_seconds > 0:
        # sustained rate: back-to-back solves for several seconds (clocks and temperatures settled)
        n_busy, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < args.busy_seconds:
            for _ in range(20):
                one_step()
            torch.cuda.synchronize(dev)
            n_busy += 20
        dt = time.perf_counter() - t0
        out["sustained"] = {"value": float(P) * n_iter * n_busy / dt, "unit": "voxel-iterations/s",
                            "seconds": dt, "solves": n_busy}

    if rank == 0 and world == 1 and cfg in (2, 3):
        # wall clock to eps on diff_z, z AND x -- the solve plus pb_fista_outputs (z = cumsum(w), x = h * z)
        taps_d = torch.from_numpy(np.ascontiguousarray(hrf, dtype=np.float64)).to(dev)
        Zb, Xb = torch.empty_like(plan.W), torch.empty_like(plan.W)

        def solve_and_outputs():
            plan.launch(cold=True)
            solver.fista_outputs_into(plan.W, taps_d, Zb, Xb)
        _, ms_out = timed_local(solve_and_outputs, max(3, min(args.steps, 10)), 1, 0.05)
        out["wall_clock_to_eps_ms_incl_outputs"] = ms_out
        del Zb, Xb
    if rank == 0 and world == 1 and cfg == 3:
        out.update(pcie_inclusive(plan, Y, hrf, args.lbda, step, n_iter, solver, torch, dev, args.extras))

    if rank == 0 and world == 1 and args.extras and cfg == 3:
        others = {}
        for tag, V2 in (("config2_10000_voxels", 10000), ("config3_shard_of_8_12500_voxels", 12500)):
            V2 = min(V2, V)
            plan2 = solver.FistaPlan(Y[:V2].contiguous(), hrf, args.lbda, step, n_iter, force=None)
            el2, k2 = timed_local(lambda: plan2.launch(cold=True), 10, 1, 0.1)
            others[tag] = {"value": V2 * n_iter * 10 / el2, "unit": "voxel-iterations/s",
                           "ms_per_solve": el2 / 10 * 1e3, "kernel_ms": k2,
                           "kernel": solver.which_kernel(N, K, V2)}
        out["other_configs"] = others

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        port, out["parity"] = cpu_port_and_parity(Y, plan.W, hrf, lam, y_rep, step, n_iter, args.cpu_seconds,
                                                  cpu_dense, n_main)
        out["cpu_baseline"] = cpu_baseline_block(cpu_dense, port, n_iter)
    if dist is not None:
        dist.destroy_process_group()
    return json.dumps(out) if rank == 0 else None


def dtype_fields(matrix_pipe):
    """`dtype` = the arithmetic the dominant kernel computes its operators in; the note says the rest."""
    if matrix_pipe:
        return {"dtype": "f16x2-split operands (22 bit), f32 accumulate, f64 iterate",
                "dtype_note": "operators (HRF convolution + cumulative sum and their adjoints): float16 hi/lo split "
                              "products (22-bit operands, three v_mfma_f32_16x16x32_f16 per tile), float32 accumulation; "
                              "iterate, gradient step, threshold and momentum float64 on chip; y float32 in HBM; "
                              "outputs float64.  Problems whose operands leave the float16 range or whose solution is "
                              "too sparse for 22-bit operators (th > 0.02 max|w|) are re-solved by the float32 vector "
                              "kernels in the same call (include/pybold_hip.h, pb_fista_solve)"}
    return {"dtype": "f32 operators, f64 iterate",
            "dtype_note": "operators (FIRs, scans, residual) float32 on the vector pipe (v_pk_fma_f32), iterate and update "
                          "float64 on chip; y float32 in HBM; outputs float64"}


def roofline_block(matrix_pipe, kernel, dom_ms, P_dom, n_iter, N, K, flops_launch, exec_launch, alg_bytes, traffic,
                   mfma_block, extra):
    """The `roofline` object of a line.  `bound` names the pipe that binds the dominant kernel:
    "mfma" (matrix-pipe form: `achieved` = the float16 flops its matrix instructions EXECUTE per
    launch / its duration, `peak` = the dense f16 MFMA peak, 2.5 PFLOP/s) or "valu_fp32" (vector
    forms: ALGORITHMIC flops of the direct form, 4NK + 12N per voxel-iteration, against the fp32
    vector peak).  `frac_algorithmic` prices the algorithmic flops against the same peak;
    `hbm_algorithmic_frac` is SURVEY 8d's figure (12 N bytes per voxel-iteration / duration against
    8 TB/s; above 1 because the state never leaves the chip during a solve); `traffic` = HBM bytes
    per launch by PMC."""
    sec = dom_ms * 1e-3
    hbm_alg_gbs = alg_bytes / sec / 1e9
    if matrix_pipe:
        ach, peak = exec_launch / sec / 1e12, MFMA_F16_PEAK_TFLOPS
        head = {"bound": "mfma", "binds": "issue", "headline": "frac_algorithmic",
                "pipe": "mfma_f16 (v_mfma_f32_16x16x32_f16, dense peak)", "achieved": ach, "peak": peak,
                "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                "frac_algorithmic": flops_launch / sec / 1e12 / peak,
                "executed_flops_per_launch": exec_launch}
        if mfma_block and not mfma_block.get("waves_per_16_problems", 1) > 1:
            # the roof that actually binds: the issue rate of ONE wave per SIMD for this kernel's instruction mix, measured
            # on its own (tools/microbench/mfma_narrow.hip, profiles/r5_mfma_narrow_and_issue_model.txt: a matrix
            # instruction followed by v vector instructions of the kernel's kinds takes 13.3 ns at v = 4, + 2.9 ns per
            # further one, whatever the number of accumulator chains)
            n_m = mfma_block["mfma_instructions_per_16_problems_and_iteration"]
            n_v = 996.0 * n_m / 228.0                          # (PMC: 996 vector instructions beside 228 matrix ones at N = 300)
            ns_model = n_m * (13.3 + 2.9 * (n_v / n_m - 4.0))
            waves = float(P_dom) / 16.0
            model_ms = ns_model * 1e-6 * n_iter * max(1.0, math.ceil(waves / 1024.0))
            head["issue_roof"] = {"matrix_instructions_per_wave_iteration": n_m, "vector_instructions_per_wave_iteration": n_v,
                                  "model_ns_per_wave_iteration": ns_model, "model_ms_per_launch": model_ms,
                                  "measured_ms_per_launch": dom_ms, "frac": model_ms / dom_ms,
                                  "note": "model = the microbenchmark's time for this mix at one wave per SIMD (box and clock of "
                                          "that run); frac ~ 1 means no issue slot is lost to latency: only fewer instructions, "
                                          "or a second wave per SIMD (x1.15-1.23, not reachable: 304 registers of state per "
                                          "wave), would be faster"}
        elif mfma_block:
            # the split forms: the same model on ONE wave's share of the instructions (SQ counters, profiles/r5_pmc_sq_600_scans.json,
            # r5_pmc_sq_1200_scans.json: 995.6 vector instructions per wave-iteration at 19 blocks over two waves, 1 128.5 at ten
            # blocks per wave over four); what the model leaves is the price of the two workgroup barriers per iteration
            wpg = mfma_block["waves_per_16_problems"]
            n_m = mfma_block["mfma_instructions_per_16_problems_and_iteration"] / float(wpg)
            blocks_w = math.ceil(N / 32.0) / 2.0 if wpg == 2 else math.ceil(N / 128.0)
            n_v = 995.6 * blocks_w / 9.5 if wpg == 2 else 1128.5 * blocks_w / 10.0
            v = n_v / n_m
            ns_model = n_m * (13.3 + (2.9 if v >= 4.0 else 1.95) * (v - 4.0))
            groups = float(P_dom) / 16.0
            model_ms = ns_model * 1e-6 * n_iter * max(1.0, math.ceil(groups * wpg / 1024.0))
            head["issue_roof"] = {"matrix_instructions_per_wave_iteration": n_m, "vector_instructions_per_wave_iteration": n_v,
                                  "model_ns_per_wave_iteration": ns_model, "model_ms_per_launch": model_ms,
                                  "measured_ms_per_launch": dom_ms, "frac": model_ms / dom_ms,
                                  "note": "split form: the microbenchmark's time for ONE wave's instruction mix (one wave per SIMD); "
                                          "what is missing to 1 is the two workgroup barriers per iteration (waves wait for the "
                                          "slowest share, then for the exchanged scalars)"}
        binding = ("issue of ONE wave per SIMD: 996 vector + 228 matrix instructions per 16 voxel-iterations at N = 300 "
                   "(profiles/r4_pmc_sq.json), at the package power limit -- neither HBM nor the MFMA peak.  Round 5 "
                   "measured the roof itself (profiles/r5_mfma_narrow_and_issue_model.txt, r5_valu_one_wave.txt): a lone "
                   "wave issues a vector instruction every 5.4-8 cycles (two waves per SIMD: every 4), and the kernel's "
                   "time equals 228 x the time of one matrix instruction followed by 4.4 vector instructions of its kinds, "
                   "whether the products form one accumulator chain or four: no latency slack is left (roofline.issue_roof); "
                   "v_mfma_f32_16x16x16_f16 costs 0.92 of the 16x16x32 form, so skipping the empty quarters of the tiles with "
                   "narrow products does not pay.  `frac` prices EXECUTED f16 flops (5.9x the algorithmic count: three split "
                   "products, structural zeros, scans folded into the tiles): a pipe-busy figure; `frac_algorithmic` is the "
                   "one that tracks progress")
    else:
        ach, peak = flops_launch / sec / 1e12, VALU_FP32_PEAK_TFLOPS
        head = {"bound": "valu_fp32", "pipe": "vector fp32 (v_pk_fma_f32)", "achieved": ach, "peak": peak,
                "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                "frac_algorithmic": ach / peak,
                "executed_flops_per_launch": exec_launch}
        binding = "vector issue (VALU busy 98 %, profiles/r2_pmc_sq_valu_utilisation.json)"
    head.update({
        "hbm_algorithmic_frac": hbm_alg_gbs / HBM_PEAK_GBS, "hbm_algorithmic_GBps": hbm_alg_gbs,
        "hbm_traffic_GBps": (traffic / sec / 1e9) if traffic else None,
        "kernel": kernel, "kernel_ms": dom_ms, "kernel_problems": P_dom, "kernel_iterations": n_iter,
        "algorithmic_flops_per_launch": flops_launch, "algorithmic_bytes_per_launch": alg_bytes,
        "flops_per_voxel_iteration": flops_per_voxel_iter(N, K),
        "executed_flops_per_voxel_iteration": exec_launch / (float(P_dom) * n_iter),
        **({"matrix_pipe": mfma_block} if mfma_block else {}),
        "binding": binding,
        "note": "dominant kernel of the step, timed on its own with HIP events on its launch stream (kernel_problems of the "
                "problems; the rest runs in short launches behind it: config.launches_per_step).  Register-resident "
                "multi-iteration kernel: a launch reads y once and writes w once (`traffic`), so SURVEY 8d's "
                "algorithmic-byte rate exceeds the HBM peak (hbm_algorithmic_frac > 1 is not a utilisation figure).  "
                "The peaks assume 2.4 GHz; under these kernels the package sits at its power limit and the shader clock "
                "at 2.04-2.33 GHz depending on the box (profiles/r4_clock_and_power_during_solve.txt)."})
    head.update(extra)
    return head


def run_config4(args, world, rank, dev, dist, V, V_total, lo, timed, timed_local, barrier):
    """BASELINE config 4: semi-blind deconvolution, ONE HRF dilation shared by every voxel of every
    rank (`distributed.bd_shared`; structure of pybold/bold_signal.py:281-382): 20 outer iterations
    of z-step (100 inner iterations, step 1/||A^T A||_F) and theta-step (normal equations ->
    all-reduce of K^2+K+2 float64 -> 1-D search on the device), then the closing z-step.  One
    bench step = one whole `bd_shared` call = 21 z-steps of 100 iterations."""
    import hashlib
    import numpy as np
    import torch
    from pybold_amd import data, distributed, solver
    from pybold_amd.hrf_model import spm_hrf
    t_r, hrf_dur, lbda, nb_outer, nb_inner, theta_true = 0.75, 20.0, 1.7, 20, 100, 0.7
    N = args.scans
    h_true = spm_hrf(theta_true, t_r, hrf_dur, False)[0]
    K = len(h_true)
    Y, _, _ = data.gen_rnd_bloc_bold_batch(V, dur=N * t_r / 60.0, tr=t_r, hrf=h_true, nb_events=5, avg_dur=12.0,
                                           std_dur=1.0, snr=10.0, seed=4000 + rank, device=dev)
    comm = distributed.Comm()                    # RCCL under torchrun, gloo in the rehearsal, nothing for one process
    res = {}
    # one bench step = one whole bd_shared call, submitted as ONE HIP graph (captured once: 21 z-steps, normal
    # equations, all-reduces, theta fits -- ~130 launches); the eager loop where capture is not possible (gloo rehearsal)
    runner = distributed.BdSharedGraph(Y, t_r, lbda=lbda, theta_0=2.0, hrf_dur=hrf_dur, nb_iter=nb_outer,
                                       nb_inner=nb_inner, comm=comm)

    def one_step():
        runner.launch()

    elapsed, step_ms = timed(one_step, args.steps, args.warmup, args.spin_seconds)
    res["W"], res["h"], res["d"] = runner.result()

    def eager_step():
        distributed.bd_shared(Y, t_r, lbda=lbda, theta_0=2.0, hrf_dur=hrf_dur, nb_iter=nb_outer, nb_inner=nb_inner, comm=comm)
    el_eager, _ = timed(eager_step, max(2, min(args.steps, 5)), 1, 0.0)       # (every rank: the loop holds collectives)
    eager_ms = el_eager / max(2, min(args.steps, 5)) * 1e3
    n_z = nb_outer + 1
    value = float(V_total) * n_z * nb_inner * args.steps / elapsed
    d = res["d"]
    # the z-steps alone (this rank, no collective): the shared-HRF pair kernel reading taps and step
    # from device memory, warm-started, n_z launches of nb_inner iterations
    taps = torch.from_numpy(np.ascontiguousarray(res["h"])).to(dev)
    stepc = 1.0 / solver.gram_frobenius_batch(taps.reshape(1, -1), N)
    Wz = torch.zeros((V, N), dtype=torch.float64, device=dev)

    def z_steps():
        for _ in range(n_z):
            solver.fista_solve_pp(Y, taps, stepc, lbda, nb_inner, W0=Wz, inplace=True)
    _, z_ms = timed_local(z_steps, 3, 1, 0.05)
    n_main, main_kernel, tail_kernel = solver.launch_plan(N, K, max(V, 1))
    V_dom = n_main if n_main > 0 else V
    Wd = torch.zeros((V_dom, N), dtype=torch.float64, device=dev)
    Yd = Y[:V_dom]
    matrix_pipe = "matrix pipe" in (main_kernel if n_main else tail_kernel)
    split_form = matrix_pipe and "split over two" in (main_kernel if n_main else tail_kernel)
    _, dom_ms = timed_local(lambda: solver.fista_solve_pp(Yd, taps, stepc, lbda, nb_inner, W0=Wd, inplace=True,
                                                          force="intermediate" if matrix_pipe else "fast2"), 10, 2, 0.05)
    dom_kernel = main_kernel if n_main else tail_kernel
    flops_launch = flops_per_voxel_iter(N, K) * float(V_dom) * nb_inner
    exec_launch = executed_flops_per_voxel_iter(N, K) * float(V_dom) * nb_inner
    mfma_block = None
    if matrix_pipe:
        n_mfma = mfma_per_wave_iteration(N, K, split_form)
        exec_launch = n_mfma * MFMA_FLOP / 16.0 * float(V_dom) * nb_inner
        mfma_block = {"mfma_instructions_per_16_problems_and_iteration": n_mfma,
                      "waves_per_16_problems": 2 if split_form else 1,
                      "f16_flops_per_voxel_iteration": n_mfma * MFMA_FLOP / 16.0}
    alg_bytes = BYTES_PER_VOXEL_ITER_PER_SCAN * N * float(V_dom) * nb_inner
    ms_step = elapsed / args.steps * 1e3
    theta = np.asarray(d["theta"], dtype=np.float64)
    dt = dtype_fields(matrix_pipe)
    dt["dtype_note"] = ("z-steps: " + dt["dtype_note"] + ".  Intermediate z-steps of the outer loop run without the "
                        "sparsity guard (PB_FLAG_NO_RHO_GUARD; the closing z-step is guarded).  Normal equations, theta "
                        "fit, HRF model, step constant and the all-reduce: float64")
    out = {
        "metric": "voxel-iterations/sec", "value": value, "unit": "voxel-iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "untimed_spin_s": args.spin_seconds,
        **({"rehearsal": True} if args.rehearse_on_one_gpu else {}),
        "ms_per_step": ms_step, "wall_clock_to_eps_ms": ms_step, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, **dt,
        "data": "synthetic",
        "config": {"workload": "BASELINE config 4: semi-blind deconvolution with ONE shared HRF dilation, %d voxels "
                               "x %d scans (TR %.2f s, HRF %g s = %d taps, true dilation %.1f, start 2.0, SNR 10 dB), "
                               "lambda=%g, %d outer x %d inner iterations + closing z-step per step%s"
                               % (V_total, N, t_r, hrf_dur, K, theta_true, lbda, nb_outer, nb_inner,
                                  "" if world == 1 else ", %d GPUs (%s scaling)" % (world, args.scaling)),
                   "baseline_config": 4, "voxels_total": V_total, "voxels_this_rank": V, "scans": N, "taps": int(K),
                   "iters_per_step": n_z * nb_inner,
                   "kernel": dom_kernel + " reading ONE shared HRF and its step from device memory",
                   "launches_per_z_step": ([{"kernel": main_kernel, "problems": n_main}] if n_main else []) +
                                          [{"kernel": tail_kernel, "problems": V - n_main}],
                   "launches_per_outer_iteration": "z-step (whole rounds + remainder), normal equations (cumsum "
                                                   "and ||w||_1 folded in) + fixed-order reduce, theta fit (HRF, "
                                                   "its step constant and the cost folded in)",
                   "parallelism": "contiguous voxel shards x%d; ONE all-reduce (SUM) of %d float64 per outer "
                                  "iteration" % (world, K * K + K + 2)},
        "config4": {"end_to_end_ms": ms_step, "submission": ("one HIP graph per bd_shared call" if runner.graph is not None
                                                             else "eager loop (%s)" % runner.fallback),
                    "eager_loop_end_to_end_ms": eager_ms,
                    "z_steps_ms": z_ms, "theta_steps_and_glue_ms": ms_step - z_ms,
                    "z_step_fraction": z_ms / ms_step,
                    "theta_final": float(theta[-1]), "theta_true": theta_true,
                    "theta_trajectory": [round(float(t), 9) for t in theta],
                    "theta_trajectory_sha256_16": hashlib.sha256(np.round(theta, 9).tobytes()).hexdigest()[:16],
                    "cost_first_last": [float(d["J"][1]), float(d["J"][-1])]},
        "roofline": roofline_block(matrix_pipe, dom_kernel, dom_ms, V_dom, nb_inner, N, K, flops_launch, exec_launch,
                                   alg_bytes, None, mfma_block,
                                   {"whole_job_frac": (exec_launch if matrix_pipe else flops_launch) * (float(V) / V_dom) * n_z /
                                                      (step_ms * 1e-3) / 1e12 /
                                                      (MFMA_F16_PEAK_TFLOPS if matrix_pipe else VALU_FP32_PEAK_TFLOPS),
                                    "note_config4": "dominant kernel = one z-step launch (100 iterations, warm start read "
                                                    "from and written to HBM) on the whole rounds of the batch, incl. the "
                                                    "6 us re-solve launch; whole_job_frac prices the whole bd_shared call "
                                                    "(theta-steps, all-reduce and launch gaps included) the same way"}),
    }
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["parity"] = config4_parity(res["W"], res["h"], d, Y, t_r, hrf_dur, lbda, nb_outer, nb_inner, N, K,
                                       n_main, distributed, solver, torch, dev)
    return out


def config4_parity(W, h, d, Y, t_r, hrf_dur, lbda, nb_outer, nb_inner, N, K, n_main, distributed, solver, torch, dev):
    """Parity of the config-4 run THAT WAS TIMED (not of a small sub-batch on other kernels):

    (a) z-steps: 96 voxels drawn over the timed batch (64 of the matrix-pipe launch, 32 of the remainder
        when there is one) re-solved by the C float64 oracle through all 21 z-steps, each with the HRF of
        the GPU's own theta trajectory (so only the z-step arithmetic is compared: `fista_mfma_kernel<...,
        TAPS_DEV>` without the sparsity guard on the 20 intermediate solves, as timed): diff_z, z, x;
    (b) the whole loop: the first min(V, 17 000) voxels -- one matrix-pipe round + remainder, the same
        kernels and flags -- solved alone on the GPU and by an oracle loop that runs from its OWN state
        (C z-steps, NumPy normal equations + 1-D search): |dtheta| after every outer iteration and
        diff_z / z / x of all those voxels."""
    import numpy as np
    from oracle import c_oracle, pybold_oracle as orc
    from oracle.shared_ops import FastOracleOps
    V = Y.shape[0]
    rng = np.random.RandomState(0)
    if 0 < n_main < V:
        idx = np.sort(np.concatenate([rng.choice(n_main, size=min(64, n_main), replace=False),
                                      n_main + rng.choice(V - n_main, size=min(32, V - n_main), replace=False)]))
    else:
        idx = np.sort(rng.choice(V, size=min(96, V), replace=False))
    sel = torch.from_numpy(idx).to(dev)
    Ys = Y[sel].cpu().numpy().astype(np.float64)
    theta = np.asarray(d["theta"], dtype=np.float64)
    Wo = np.zeros((len(idx), N))
    for it in range(nb_outer + 1):                      # z-step `it` uses the HRF of theta[it] (bold_signal.py:320-335)
        hk = orc.spm_hrf(float(theta[it]), t_r, hrf_dur, False)[0]
        Wo, _, _ = c_oracle.fista_batch(Ys, hk, lbda, 1.0 / orc.gram_lipschitz(hk, N), nb_inner, W0=Wo, threads=0)

    def rel(a, b):
        return float((np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-300)).max())
    hl = orc.spm_hrf(float(theta[-1]), t_r, hrf_dur, False)[0]
    Xg, Zg = solver.fista_outputs(W[sel].contiguous(), hl)
    Zo = np.cumsum(Wo, axis=1)
    Xo = orc.causal_conv(hl, Zo)
    par = {"tolerance": 1e-5,
           "z_steps_as_timed": {
               "max_rel_l2_diff_z_vs_cpu_oracle": rel(W[sel].cpu().numpy(), Wo),
               "max_rel_l2_z_vs_cpu_oracle": rel(Zg.cpu().numpy(), Zo), "max_rel_l2_x_vs_cpu_oracle": rel(Xg.cpu().numpy(), Xo),
               "voxels_checked": int(len(idx)), "of_the_matrix_pipe_launch": int((idx < n_main).sum()) if n_main else 0,
               "sample": "voxels of the TIMED batch through all %d z-steps on the C float64 oracle, HRF of the GPU's theta "
                         "trajectory in each" % (nb_outer + 1)}}
    # (b) whole loop on one matrix-pipe round + remainder, against an oracle loop with its own state
    Vs = min(V, 17000)
    Ysub = Y[:Vs].contiguous()

    class _OneProcess:
        world_size, rank = 1, 0

        @staticmethod
        def allreduce_(t):
            return t
    Wg, hg, dg = distributed.bd_shared(Ysub, t_r, lbda=lbda, theta_0=2.0, hrf_dur=hrf_dur, nb_iter=nb_outer, nb_inner=nb_inner)
    Wo2, ho, do = distributed.bd_shared(Ysub.cpu(), t_r, lbda=lbda, theta_0=2.0, hrf_dur=hrf_dur, nb_iter=nb_outer,
                                        nb_inner=nb_inner, ops=FastOracleOps(N, t_r, hrf_dur), comm=_OneProcess())
    Wo2 = Wo2.numpy()
    Xg, Zg = solver.fista_outputs(Wg, hg)
    Zo = np.cumsum(Wo2, axis=1)
    Xo = orc.causal_conv(ho, Zo)
    sub_main, sub_kernel, _ = solver.launch_plan(N, K, Vs)
    par["whole_loop"] = {
        "voxels": Vs, "outer_iterations_checked": nb_outer, "kernel": sub_kernel, "problems_on_it": sub_main,
        "max_abs_dtheta_vs_cpu_oracle": float(np.abs(np.asarray(dg["theta"]) - np.asarray(do["theta"])).max()),
        "max_rel_l2_diff_z_vs_cpu_oracle": rel(Wg.cpu().numpy(), Wo2),
        "max_rel_l2_z_vs_cpu_oracle": rel(Zg.cpu().numpy(), Zo), "max_rel_l2_x_vs_cpu_oracle": rel(Xg.cpu().numpy(), Xo),
        "sample": "the first %d voxels solved alone (same kernels and flags as the timed run), whole loop (z-steps, normal "
                  "equations, theta fits) on the GPU vs an oracle loop running from its own state" % Vs}
    par["max_rel_l2_diff_z_vs_cpu_oracle"] = max(par["z_steps_as_timed"]["max_rel_l2_diff_z_vs_cpu_oracle"],
                                                 par["whole_loop"]["max_rel_l2_diff_z_vs_cpu_oracle"])
    par["max_abs_dtheta_vs_cpu_oracle"] = par["whole_loop"]["max_abs_dtheta_vs_cpu_oracle"]
    return par


def pcie_inclusive(plan, Y, hrf, lbda, step, n_iter, solver, torch, dev, pipelined):
    """Host-buffer variants of the boundary (never the headline value): y starts in pinned
    host memory; (a) results stay in HBM, (b) diff_z also returns to the host as float32.
    Default: copy, solve, copy one after the other (launches of the same size as the timed
    ones, so that a rocprofv3 --stats of the default command still averages ONE launch shape
    of the dominant kernel).  `pipelined` (--extras): also chunked over three streams
    (solver.HostPipeline: the copies of the neighbouring chunks overlap the solve of the
    current one) as `..._pipelined`."""
    Yh = Y.cpu().pin_memory()
    res = {}

    def clock(fn):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:      # the device is back at idle clocks after the allocations
            fn()
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / 5 * 1e3

    # every buffer first, every measurement after (freeing pinned memory between them stalls)
    pipes = {}
    if pipelined:
        pipes = {key: solver.HostPipeline(Y.shape[0], Y.shape[1], hrf, lbda, step, n_iter, out_dtype=od, dev=dev)
                 for key, od in (("wall_clock_to_eps_ms_incl_h2d_pipelined", None),
                                 ("wall_clock_to_eps_ms_incl_h2d_d2h_f32_pipelined", torch.float32))}
    Yd = torch.empty_like(Y)
    planp = solver.FistaPlan(Yd, hrf, lbda, step, n_iter, force=None)
    Wh = torch.empty(planp.W.shape, dtype=torch.float32).pin_memory()

    def h2d_solve():
        Yd.copy_(Yh, non_blocking=True)
        planp.run()

    def h2d_solve_d2h():
        h2d_solve()
        Wh.copy_(planp.W.float(), non_blocking=True)

    res["wall_clock_to_eps_ms_incl_h2d"] = clock(h2d_solve)
    res["wall_clock_to_eps_ms_incl_h2d_d2h_f32"] = clock(h2d_solve_d2h)
    for key, pipe in pipes.items():
        res[key] = clock(lambda: pipe.run(Yh))
        res["host_pipeline_chunk_voxels"] = pipe.chunk
    del pipes
    del Yh, Yd, Wh, planp
    return res


def cpu_topology():
    info = {"cpu_count": os.cpu_count()}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        info["affinity"] = info["cpu_count"]
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        info["cgroup_cpu_quota"] = None if quota == "max" else float(quota) / float(period)
    except Exception:
        info["cgroup_cpu_quota"] = None
    usable = info["affinity"]
    if info["cgroup_cpu_quota"]:
        usable = max(1, min(usable, int(info["cgroup_cpu_quota"] + 0.5)))
    info["usable_cores"] = usable
    return info


def cpu_dense_baseline(n_scans, n_iter, lbda, target_s):
    """The reference's own formulation (dense N x N Toeplitz mat-vecs in float64 NumPy,
    pybold/linear.py:73-113; no cost trace, no print) fanned out over the host cores like
    the reference's joblib axis (examples/icassp_2019/simulation.py:62-72): one worker
    process per usable core, BLAS threads pinned to 1 as examples/synth_data/deconv.py:11-14
    does.  Runs before the GPU is initialised (oracle/dense_worker.py, plain NumPy)."""
    topo = cpu_topology()
    workers = int(os.environ.get("PYBOLD_BENCH_CPU_WORKERS", topo["usable_cores"]))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))

    def fan(n_vox):
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.dense_worker", str(n_vox),
                                   str(n_scans), str(n_iter), str(lbda), str(w)],
                                  env=env, cwd=ROOT, stdout=subprocess.PIPE) for w in range(workers)]
        inner = [float(p.communicate()[0].decode().strip().split()[-1]) for p in procs]
        if any(p.returncode != 0 for p in procs):
            raise RuntimeError("oracle.dense_worker failed")
        return time.perf_counter() - t0, max(inner)

    _, t_cal = fan(2)                                     # calibrate: 2 voxels per worker
    per_voxel = max(t_cal / 2.0, 1e-6)
    n_vox = int(max(2, min(256, target_s / per_voxel)))
    wall, inner = fan(n_vox)
    return {"value": workers * n_vox * n_iter / inner, "workers": workers, "voxels_per_worker": n_vox,
            "seconds": inner, "seconds_incl_process_start": wall, "topology": topo}


def cpu_port_and_parity(Y, W_gpu, hrf, lam, y_rep, step, n_iter, target_s, dense, n_main):
    """The matrix-free C/OpenMP float64 port (oracle/fista_oracle.c) on the usable host cores, timed
    on a bounded sample of the same workload; its output is also the parity check of the GPU
    result.  The sample is drawn at RANDOM over the whole batch and always holds >= 64 problems
    of the remainder launch (problems >= n_main, another kernel form) and >= 64 of the second
    half of the main launch."""
    import numpy as np
    import torch
    from oracle import c_oracle
    topo = dense["topology"]
    cores = int(os.environ.get("PYBOLD_BENCH_CPU_THREADS", topo["usable_cores"]))
    P = W_gpu.shape[0]
    rng = np.random.RandomState(12345)

    def rows(idx):
        idx = np.asarray(idx)
        Yh = Y[torch.from_numpy(idx // y_rep).to(Y.device)].cpu().numpy().astype(np.float64)
        lv = lam[torch.from_numpy(idx).to(lam.device)].cpu().numpy() if torch.is_tensor(lam) else lam
        return Yh, lv

    cal = rng.choice(P, size=min(P, 2 * cores), replace=False)
    Yc, lc = rows(cal)
    t0 = time.perf_counter()
    c_oracle.fista_batch(Yc, hrf, lc, step, n_iter, threads=cores)          # calibrate
    per_problem = (time.perf_counter() - t0) / len(cal)
    n_sample = int(min(P, max(2 * cores, target_s / max(per_problem, 1e-9))))
    parts = []
    if 0 < n_main < P:
        parts.append(n_main + rng.choice(P - n_main, size=min(64, P - n_main), replace=False))
        parts.append(n_main // 2 + rng.choice(n_main - n_main // 2, size=min(64, n_main - n_main // 2), replace=False))
    else:
        parts.append(P // 2 + rng.choice(P - P // 2, size=min(64, P - P // 2), replace=False))
    forced = np.unique(np.concatenate(parts))
    rest = np.setdiff1d(rng.choice(P, size=min(P, n_sample), replace=False), forced)[:max(0, n_sample - len(forced))]
    idx = np.sort(np.concatenate([forced, rest]))
    Yh, lv = rows(idx)
    t0 = time.perf_counter()
    Wc, _, used = c_oracle.fista_batch(Yh, hrf, lv, step, n_iter, threads=cores)
    dt = time.perf_counter() - t0
    from pybold_amd import solver
    from oracle import pybold_oracle as orc
    Wsel = W_gpu[torch.from_numpy(idx).to(W_gpu.device)].contiguous()
    Xg, Zg = solver.fista_outputs(Wsel, hrf)             # z = cumsum(diff_z), x = hrf * z (bold_signal.py:74-75)
    Wg, Xg, Zg = Wsel.cpu().numpy(), Xg.cpu().numpy(), Zg.cpu().numpy()
    Zc = np.cumsum(Wc, axis=1)
    Xc = orc.causal_conv(np.asarray(hrf, dtype=np.float64), Zc)
    nrm = np.linalg.norm(Wc, axis=1)
    ok = nrm > 0                      # lambda = lambda_max: the solution is exactly 0 on both sides
    err = float((np.linalg.norm(Wg - Wc, axis=1)[ok] / nrm[ok]).max()) if ok.any() else 0.0
    err_z = float((np.linalg.norm(Zg - Zc, axis=1)[ok] / np.linalg.norm(Zc, axis=1)[ok]).max()) if ok.any() else 0.0
    err_x = float((np.linalg.norm(Xg - Xc, axis=1)[ok] / np.linalg.norm(Xc, axis=1)[ok]).max()) if ok.any() else 0.0
    zero_ok = bool(np.abs(Wg[~ok]).max() == 0.0) if (~ok).any() else True
    port = {"port_value": len(idx) * n_iter / dt, "port_threads": int(used),
            "port_sample": "%d problems drawn at random over the batch x %d iterations, C/OpenMP float64 "
                           "matrix-free port (oracle/fista_oracle.c), %d threads, %.1f s"
                           % (len(idx), n_iter, int(used), dt)}
    parity = {"max_rel_l2_diff_z_vs_cpu_oracle": err, "max_rel_l2_z_vs_cpu_oracle": err_z,
              "max_rel_l2_x_vs_cpu_oracle": err_x, "voxels_checked": int(len(idx)), "tolerance": 1e-5,
              "sample": "random over the whole batch; %d of the remainder launch, %d of the second half of the "
                        "main launch" % (int((idx >= n_main).sum()) if 0 < n_main < P else 0,
                                         int(((idx >= n_main // 2) & (idx < n_main)).sum()) if 0 < n_main < P
                                         else int((idx >= P // 2).sum())),
              "all_zero_solutions_match": zero_ok}
    return port, parity


def cpu_baseline_block(dense, port, n_iter):
    """`value`: the reference's dense-Toeplitz NumPy formulation on all usable host cores (see
    cpu_dense_baseline).  `port_value` (when measured): the matrix-free C/OpenMP port."""
    topo = dense["topology"]
    base = {"value": dense["value"], "unit": "voxel-iterations/s", "cores": dense["workers"],
            "kind": "port",
            "sample": "%d worker processes x %d synthetic voxels x %d iterations of the reference's "
                      "dense-Toeplitz float64 NumPy formulation (oracle.pybold_oracle.fista_batch, "
                      "dense=True), BLAS threads = 1, %.1f s"
                      % (dense["workers"], dense["voxels_per_worker"], n_iter, dense["seconds"]),
            "cpu_count": topo["cpu_count"], "workers": dense["workers"],
            "affinity": topo["affinity"], "cgroup_cpu_quota": topo["cgroup_cpu_quota"]}
    if port:
        base.update(port)
    return base


if __name__ == "__main__":
    main()
