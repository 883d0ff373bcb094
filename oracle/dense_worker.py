"""One worker of bench.py's `cpu_baseline`: the reference's dense-Toeplitz formulation of
the fixed-lambda loop (pybold/bold_signal.py:49-72 over pybold/linear.py:73-113: two dense
N x N float64 mat-vecs and two cumulative sums per iteration), one voxel at a time like the
reference's joblib tasks (examples/icassp_2019/simulation.py:62-72).  Plain NumPy, never
touches the GPU.  TEST INFRASTRUCTURE / CPU BASELINE ONLY.

    python -m oracle.dense_worker <n_voxels> <n_scans> <n_iter> <lbda> <seed>

prints the seconds spent in the solves (data generation and operator set-up excluded, which
favours the CPU)."""
import sys
import time

import numpy as np

from oracle import pybold_oracle as orc


def main():
    n_vox, n, n_iter, lbda, seed = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]),
                                    float(sys.argv[4]), int(sys.argv[5]))
    hrf = orc.spm_hrf(1.0, t_r=1.0, dur=30.0)[0]
    rng = np.random.RandomState(seed)
    clean = orc.gen_regular_bloc_bold(dur=(n + 0.5) / 60.0, tr=1.0, dur_bloc=30.0, hrf=hrf)[1]
    Y = np.stack([orc.gen_regular_bloc_bold(dur=(n + 0.5) / 60.0, tr=1.0, dur_bloc=30.0, hrf=hrf, snr=1.0,
                                            noise=rng.randn(n))[0] for _ in range(n_vox)])
    assert Y.shape == (n_vox, n) and len(clean) == n
    H = orc.DenseH(hrf, n, n)
    step = 1.0 / (0.9 * orc.spectral_radius_est(H, rng.randn(n)))
    t0 = time.perf_counter()
    for y in Y:
        orc.fista_batch(y[None], hrf, lbda, step, n_iter, dense=True)
    print("%.6f" % (time.perf_counter() - t0))


if __name__ == "__main__":
    main()
