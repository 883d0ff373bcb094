"""Adversarial-data sweep of the matrix-pipe kernel's guards through the DEFAULT dispatch (round 4).

Every call holds >= one matrix-pipe round (16 384 problems + a remainder on the vector forms) of
series unlike the generator's: DC baselines 10x / 100x / 1000x the fluctuation (raw fMRI: mean >>
fluctuation), Student-t noise, SNR -10 ... +30 dB, constant / single-spike / all-zero series;
lambda / lambda_max in logspace(-3, 0); N in {129, 160, 300, 320}, K in {1, 2, 30, 33, 48}.
For a random sample of every family: relative L2 error of diff_z, z and x (and diff_z[1:] for the DC
families) against the C float64 oracle, split into problems the matrix-pipe kernel kept and problems
its guards handed back to the float32 operators; hand-back rate per family.

usage: python tools/r4_adversarial_sweep.py [n_iter] [sample_per_family] > profiles/r4_adversarial_sweep.txt
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import c_oracle, pybold_oracle as orc          # noqa: E402
from pybold_amd import solver                              # noqa: E402

from tests.adversarial_data import DC_FAMILIES, FAMILIES, P_FAMILY, hrf_for, make_batch   # noqa: E402


def rel(a, b):
    return np.linalg.norm(a - b, axis=1) / (np.linalg.norm(b, axis=1) + 1e-300)


def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    n_s = int(sys.argv[2]) if len(sys.argv) > 2 else 192
    dev = torch.device("cuda")
    shapes = [(300, 30), (129, 1), (160, 2), (320, 33), (300, 48)]
    cs = [1e-3, 1e-2, 1e-1, 1.0]
    print("# adversarial sweep, default dispatch, %d iterations, %d problems per call (%d per family), oracle sample %d per family"
          % (n_iter, len(FAMILIES) * P_FAMILY, P_FAMILY, n_s))
    print("# columns: max relative L2 error vs the C float64 oracle of diff_z | z | x | diff_z[1:] over the sampled problems the")
    print("# matrix-pipe kernel KEPT (k) and over those its guards HANDED BACK to the float32 operators (h); hb = hand-back rate")
    worst_kept = worst_all = 0.0
    t0 = time.time()
    for (N, K) in shapes:
        hrf = hrf_for(K)
        A = orc.toeplitz_from_kernel(hrf, N, N).dot(np.tril(np.ones((N, N))))
        step = 1.0 / (0.9 * np.linalg.norm(A, 2) ** 2)
        Y, fam = make_batch(N, hrf, 100 + N + K, dev)
        P = Y.shape[0]
        plan = solver.launch_plan(N, K, P)
        print("\n## N = %d, K = %d: %d problems, plan %s" % (N, K, P, plan))
        rng = np.random.RandomState(N * 100 + K)
        samp = np.sort(np.concatenate([rng.choice(np.nonzero(fam == f)[0], n_s, replace=False) for f in range(len(FAMILIES))]))
        sel = torch.from_numpy(samp).to(dev)
        Ys = Y[sel].cpu().numpy().astype(np.float64)
        for c in cs:
            W, _, nd = solver.fista_solve(Y, hrf, c, step, n_iter)
            assert bool(torch.isfinite(W).all()) and int(nd.min()) == n_iter
            _, _, nd0 = solver.fista_solve(Y, hrf, c, step, n_iter, force="noresolve")
            back = (nd0 < 0).cpu().numpy()
            Wo, _, _ = c_oracle.fista_batch(Ys, hrf, c, step, n_iter, threads=0)
            Xg, Zg = solver.fista_outputs(W[sel].contiguous(), hrf)
            Wg, Xg, Zg = W[sel].cpu().numpy(), Xg.cpu().numpy(), Zg.cpu().numpy()
            Zo = np.cumsum(Wo, axis=1)
            Xo = orc.causal_conv(hrf, Zo)
            zero = np.linalg.norm(Wo, axis=1) == 0
            assert (np.abs(Wg[zero]).max() if zero.any() else 0.0) == 0.0      # an all-zero solution is all-zero on the GPU
            e = {"dz": rel(Wg, Wo), "z": rel(Zg, Zo), "x": rel(Xg, Xo), "dz1": rel(Wg[:, 1:], Wo[:, 1:])}
            for k in e:
                e[k][zero] = 0.0
            print("  lambda/lambda_max = %g   (hand-back over the whole call: %.1f %%)" % (c, 100.0 * back.mean()))
            for f, name in enumerate(FAMILIES):
                m = fam[samp] == f
                hb = back[samp][m]
                def mx(key, mask):
                    return e[key][m][mask].max() if mask.any() else float("nan")
                cols = ["dz", "z", "x"] + (["dz1"] if f in DC_FAMILIES else [])
                print("    %-32s hb %5.1f %%  k: %s   h: %s" % (
                    name, 100.0 * back[fam == f].mean(),
                    " ".join("%s %.1e" % (k, mx(k, ~hb)) for k in cols), " ".join("%s %.1e" % (k, mx(k, hb)) for k in cols)))
                for k in cols:
                    if (~hb).any():
                        worst_kept = max(worst_kept, float(np.nanmax(e[k][m][~hb])))
                    worst_all = max(worst_all, float(np.nanmax(e[k][m])))
            sys.stdout.flush()
    print("\n# worst error over everything the matrix-pipe kernel kept: %.2e; over all problems: %.2e   (%.0f s)"
          % (worst_kept, worst_all, time.time() - t0))


if __name__ == "__main__":
    main()
