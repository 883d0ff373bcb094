"""Round-5 golden fixtures (``auto_lbda.npz``): the reference's ``deconv(lbda=None)`` branch
(pybold/bold_signal.py:99-214) run by the REAL reference, with the one thing it needs from the
absent PyWavelets -- the scalar ``sigma = mad_daub_noise_est(y)`` (:103, imported by name at
:10) -- injected after import:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden_r5.py

Same import recipe as ``make_golden.py`` (build container only; data only, no reference source).
That pins everything of the branch except the noise estimate itself: the alpha / lambda updates
(:141-145), the warm-started inner solves with the momentum restart (:110-138), both stop windows
(:125-138, :164-178), the closing solve (:181-209) and the ``(x, z, diff_z, J, R, G)`` lists.

Runs: small budgets ``nb_iter`` in {1, 2, 5, 20} x ``nb_sub_iter`` in {1, 10, 50} with the window rules
on and off (at ``tol = 1e-6`` neither rule can fire within such budgets: on == off, which is itself
the pin), runs with ``tol`` in {1e-2, 1e-3} and ``wind`` in {4, 6} where BOTH rules fire, and the
reference's DEFAULT call (1000 x 1000 iterations, ``tol = 1e-6``, ``wind = 6``).

Keys, per run tag ``<case>_s<sigma index>_o<nb_iter>_i<nb_sub_iter>_e<0|1>[_t<tol>_w<wind>]`` (and ``…_default``):
  kw_                          [nb_iter, nb_sub_iter, early_stopping, tol, wind] of the call
  x_, z_, dz_, J_, R_, G_      the reference's outputs (lists as arrays)
  alpha_                       alpha after every outer iteration, recomputed from R with the
                               reference's own expression (``alpha += mu * (R_i - N sigma^2)``:
                               ``grad`` and ``r`` are the same sum at :143 and :153)
plus per case ``<case>_y``, ``<case>_hrf``, ``<case>_lipschitz``, ``<case>_x0`` (the start vector
``spectral_radius_est`` drew, :52) and ``<case>_sigma`` (the three injected values: 0.5x, 1x, 2x
the in-package db3 MAD estimate of that series, rounded to 6 digits so that the fixture does not
depend on that estimate's last bits).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden import _import_reference, quiet  # noqa: E402


def main():
    bs, cv, data, hm, lin, ut = _import_reference()
    from oracle import pybold_oracle as orc        # only for the size of sigma (a number we CHOOSE)
    out = {}
    budgets = [(o, i, e) for o in (1, 2, 5, 20) for i in (1, 10, 50) for e in (0, 1)]
    for case, delta in (("c1", 1.0), ("c2", 1.5)):
        hrf = hm.spm_hrf(delta, t_r=1.0, dur=30.)[0]
        y = data.gen_regular_bloc_bold(dur=5, tr=1.0, hrf=hrf, snr=1.0, random_state=0)[0]
        n = len(y)
        np.random.seed(0)
        x0 = np.random.randn(n)
        np.random.seed(0)
        H = lin.ConvAndLinear(lin.DiscretInteg(), hrf, dim_in=n, dim_out=n)
        lip = 0.9 * ut.spectral_radius_est(H, (n,))
        s_hat = float(orc.mad_daub_noise_est(y))
        sigmas = np.array([float("%.6g" % (f * s_hat)) for f in (0.5, 1.0, 2.0)])
        out.update({case + "_y": y, case + "_hrf": hrf, case + "_lipschitz": lip, case + "_x0": x0,
                    case + "_sigma": sigmas})

        def run(tag, sigma, **kw):
            bs.mad_daub_noise_est = lambda x: sigma          # bold_signal.py:10 bound the name at import
            np.random.seed(0)
            t0 = time.time()
            x, z, dz, J, R, G = quiet(bs.deconv, y, 1.0, hrf, lbda=None, **kw)
            alpha, a = [], 1.0
            for r in R:
                a += 1.0e-4 * (r - n * sigma ** 2)
                alpha.append(a)
            full = dict(nb_iter=1000, nb_sub_iter=1000, early_stopping=True, tol=1.0e-6, wind=6)   # :13-14
            full.update(kw)
            out["kw_" + tag] = np.array([full["nb_iter"], full["nb_sub_iter"], float(full["early_stopping"]),
                                         full["tol"], full["wind"]])
            out.update({"x_" + tag: x, "z_" + tag: z, "dz_" + tag: dz, "J_" + tag: np.array(J),
                        "R_" + tag: np.array(R), "G_" + tag: np.array(G), "alpha_" + tag: np.array(alpha)})
            print("%-28s outer %4d  lbda_end %.6g  |dz| %.6g  %.1f s" % (
                tag, len(J), 1.0 / (2.0 * alpha[-1]), np.linalg.norm(dz), time.time() - t0), flush=True)

        real = bs.mad_daub_noise_est
        try:
            for si, sigma in enumerate(sigmas):
                for o, i, e in budgets:
                    if case == "c2" and (si != 1 or i == 1):   # case 2: the middle sigma, fewer budgets
                        continue
                    run("%s_s%d_o%d_i%d_e%d" % (case, si, o, i, e), float(sigma), nb_iter=o, nb_sub_iter=i,
                        early_stopping=bool(e))
            # both window rules firing: looser tolerances, two window lengths
            for si in ((0, 1, 2) if case == "c1" else (1,)):
                for tol, wind in ((1.0e-2, 6), (1.0e-3, 6), (1.0e-2, 4), (1.0e-3, 4)):
                    run("%s_s%d_o60_i300_e1_t%g_w%d" % (case, si, tol, wind), float(sigmas[si]), nb_iter=60,
                        nb_sub_iter=300, early_stopping=True, tol=tol, wind=wind)
            # the reference's DEFAULT call: nb_iter = nb_sub_iter = 1000, window rule on, tol 1e-6
            for si in ((0, 1, 2) if case == "c1" else (1,)):
                run("%s_s%d_default" % (case, si), float(sigmas[si]))
        finally:
            bs.mad_daub_noise_est = real
    np.savez_compressed(os.path.join(HERE, "auto_lbda.npz"), **out)
    print("auto_lbda.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
