"""Round-5 golden fixture ``long_series.npz``: the REAL reference's fixed-lambda ``deconv`` (pybold/bold_signal.py:49-97) on series
of HCP length -- 1 200 scans at TR 0.72 s with a 20 s HRF (28 taps), the shape of examples/icassp_2019/validation.py:41-48 -- and
of 900 scans with a 30 s HRF (42 taps: three near tiles on the split matrix-pipe forms), so that the four-wave kernel is compared
with the reference itself and not only with its restatement:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden_r5_long.py

Same import recipe as ``make_golden.py`` (build container only; data only, no reference source).  Keys per case ``<c>``:
``<c>_y``, ``<c>_hrf``, ``<c>_t_r``, ``<c>_lipschitz`` (0.9 x the reference's spectral-radius estimate, seed 0, :52), and per run
``<c>_l<lambda>_n<nb_iter>[_es]``: ``x_``, ``z_``, ``dz_``, ``J_`` (``_es``: early stopping on, tol 1e-2: the window rule fires)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _import_reference, quiet  # noqa: E402


def main():
    bs, cv, data, hm, lin, ut = _import_reference()
    out = {}
    for case, n, t_r, dur in (("hcp", 1200, 0.72, 20.0), ("long42", 900, 0.72, 30.0)):
        hrf = hm.spm_hrf(1.0, t_r=t_r, dur=dur)[0]
        y = data.gen_regular_bloc_bold(dur=n * t_r / 60.0 + 1.0, tr=t_r, hrf=hrf, snr=1.0, random_state=1)[0][:n]
        assert len(y) == n, len(y)
        np.random.seed(0)
        H = lin.ConvAndLinear(lin.DiscretInteg(), hrf, dim_in=n, dim_out=n)
        lip = 0.9 * ut.spectral_radius_est(H, (n,))
        out.update({case + "_y": y, case + "_hrf": hrf, case + "_t_r": t_r, case + "_lipschitz": lip})
        for lbda in (0.5, 2.0):
            for nb_iter, es in ((100, False), (400, True)):
                np.random.seed(0)
                x, z, dz, J, _, _ = quiet(bs.deconv, y, t_r, hrf, lbda=lbda, nb_iter=nb_iter, early_stopping=es, tol=1.0e-2, wind=6)
                tag = "%s_l%g_n%d%s" % (case, lbda, nb_iter, "_es" if es else "")
                out.update({"x_" + tag: x, "z_" + tag: z, "dz_" + tag: dz, "J_" + tag: np.asarray(J)})
                print(tag, "K =", len(hrf), "iterations run:", len(J), "||diff_z|| = %.6g" % np.linalg.norm(dz))
    np.savez_compressed(os.path.join(HERE, "long_series.npz"), **out)
    print("wrote long_series.npz: %d arrays" % len(out))


if __name__ == "__main__":
    main()
