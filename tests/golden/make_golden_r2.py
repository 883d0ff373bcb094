"""Round-2 golden fixtures (``round2.npz``), generated from the REAL reference exactly like
``make_golden.py`` (same in-process stand-ins for the two absent third-party imports; run
in the build container only):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden_r2.py

Kept in a file of its own so that the round-1 fixtures stay byte-identical.  Data only.

  bd_theta          the dilation after each outer iteration of the golden ``bd`` run of
                    bd.npz (recorded by wrapping the optimiser the reference calls at
                    pybold/bold_signal.py:330-333), and its cost trace again as a check
  bdw_*             ``bd`` warm-started from a block signal ``z_0`` and ``theta_0 = 1.0``
                    (:291-301), 3 outer iterations
  reg_*             ``gen_regular_bloc_bold`` (pybold/data.py:10-41) for two settings
  inf_*             ``inf_norm`` (pybold/utils.py:112-138): 1-D, 2-D (both axes), 3-D, list
  fit_theta/fit_err ``hrf_fit_err`` on a fine theta grid around its minimum (the objective
                    the device theta-step minimises), and ``hrf_estim``'s minimiser
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _import_reference, quiet  # noqa: E402


def main():
    bs, cv, data, hm, lin, ut = _import_reference()
    out = {}
    t_r, hrf_dur = 0.75, 20.0
    h_true = hm.spm_hrf(0.7, t_r, hrf_dur, False)[0]
    gen = data.gen_regular_bloc_bold(dur=3, tr=t_r, hrf=h_true, snr=10.0, random_state=0)
    yb, z_true = gen[0], gen[2]

    # ---- bd: theta after each outer iteration ------------------------------------------
    thetas = []
    real = bs.fmin_l_bfgs_b

    def recording(*a, **k):
        res = real(*a, **k)
        thetas.append(float(np.ravel(res[0])[0]))
        return res
    bs.fmin_l_bfgs_b = recording
    try:
        x, z, dz, h, d = bs.bd(yb, t_r, lbda=1.7, hrf_dur=hrf_dur, nb_iter=5)
        out["bd_theta"] = np.array(thetas)
        out["bd_J"] = d["J"]
        out["bd_h"] = h
        del thetas[:]
        x, z, dz, h, d = bs.bd(yb, t_r, lbda=1.7, theta_0=1.0, z_0=z_true, hrf_dur=hrf_dur,
                               nb_iter=3)
        out.update(bdw_y=yb, bdw_z0=z_true, bdw_theta=np.array(thetas), bdw_x=x, bdw_z=z,
                   bdw_diff_z=dz, bdw_h=h, bdw_J=d["J"], bdw_r=d["r"], bdw_g=d["g"])
    finally:
        bs.fmin_l_bfgs_b = real
    print("bd thetas", out["bd_theta"], "warm", out["bdw_theta"])

    # ---- regular-block generator ------------------------------------------------------------
    for tag, kw in (("a", dict(dur=3, tr=0.75, dur_bloc=30.0, snr=10.0, random_state=0)),
                    ("b", dict(dur=5, tr=1.0, dur_bloc=20.0, snr=1.0, random_state=3))):
        hrf = hm.spm_hrf(1.0, t_r=kw["tr"], dur=20.0, normalized_hrf=False)[0]
        noisy, ar_s, ai_s, i_s, t, _, noise = data.gen_regular_bloc_bold(hrf=hrf, **kw)
        out.update({"reg_%s_p" % tag: np.array([kw["dur"], kw["tr"], kw["dur_bloc"], kw["snr"],
                                                kw["random_state"]]),
                    "reg_%s_hrf" % tag: hrf, "reg_%s_noisy" % tag: noisy, "reg_%s_clean" % tag: ar_s,
                    "reg_%s_ai_s" % tag: ai_s, "reg_%s_i_s" % tag: i_s, "reg_%s_noise" % tag: noise})

    # ---- inf_norm -----------------------------------------------------------------------------
    rng = np.random.RandomState(11)
    a1, a2, a3 = rng.randn(300) * 4.0, rng.randn(7, 240) * 2.0, rng.randn(3, 4, 50)
    lst = ut.inf_norm([a1, a2, a3])
    out.update(inf_a1=a1, inf_a2=a2, inf_a3=a3, inf_o1=ut.inf_norm(a1), inf_o2=ut.inf_norm(a2),
               inf_o2_axis0=ut.inf_norm(a2, axis=0), inf_o3=ut.inf_norm(a3), inf_l0=lst[0],
               inf_l1=lst[1], inf_l2=lst[2])

    # ---- hrf_fit_err around its minimum, hrf_estim's theta ----------------------------------
    h_est, _ = bs.hrf_estim(z_true, yb, t_r, hrf_dur)
    grid = np.linspace(0.6, 1.9, 131)
    errs = np.array([bs.hrf_fit_err(t, z_true, yb, t_r, hrf_dur) for t in grid])
    out.update(fit_z=z_true, fit_y=yb, fit_theta=grid, fit_err=errs, fit_h_estim=h_est)
    np.savez_compressed(os.path.join(HERE, "round2.npz"), **out)
    print("round2.npz:", len(out), "arrays; argmin of hrf_fit_err on the grid:", grid[np.argmin(errs)])


if __name__ == "__main__":
    main()
