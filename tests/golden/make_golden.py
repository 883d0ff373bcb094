"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (the reference never travels to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden.py

The reference checkout at /root/reference is imported read-only.  Two of its
third-party imports are absent from the image (``numba`` at
pybold/bold_signal.py:5, ``pywt`` at pybold/utils.py:7) and ``np.float``
(pybold/data.py:441) was removed from NumPy; they are satisfied in-process
below exactly as SURVEY.md §8(c) records:  ``numba.jit`` becomes the identity
decorator (so ``_loops_deconv`` runs its body as NumPy) and ``pywt`` is an
empty module (so ``deconv(lbda=None)`` cannot be run and is not captured).

Outputs: ``*.npz`` files holding inputs and the reference's outputs.  Data
only: no reference source text is stored.
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("PYBOLD_REFERENCE", "/root/reference")


def _import_reference():
    sys.dont_write_bytecode = True
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)

    class _T:
        def __getitem__(self, k):
            return self
    for name in ("float64", "int64", "boolean"):
        setattr(nb, name, _T())
    sys.modules.setdefault("numba", nb)
    sys.modules.setdefault("pywt", types.ModuleType("pywt"))
    if not hasattr(np, "float"):
        np.float = float
    sys.path.insert(0, REF)
    import pybold  # noqa: F401
    from pybold import bold_signal, convolution, data, hrf_model, linear, utils
    return bold_signal, convolution, data, hrf_model, linear, utils


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    bs, cv, data, hm, lin, ut = _import_reference()
    out = {}

    # ---- case 1 / 2: deconv fixed lambda, 500 it, delta = 1.0 and 1.5 -----
    def lipschitz_of(y, hrf, seed):
        np.random.seed(seed)
        x0 = np.random.randn(len(y))
        np.random.seed(seed)
        H = lin.ConvAndLinear(lin.DiscretInteg(), hrf, dim_in=len(y),
                              dim_out=len(y))
        rho = ut.spectral_radius_est(H, (len(y),))
        return x0, 0.9 * rho

    for name, delta in (("case1", 1.0), ("case2", 1.5)):
        hrf = hm.spm_hrf(delta, t_r=1.0, dur=30.)[0]
        y = data.gen_regular_bloc_bold(dur=5, tr=1.0, hrf=hrf, snr=1.0,
                                       random_state=0)[0]
        x0, lip = lipschitz_of(y, hrf, 0)
        np.random.seed(0)
        x, z, dz, J, R, G = quiet(bs.deconv, y, 1.0, hrf, lbda=1.0,
                                  nb_iter=500, early_stopping=False)
        assert R is None and G is None
        np.savez(os.path.join(HERE, name + ".npz"), hrf=hrf, y=y, x0=x0,
                 lipschitz=lip, lbda=1.0, nb_iter=500, x=x, z=z, diff_z=dz,
                 J=J)
        print(name, lip, np.linalg.norm(dz), J[-1])

    # ---- case 3: grid lambda x seed x iterations --------------------------
    hrf = hm.spm_hrf(1.0, t_r=1.0, dur=30.)[0]
    grid = {"hrf": hrf}
    for seed in range(4):
        y = data.gen_regular_bloc_bold(dur=5, tr=1.0, hrf=hrf, snr=1.0,
                                       random_state=seed)[0]
        x0, lip = lipschitz_of(y, hrf, seed)
        grid["y_s%d" % seed] = y
        grid["x0_s%d" % seed] = x0
        grid["lip_s%d" % seed] = lip
        for lbda in (0.1, 1.0, 10.0):
            for nit in (1, 2, 3, 10, 500):
                np.random.seed(seed)
                x, z, dz, J, _, _ = quiet(bs.deconv, y, 1.0, hrf, lbda=lbda,
                                          nb_iter=nit, early_stopping=False)
                key = "s%d_l%g_n%d" % (seed, lbda, nit)
                grid["dz_" + key] = dz
                grid["J_" + key] = J
    np.savez_compressed(os.path.join(HERE, "grid.npz"), **grid)
    print("grid", len(grid))

    # ---- case 4: early stopping ------------------------------------------
    y = data.gen_regular_bloc_bold(dur=5, tr=1.0, hrf=hrf, snr=1.0,
                                   random_state=0)[0]
    x0, lip = lipschitz_of(y, hrf, 0)
    es = {"hrf": hrf, "y": y, "x0": x0, "lipschitz": lip}
    for tol in (0.1, 0.03, 0.01, 0.005):
        np.random.seed(0)
        x, z, dz, J, _, _ = quiet(bs.deconv, y, 1.0, hrf, lbda=1.0,
                                  nb_iter=1000, early_stopping=True, tol=tol,
                                  wind=6)
        es["n_%g" % tol] = len(J)
        es["dz_%g" % tol] = dz
        es["x_%g" % tol] = x
        print("early stop tol", tol, "->", len(J))
    np.random.seed(0)
    x, z, dz, J, _, _ = quiet(bs.deconv, y, 1.0, hrf, lbda=1.0)  # defaults
    es["n_default"] = len(J)
    es["dz_default"] = dz
    np.savez_compressed(os.path.join(HERE, "early_stop.npz"), **es)

    # ---- case 5: _loops_deconv (numba body executed as NumPy) ------------
    t_r, hrf_dur = 0.75, 20.0
    h_true = hm.spm_hrf(0.7, t_r, hrf_dur, False)[0]
    yb = data.gen_regular_bloc_bold(dur=3, tr=t_r, hrf=h_true, snr=10.0,
                                    random_state=0)[0]
    h0 = hm.spm_hrf(2.0, t_r, hrf_dur, False)[0]
    Hd = cv.toeplitz_from_kernel(h0, dim_in=len(yb), dim_out=len(yb))
    ld = {"y": yb, "h": h0, "h_true": h_true, "t_r": t_r, "hrf_dur": hrf_dur}
    for nit in (1, 2, 5, 100):
        for es_on, tol in ((False, 1e-12), (True, 1e-2), (True, 1e-3)):
            w = bs._loops_deconv(yb.copy(), np.zeros_like(yb), Hd, 1.7, nit,
                                 es_on, 4, tol)
            ld["w_n%d_es%d_tol%g" % (nit, es_on, tol)] = w
    # warm start
    rng = np.random.RandomState(3)
    w0 = 0.05 * rng.randn(len(yb))
    ld["w0"] = w0
    ld["w_warm_n5"] = bs._loops_deconv(yb.copy(), w0.copy(), Hd, 1.7, 5,
                                       False, 4, 1e-12)
    np.savez_compressed(os.path.join(HERE, "loops_deconv.npz"), **ld)
    print("loops_deconv N=%d K=%d" % (len(yb), len(h0)))

    # ---- case 6: bd, 5 outer iterations ---------------------------------
    x, z, dz, h, d = bs.bd(yb, t_r, lbda=1.7, hrf_dur=hrf_dur, nb_iter=5)
    np.savez_compressed(os.path.join(HERE, "bd.npz"), y=yb, t_r=t_r,
                        hrf_dur=hrf_dur, lbda=1.7, nb_iter=5, x=x, z=z,
                        diff_z=dz, h=h, J=d['J'], r=d['r'], g=d['g'])
    print("bd J", d['J'])
    # hrf_estim / hrf_fit_err
    z_true = data.gen_regular_bloc_bold(dur=3, tr=t_r, hrf=h_true, snr=10.0,
                                        random_state=0)[2]
    he_h, he_J = bs.hrf_estim(z_true, yb, t_r, hrf_dur)
    thetas = np.array([0.6, 0.7, 1.0, 1.3, 1.9])
    errs = np.array([bs.hrf_fit_err(t, z_true, yb, t_r, hrf_dur)
                     for t in thetas])
    np.savez_compressed(os.path.join(HERE, "hrf_estim.npz"), z=z_true, y=yb,
                        t_r=t_r, hrf_dur=hrf_dur, h=he_h, J=np.array(he_J),
                        thetas=thetas, errs=errs)

    # ---- case 7: operator known-answer tests ------------------------------
    rng = np.random.RandomState(7)
    ops = {}
    for tag, (n, t_r_, dur_, delta) in {
            "a": (300, 1.0, 30., 1.0), "b": (240, 0.75, 20., 0.7),
            "c": (600, 1.0, 30., 1.5), "d": (97, 2.0, 60., 0.5),
            "e": (600, 0.1, 60., 1.0)}.items():
        k = hm.spm_hrf(delta, t_r=t_r_, dur=dur_)[0]
        x = rng.randn(n)
        Hc = lin.ConvAndLinear(lin.DiscretInteg(), k, dim_in=n, dim_out=n)
        ops[tag + "_k"] = k
        ops[tag + "_x"] = x
        ops[tag + "_op"] = Hc.op(x)
        ops[tag + "_adj"] = Hc.adj(x)
        ops[tag + "_integ_op"] = lin.DiscretInteg().op(x)
        ops[tag + "_integ_adj"] = lin.DiscretInteg().adj(x)
        ops[tag + "_conv"] = cv.simple_convolve(k, x)
        ops[tag + "_retro"] = cv.simple_retro_convolve(k, x)
        ops[tag + "_spec"] = cv.spectral_convolve(k, x)
        ops[tag + "_spec_retro"] = cv.spectral_retro_convolve(k, x)
    # rectangular case of pybold/tests/test_convolution.py:169-176
    k = hm.spm_hrf(1.0, t_r=2.0)[0]
    sig = rng.randn(90)
    ops["rect_k"] = k
    ops["rect_sig"] = sig
    ops["rect_T"] = cv.toeplitz_from_kernel(sig, dim_in=len(k), dim_out=len(sig))
    ops["rect_conv"] = cv.simple_convolve(sig, k, dim_out=len(sig))
    ops["toep_small"] = cv.toeplitz_from_kernel(np.arange(1., 5.), 6, 6)
    np.savez_compressed(os.path.join(HERE, "operators.npz"), **ops)

    # ---- spm_hrf values, spectral radius ----------------------------------
    hv = {}
    for i, (delta, t_r_, dur_, norm) in enumerate([
            (1.0, 1.0, 30., True), (1.5, 1.0, 30., True),
            (2.0, 0.75, 20., False), (0.7, 0.75, 20., False),
            (0.5, 2.0, 60., True), (1.234, 0.72, 25., False)]):
        h, t = hm.spm_hrf(delta, t_r=t_r_, dur=dur_, normalized_hrf=norm)
        hv["p%d" % i] = np.array([delta, t_r_, dur_, float(norm)])
        hv["h%d" % i] = h
        hv["t%d" % i] = t
    np.savez_compressed(os.path.join(HERE, "spm_hrf.npz"), **hv)

    # ---- one random-block generator sample (distribution reference) ------
    # a seeded generator gets a single try (pybold/data.py:188): scan seeds
    smp = None
    for seed in range(1, 200):
        try:
            smp = data.gen_rnd_bloc_bold(dur=5, tr=1.0, hrf=hrf, nb_events=5,
                                         avg_dur=12, std_dur=1,
                                         overlapping=False, snr=1.0,
                                         random_state=seed)
            break
        except RuntimeError:
            continue
    np.savez_compressed(os.path.join(HERE, "rnd_bloc.npz"), seed=seed,
                        noisy=smp[0], clean=smp[1], ai_s=smp[2], i_s=smp[3],
                        noise=smp[6])
    print("done")


if __name__ == "__main__":
    main()
