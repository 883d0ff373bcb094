"""GPU tests of the rows next to the hot loop: the shared-HRF blind step
(config 4, one rank here; the 2-rank logic is covered on CPU with gloo) and the
vectorised synthetic generator (pybold/data.py distributions)."""
import os

import numpy as np
import pytest
import torch

from oracle import pybold_oracle as orc

pytestmark = pytest.mark.gpu


def test_generator_properties():
    from pybold_amd import data
    hrf = orc.spm_hrf(1.0, t_r=1.0, dur=30.)[0]
    Y, clean, blocks = data.gen_rnd_bloc_bold_batch(512, dur=5, tr=1.0, hrf=hrf, nb_events=5,
                                                    avg_dur=12.0, std_dur=1.0, snr=1.0, seed=3)
    assert Y.shape == (512, 300) and Y.dtype == torch.float32 and Y.is_cuda
    b = blocks.cpu().numpy()
    assert set(np.unique(b)) <= {0.0, 1.0}
    onsets = (np.diff(np.concatenate([np.zeros((512, 1)), b], axis=1), axis=1) > 0.5).sum(axis=1)
    assert (onsets == 5).all()                       # 5 separated blocks (data.py:228-230)
    dur = b.sum(axis=1) / 5.0
    assert 10.5 < dur.mean() < 13.0                  # ~ N(12 s, 1 s), floor to samples
    c = clean.cpu().numpy()
    np.testing.assert_allclose(c, orc.causal_conv(hrf, b), atol=1e-10)   # data.py:324
    noise = Y.cpu().numpy().astype(np.float64) - c
    snr = 20 * np.log10(np.linalg.norm(c, axis=1) / np.linalg.norm(noise, axis=1))
    np.testing.assert_allclose(snr, 1.0, atol=1e-3)  # exact-SNR scaling (data.py:436-444)
    Y2, _, _ = data.gen_rnd_bloc_bold_batch(512, dur=5, tr=1.0, hrf=hrf, seed=3)
    assert torch.equal(Y, Y2)                        # seeded


def test_bd_shared_single_voxel_tracks_bd(golden):
    """V = 1: shared-theta blind deconvolution IS the per-voxel `bd` (pybold/bold_signal.py:281-382;
    SURVEY 0.1).  Strict form: float64 series and the reference's own optimiser (L-BFGS-B with a
    finite-difference gradient) -> the golden `bd` run at 1e-5 (h) / 1e-4 (diff_z); the production
    form (float32 series, theta fitted on the device from normal equations) lands within 5e-5 of
    that dilation after every outer iteration."""
    from pybold_amd import distributed
    g, g2 = golden("bd"), golden("round2")
    y, t_r, dur = g["y"], float(g["t_r"]), float(g["hrf_dur"])
    Yd = torch.from_numpy(y[None].astype(np.float64)).cuda()
    # the reference's bd runs nb_iter(=5) inner iterations per outer one (:324)
    W, h, d = distributed.bd_shared(Yd, t_r, lbda=float(g["lbda"]), hrf_dur=dur, nb_iter=5, nb_inner=5,
                                    theta_solver="lbfgsb")
    assert np.linalg.norm(h - g["h"]) / np.linalg.norm(g["h"]) < 1e-5
    w = W.cpu().numpy()[0]
    assert np.linalg.norm(w - g["diff_z"]) / np.linalg.norm(g["diff_z"]) < 1e-4
    np.testing.assert_allclose(d["theta"][1:], g2["bd_theta"], atol=2e-6)   # the dilation after every outer iteration
    Y = torch.from_numpy(y[None].astype(np.float32)).cuda()
    W2, h2, d2 = distributed.bd_shared(Y, t_r, lbda=float(g["lbda"]), hrf_dur=dur, nb_iter=5, nb_inner=5)
    assert np.abs(np.asarray(d2["theta"]) - np.asarray(d["theta"])).max() < 5e-5
    assert np.linalg.norm(h2 - g["h"]) / np.linalg.norm(g["h"]) < 1e-4
    assert np.linalg.norm(W2.cpu().numpy()[0] - g["diff_z"]) / np.linalg.norm(g["diff_z"]) < 1e-3
    assert d2["J"][-1] < d2["J"][1] < 1.0


def test_bd_shared_recovers_common_dilation():
    from pybold_amd import data, distributed
    t_r, dur, theta_true = 0.75, 20.0, 0.7
    h_true = orc.spm_hrf(theta_true, t_r, dur, False)[0]
    Y, _, _ = data.gen_rnd_bloc_bold_batch(256, dur=3.75, tr=t_r, hrf=h_true, nb_events=5,
                                           avg_dur=12.0, std_dur=1.0, snr=10.0, seed=0)
    W, h, d = distributed.bd_shared(Y, t_r, lbda=1.7, hrf_dur=dur, nb_iter=8, nb_inner=100)
    # starts at the upper bound 1.9 and moves towards the generating dilation (the
    # L1-regularised joint problem is biased, so only the direction is asserted)
    assert d["theta"][1] <= 1.9 and abs(d["theta"][-1] - theta_true) < 0.4
    assert d["theta"][-1] < d["theta"][1]
    assert (np.diff(d["J"][1:]) < 1e-6).all()        # global cost decreases


def test_regular_block_generator_matches_reference(golden):
    """gen_regular_bloc_bold_batch vs the reference's gen_regular_bloc_bold (round-2 fixture):
    innovation, block signal and clean BOLD signal identical; every voxel's noise has the
    exact SNR; per-voxel SNR values are honoured."""
    from pybold_amd import data
    g = golden("round2")
    for tag in "ab":
        dur, tr, dur_bloc, snr, _ = g["reg_%s_p" % tag]
        noisy, clean, ai_s, i_s = data.gen_regular_bloc_bold_batch(
            64, dur=int(dur), tr=tr, dur_bloc=dur_bloc, hrf=g["reg_%s_hrf" % tag], snr=snr, seed=1)
        np.testing.assert_allclose(i_s.cpu().numpy(), g["reg_%s_i_s" % tag], rtol=0, atol=1e-13)
        np.testing.assert_allclose(ai_s.cpu().numpy(), g["reg_%s_ai_s" % tag], rtol=0, atol=1e-12)
        np.testing.assert_allclose(clean.cpu().numpy(), g["reg_%s_clean" % tag], rtol=1e-12, atol=1e-12)
        noise = noisy.double() - clean[None]
        got = 20 * torch.log10(clean.norm() / noise.norm(dim=1))
        np.testing.assert_allclose(got.cpu().numpy(), snr, atol=1e-3)     # float32 storage of noisy
        assert noisy.shape == (64, len(g["reg_%s_clean" % tag])) and noisy.dtype == torch.float32
        assert float((noise[0] - noise[1]).abs().max()) > 0                 # one draw per voxel
    snrs = np.array([1.0, 5.0, 10.0, 20.0])
    hrf = g["reg_a_hrf"]
    noisy, clean, _, _ = data.gen_regular_bloc_bold_batch(4, dur=3, tr=0.75, hrf=hrf, snr=snrs, seed=2)
    got = 20 * torch.log10(clean.norm() / (noisy.double() - clean[None]).norm(dim=1))
    np.testing.assert_allclose(got.cpu().numpy(), snrs, atol=1e-3)
    Y, c, b = data.gen_rnd_bloc_bold_batch(4, dur=5, tr=1.0, hrf=golden("case1")["hrf"], snr=snrs, seed=2)
    got = 20 * torch.log10(c.norm(dim=1) / (Y.double() - c).norm(dim=1))
    np.testing.assert_allclose(got.cpu().numpy(), snrs, atol=1e-3)


def test_reference_random_block_sample(golden):
    """The captured sample of the reference's gen_rnd_bloc_bold (rnd_bloc.npz): the batched
    generator produces signals of the same family -- 5 unit blocks, clean = hrf * blocks on the
    GPU Toeplitz kernel (pybold/data.py:324), noisy - clean at the requested SNR -- and the
    device inf_norm reproduces the reference's normalisation of it."""
    from pybold_amd import data, solver
    from pybold_amd.utils import inf_norm
    g, hrf = golden("rnd_bloc"), golden("case1")["hrf"]
    ai_s, clean, noisy = g["ai_s"], g["clean"], g["noisy"]
    dev_clean = solver.conv(torch.from_numpy(ai_s[None].copy()).cuda(), hrf).cpu().numpy()[0]
    np.testing.assert_allclose(dev_clean, clean, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(noisy - clean, g["noise"], rtol=0, atol=1e-12)
    assert 20 * np.log10(np.linalg.norm(clean) / np.linalg.norm(g["noise"])) == pytest.approx(1.0, abs=1e-9)
    n_blocks_ref = int((np.diff(np.concatenate([[0.0], ai_s])) > 0).sum())
    Y, c, b = data.gen_rnd_bloc_bold_batch(256, dur=5, tr=1.0, hrf=hrf, nb_events=5, avg_dur=12.0,
                                           std_dur=1.0, snr=1.0, seed=7)
    rises = (torch.diff(b, dim=1, prepend=torch.zeros((256, 1), dtype=b.dtype, device=b.device)) > 0).sum(dim=1)
    assert n_blocks_ref == 5 and bool((rises == 5).all())
    assert set(np.unique(ai_s)) == {0.0, 1.0} and set(torch.unique(b).cpu().numpy()) == {0.0, 1.0}
    # block durations of the same distribution: mean 12 s, a few seconds of spread
    dur_ref = ai_s.sum() / 5
    dur_gen = float(b.sum(dim=1).mean() / 5)
    assert abs(dur_gen - 12.0) < 1.0 and abs(dur_ref - 12.0) < 3.0
    out = inf_norm([noisy, clean, ai_s])
    for o, a in zip(out, (noisy, clean, ai_s)):
        np.testing.assert_allclose(o, a / (np.abs(a).max() + 1e-12), rtol=1e-15)


def test_icassp_simulation_example_runs():
    """examples/icassp_simulation.py (generate -> batched bd with per-voxel lambda -> inf_norm)
    end to end on a reduced setting."""
    import runpy
    import sys
    argv = sys.argv
    sys.argv = ["icassp_simulation.py", "--voxels", "8", "--iters", "6"]
    try:
        runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "examples", "icassp_simulation.py"), run_name="__main__")
    finally:
        sys.argv = argv
