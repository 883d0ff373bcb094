"""Series unlike the synthetic generator's, for the guards of the matrix-pipe kernel (round 4): shared by
tests/test_gpu_round4.py and tools/r4_adversarial_sweep.py.  Families, interleaved over the batch (problem p
-> family p % 8): the generator's own block signals at SNR 1 dB (control), DC baselines 10x / 100x / 1000x the
fluctuation (raw fMRI: mean >> fluctuation), Student-t noise (df 2.5), SNR -10 dB and +30 dB, and degenerate
series (constant, single spike, spike at sample 0, all zero).  Every series is divided by its own lambda_max
(of the fluctuating part), so that a call with the scalar `lbda = c` solves every problem at
lambda / lambda_max = c."""
import numpy as np
import torch

from oracle import pybold_oracle as orc
from pybold_amd import data, solver

FAMILIES = ["standard (SNR 1 dB)", "DC baseline 10x", "DC baseline 100x", "DC baseline 1000x",
            "Student-t noise (df 2.5)", "SNR -10 dB", "SNR +30 dB", "constant / spike / zero series"]
DC_FAMILIES = (1, 2, 3)
P_FAMILY = 2304                                            # x 8 families = 18 432 problems per call


def hrf_for(K):
    if K >= 20:
        return orc.spm_hrf(1.0, 1.0, float(K), False)[0][:K].copy()
    if K == 1:
        return np.array([0.8])
    return np.array([0.0, 0.6, 0.3, 0.1][:K]) if K <= 4 else np.hanning(K + 2)[1:-1] * 0.3


def make_batch(N, hrf, seed, dev):
    """(Y float32 (P, N) with the families interleaved (problem p -> family p % 8), lambda_max of the
    fluctuating part per problem)."""
    nf = len(FAMILIES)
    rng = np.random.RandomState(seed)
    snr = np.full(nf * P_FAMILY, 1.0)
    fam = np.arange(nf * P_FAMILY) % nf
    snr[fam == 5] = -10.0
    snr[fam == 6] = 30.0
    K = len(hrf)
    ev = 5 if N >= 250 else 2
    Yn, clean, _ = data.gen_rnd_bloc_bold_batch(nf * P_FAMILY, dur=(N + 0.5) / 60.0, tr=1.0, hrf=hrf, nb_events=ev,
                                                avg_dur=12.0 if N >= 250 else 8.0, std_dur=1.0,
                                                snr=torch.from_numpy(snr).to(dev), seed=seed, device=dev)
    Y = Yn.double()
    # Student-t noise at the family's SNR
    m = torch.from_numpy(fam == 4).to(dev)
    t = torch.from_numpy(rng.standard_t(2.5, size=(int(m.sum()), N))).to(dev)
    Y[m] = clean[m] + data._scale_noise(t, clean[m], 1.0)
    # degenerate series
    idx = np.nonzero(fam == 7)[0]
    for i, p in enumerate(idx):
        kind = i % 4
        row = torch.zeros(N, dtype=torch.float64, device=dev)
        if kind == 0:
            row += float(rng.uniform(0.5, 3.0)) * (1 if i % 8 < 4 else -1)
        elif kind == 1:
            row[int(rng.randint(1, N))] = float(rng.uniform(0.5, 3.0))
        elif kind == 2:
            row[0] = float(rng.uniform(0.5, 3.0))
        Y[p] = row
    lmax = solver.lambda_max(Y.float(), hrf)               # of the fluctuating part
    lmax = torch.where(lmax > 0, lmax, torch.ones_like(lmax))
    Y = Y / lmax[:, None]                                   # lambda / lambda_max = lambda for every problem
    # DC baselines, relative to the standard deviation of the (normalised) fluctuation
    sd = Y.std(dim=1, keepdim=True)
    for f, mult in zip(DC_FAMILIES, (10.0, 100.0, 1000.0)):
        m = torch.from_numpy(fam == f).to(dev)
        sign = torch.where(torch.arange(int(m.sum()), device=dev) % 2 == 0, 1.0, -1.0).double()[:, None]
        Y[m] = Y[m] + sign * mult * sd[m]
    return Y.float().contiguous(), fam
