"""Vectorised synthetic BOLD generator on the GPU.

Reproduces the distributions of the reference's single-voxel generators
(``gen_rnd_ai_s`` / ``gen_rnd_bloc_bold`` / ``add_gaussian_noise``,
pybold/data.py:125-329, :403-446) for a whole batch of voxels at once, so that
the ICASSP-style simulations (examples/icassp_2019/simulation.py:100-115) need
no per-voxel Python loop:

  * ``nb_events`` unit blocks per voxel, durations ``avg_dur + std_dur**2 *
    randn`` seconds (the reference scales the *variance* by ``1/dt``,
    data.py:180-194, so the standard deviation is ``std_dur**2`` seconds),
    truncated below at one sample;
  * onsets uniform over the non-overlapping placements (the reference rejects
    overlapping draws, data.py:212-213; here the free space between blocks is
    drawn directly);
  * clean signal = causal convolution with the HRF (data.py:324);
  * white Gaussian noise rescaled to the exact SNR in dB (data.py:436-444).
"""
import numpy as np
import torch

from . import solver


def gen_rnd_bloc_bold_batch(n_voxels, dur=5, tr=1.0, hrf=None, nb_events=5, avg_dur=12.0,
                            std_dur=1.0, snr=1.0, seed=0, device=None):
    """Returns ``(noisy float32 (V, N), clean float64 (V, N), blocks float64 (V, N))``
    as CUDA tensors, ``N = int(dur * 60 / tr)``."""
    if hrf is None:
        raise ValueError("an HRF is required")
    dev = solver.device(device)
    N = int(dur * 60 / tr)
    V = int(n_voxels)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    # durations in samples, >= 1
    d = avg_dur + (std_dur ** 2) * torch.randn((V, nb_events), generator=gen, device=dev,
                                               dtype=torch.float64)
    d = torch.clamp((d / tr).floor(), min=1.0)
    total = d.sum(dim=1, keepdim=True)
    if float(total.max()) + nb_events > N:
        raise ValueError("events do not fit in the run")
    # free samples are split into nb_events + 1 gaps; inner gaps >= 1 keep blocks apart
    free = N - total - (nb_events - 1)
    cuts = torch.rand((V, nb_events), generator=gen, device=dev, dtype=torch.float64)
    cuts = (cuts.sort(dim=1).values * (free + 1)).floor()
    lead = torch.diff(cuts, dim=1, prepend=torch.zeros((V, 1), device=dev, dtype=torch.float64))
    lead[:, 1:] += 1.0
    onset = torch.cumsum(lead, dim=1) + torch.cumsum(d, dim=1) - d
    t = torch.arange(N, device=dev, dtype=torch.float64)[None, None, :]
    inside = (t >= onset[:, :, None]) & (t < (onset + d)[:, :, None])
    blocks = inside.any(dim=1).to(torch.float64)
    clean = solver.conv(blocks, np.asarray(hrf, dtype=np.float64))
    noise = torch.randn((V, N), generator=gen, device=dev, dtype=torch.float64)
    scale = clean.norm(dim=1, keepdim=True) / (noise.norm(dim=1, keepdim=True) +
                                               np.finfo(np.float64).eps)
    noise = noise * scale / np.sqrt(10.0 ** (snr / 10.0))
    return (clean + noise).to(torch.float32), clean, blocks
