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
  * white Gaussian noise rescaled to the exact SNR in dB (data.py:436-444), one SNR for
    the batch or one per voxel.

``gen_regular_bloc_bold_batch`` is the batched regular block design (data.py:10-41).
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
    return (clean + _scale_noise(noise, clean, snr)).to(torch.float32), clean, blocks


def _scale_noise(noise, clean, snr):
    """``add_gaussian_noise`` (pybold/data.py:436-444) row-wise: the unit draw is rescaled so
    that ``20 log10(||clean|| / ||noise||) = snr`` exactly; ``snr`` is a scalar or one value
    per voxel."""
    ratio = clean.norm(dim=1, keepdim=True) / (noise.norm(dim=1, keepdim=True) +
                                               np.finfo(np.float64).eps)
    snr = torch.as_tensor(snr, dtype=torch.float64, device=clean.device).reshape(-1, 1)
    return noise * ratio / torch.sqrt(10.0 ** (snr / 10.0))


def gen_regular_bloc_bold_batch(n_voxels, dur=10, tr=1.0, dur_bloc=30.0, hrf=None, snr=1.0, seed=0,
                                device=None, centered=True):
    """Batched ``gen_regular_bloc_bold`` (pybold/data.py:10-41): the innovation alternates
    +1 / -1 every ``int(dur_bloc / tr)`` samples from sample 0 (:14-16), the block signal is
    its cumulative sum, both centred (:20-22), the clean BOLD signal is ``hrf * blocks``
    (:34, on the GPU), and every voxel gets its own white-noise draw rescaled to the exact
    SNR (scalar or per voxel).  Returns ``(noisy float32 (V, N), clean float64 (N,),
    blocks float64 (N,), innovation float64 (N,))`` as CUDA tensors."""
    if hrf is None:
        raise ValueError("an HRF is required")
    dev = solver.device(device)
    N = int(dur * 60 / tr)
    V = int(n_voxels)
    i_s = torch.zeros((N,), dtype=torch.float64, device=dev)
    marks = torch.arange(0, N, int(dur_bloc / tr), device=dev)
    i_s[marks] = -1.0
    i_s[marks[::2]] = 1.0
    ai_s = torch.cumsum(i_s, dim=0)
    if centered:
        ai_s = ai_s - ai_s.mean()
        i_s = i_s - i_s.mean()
    clean = solver.conv(ai_s.reshape(1, N), np.asarray(hrf, dtype=np.float64))[0]
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    noise = torch.randn((V, N), generator=gen, device=dev, dtype=torch.float64)
    noisy = clean[None, :] + _scale_noise(noise, clean[None, :].expand(V, N), snr)
    return noisy.to(torch.float32), clean, ai_s, i_s
