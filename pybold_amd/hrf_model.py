"""SPM two-gamma HRF with a time-dilation parameter (``pybold/hrf_model.py``).

Host helper of the blind step.  Unlike the reference it evaluates the gamma
densities only at the ``K`` decimated sample times when no max-normalisation
is requested (the reference evaluates a 1 ms grid of ``dur/dt`` points and then
keeps every ``t_r/dt``-th one, pybold/hrf_model.py:25-37).
"""
import numpy as np
from scipy.stats import gamma

MIN_DELTA = 0.5
MAX_DELTA = 2.0


def spm_hrf(delta, t_r=1.0, dur=60.0, normalized_hrf=True, dt=0.001, p_delay=6,
            undershoot=16.0, p_disp=1.0, u_disp=1.0, p_u_ratio=0.167, onset=0.0):
    """Same signature and values as pybold/hrf_model.py:12-39; returns
    ``(hrf, t_hrf)``.  Raises ``ValueError`` for ``delta`` outside [0.5, 2]."""
    if not (MIN_DELTA <= delta <= MAX_DELTA):
        raise ValueError("HRF dilation delta=%r outside the supported range [%g, %g]"
                         % (delta, MIN_DELTA, MAX_DELTA))
    n_fine = int(float(dur) / dt)
    dec = int(t_r / dt)
    shift = float(onset) / dt

    def density(t):
        ts = delta * t
        peak = gamma.pdf(ts, p_delay / p_disp, loc=dt / p_disp)
        under = gamma.pdf(ts, undershoot / u_disp, loc=dt / u_disp)
        return peak - p_u_ratio * under

    if normalized_hrf:
        t = np.linspace(0, dur, n_fine) - shift
        hrf = density(t)
        hrf = hrf / np.max(hrf + 1.0e-30)
        return hrf[::dec], t[::dec]
    # grid point i of linspace(0, dur, n_fine) is i * dur / (n_fine - 1)
    idx = np.arange(0, n_fine, dec)
    t = np.linspace(0, dur, n_fine)[idx] - shift
    return density(t), t
