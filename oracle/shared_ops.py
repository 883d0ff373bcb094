"""CPU stand-in for `pybold_amd.distributed.HipOps`: the five compute steps of the shared-HRF
blind loop (z-step with the Frobenius step of pybold/bold_signal.py:249-255, normal equations of
`hrf_fit_err` :217-222, 1-D search over the dilation) on the float64 NumPy oracle.
TEST INFRASTRUCTURE ONLY: used by the gloo tests of the multi-rank logic and by the parity leg of
`bench.py --config 4`; the product never imports it."""
import numpy as np
import torch

from oracle import pybold_oracle as orc


class OracleOps:
    def __init__(self, n, t_r, hrf_dur):
        self.n, self.t_r, self.hrf_dur = n, t_r, hrf_dur

    def hrf(self, theta):
        return torch.from_numpy(orc.spm_hrf(float(theta[0]), self.t_r, self.hrf_dur, False)[0].copy())

    def z_step(self, Y, taps, lbda, nb_inner, W):
        if Y.shape[0] == 0:
            return W
        h = taps.numpy()
        step = 1.0 / orc.gram_lipschitz(h, self.n)
        return torch.from_numpy(orc.fista_batch(Y.numpy().astype(np.float64), h, lbda, step,
                                                nb_inner, W0=W.numpy()))

    def normal_eq(self, W, Y, K):
        G, b, yy = orc.hrf_normal_eq(np.cumsum(W.numpy(), axis=1), Y.numpy().astype(np.float64), K) \
            if Y.shape[0] else (np.zeros((K, K)), np.zeros(K), 0.0)
        return torch.from_numpy(np.concatenate([G.ravel(), b, [yy]]))

    def theta_fit(self, ne, bounds):
        K = int(round((-1 + np.sqrt(1 + 4 * (ne.numel() - 1))) / 2))
        v = ne.numpy()
        th, f, h = orc.theta_fit_normal_eq(v[:K * K].reshape(K, K), v[K * K:K * K + K], v[-1],
                                           self.t_r, self.hrf_dur, bounds)
        return torch.tensor([th]), torch.tensor([f]), torch.from_numpy(h.copy())
