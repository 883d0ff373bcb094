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


def hrf_normal_eq_blas(Z, Y, K):
    """:func:`oracle.pybold_oracle.hrf_normal_eq` for thousands of voxels: the same sums
    (``G[m, m'] = sum_v sum_i z_v[i - m] z_v[i - m']``, ``b[m] = sum_v sum_i z_v[i - m] y_v[i]``,
    ``yy``; pybold/bold_signal.py:217-222 written as a quadratic form in the taps) without one dense
    Toeplitz matrix per voxel: ``G[m, m'] = R_d(n - 1 - max(m, m'))`` with ``d = |m - m'|`` and the
    partial autocorrelations ``R_d(T) = sum_v sum_{j <= T} z_v[j] z_v[j + d]`` -- K products of the
    batch with a shifted copy of itself and a cumulative sum each.  Checked against the per-voxel
    form in tests/test_host_logic.py."""
    Z = np.atleast_2d(np.asarray(Z, dtype=np.float64))
    Y = np.atleast_2d(np.asarray(Y, dtype=np.float64))
    n = Z.shape[1]
    G, b = np.zeros((K, K)), np.zeros(K)
    for d in range(min(K, n)):
        R = np.cumsum(np.einsum("vj,vj->j", Z[:, :n - d], Z[:, d:]))      # R_d(T), T = 0 .. n-1-d
        for m in range(K - d):
            if n - 1 - (m + d) >= 0:
                G[m, m + d] = G[m + d, m] = R[n - 1 - (m + d)]
        b[d] = float(np.einsum("vj,vj->", Z[:, :n - d], Y[:, d:]))
    return G, b, float(np.einsum("vj,vj->", Y, Y))


class FastOracleOps(OracleOps):
    """:class:`OracleOps` for batches of 10^4 voxels (the full-size config-4 parity test): z-steps on
    the C/OpenMP form of the oracle (oracle/fista_oracle.c, same recurrence, pinned to the same
    goldens), normal equations through :func:`hrf_normal_eq_blas`."""

    def __init__(self, n, t_r, hrf_dur, threads=0):
        super().__init__(n, t_r, hrf_dur)
        self.threads = threads

    def z_step(self, Y, taps, lbda, nb_inner, W):
        from oracle import c_oracle
        if Y.shape[0] == 0:
            return W
        h = taps.numpy()
        step = 1.0 / orc.gram_lipschitz(h, self.n)
        Wn, _, _ = c_oracle.fista_batch(Y.numpy().astype(np.float64), h, lbda, step, nb_inner, W0=W.numpy(),
                                        threads=self.threads)
        return torch.from_numpy(Wn)

    def normal_eq(self, W, Y, K):
        G, b, yy = hrf_normal_eq_blas(np.cumsum(W.numpy(), axis=1), Y.numpy().astype(np.float64), K) \
            if Y.shape[0] else (np.zeros((K, K)), np.zeros(K), 0.0)
        return torch.from_numpy(np.concatenate([G.ravel(), b, [yy]]))
