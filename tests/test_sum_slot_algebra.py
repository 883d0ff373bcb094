"""CPU-only: the block algebra of the matrix-pipe kernel's sum-slot form (pybold_amd/csrc/fista_mfma.h)
restated in NumPy float64 -- tiles built by the kernel's rules, the forward and adjoint recurrences block by block --
against the dense operators of the reference (pybold/linear.py:73-113: x = K_h cumsum(w), g = cumsum^T K_h^T r).

A block is 32 slots: 31 samples + one SUM SLOT (slot 31).  As an output row of tile 0 the slot adds up the block
(entries 2^-9, and 1 at [31][31]): D_{q+1} = 2^-9 sum(w_q) + D_q; as an input column of the LAST near tile (entries
2^9 S) it feeds S sum_{b <= q-NT} w_b to the 31 real rows of block q.  No carry tile, no far-field product of its own.
The test pins the identity (it holds exactly in exact arithmetic), the limits on the HRF length (K <= 33 with two near
tiles, K <= 64 with three) and the first version's mistake (telescoping x block against block needs three blocks)."""
import numpy as np
import pytest

from oracle import pybold_oracle as orc

SPAN, SSHIFT = 31, 9


def tiles(c, S, NT):
    """[o] -> (A, B): forward / adjoint near tile o as 32 x 32 arrays indexed [output slot][input slot] -- the rules of
    fista_mfma.h's tile builder (`special`, `vs`, `cval`)."""
    eps, sfar = 2.0 ** -SSHIFT, S * 2.0 ** SSHIFT

    def cval(lag):
        return 0.0 if lag < 0 else (c[lag] if lag < len(c) else S)
    out = []
    for o in range(NT):
        A, B = np.zeros((32, 32)), np.zeros((32, 32))
        for ko in range(32):
            for ki in range(32):
                if ko == 31:
                    v = (1.0 if ki == 31 else eps) if o == 0 else 0.0
                    A[ko, ki] = B[ko, ki] = v
                elif ki == 31:
                    A[ko, ki] = B[ko, ki] = sfar if o == NT - 1 else 0.0
                else:
                    A[ko, ki] = cval(SPAN * o + ko - ki)          # forward: input block q-o, lag = 31 o + ko - ki
                    B[ko, ki] = cval(SPAN * o + ki - ko)          # adjoint: input block q+o, lag = 31 o + ki - ko
        out.append((A, B))
    return out


def slots(v, NB):
    """series (N,) -> (NB, 32) slots: sample 31 q + k in slot k < 31, slot 31 = 0 (padding behind N: 0)"""
    s = np.zeros((NB, 32))
    flat = np.zeros(NB * SPAN)
    flat[:len(v)] = v
    s[:, :SPAN] = flat.reshape(NB, SPAN)
    return s


def forward(w, y, T, NB, NT, N):
    """r = T_c w - y through the blocks, ascending; the sum slot of block q takes the finished sum row of block q-1"""
    W, Y = slots(w, NB), slots(y, NB)
    R = np.zeros((NB, 32))
    d = 0.0
    frags = []
    for q in range(NB):
        f = W[q].copy()
        f[31] = d                                                   # D_q = 2^-9 (w_0 + ... + w_{q-1})
        frags.append(f)
        acc = -Y[q].copy()                                          # the accumulators start from -y (sum row: from 0)
        for o in range(NT - 1, -1, -1):                             # oldest tile first
            if q >= o:
                acc += T[o][0] @ frags[q - o]
        d = acc[31]                                                 # the finished sum row: D_{q+1}
        R[q] = acc
        R[q, 31] = 0.0                                              # (the stored fragment's slot is rebuilt by the adjoint)
    r = R[:, :SPAN].reshape(-1)
    r[N:] = 0.0                                                     # padding behind sample N-1
    return r[:N], d


def adjoint(r, T, NB, NT, N):
    """g = T_c^T r through the blocks, descending; the sum slot of block q takes the finished sum row of block q+1"""
    R = slots(r, NB)
    G = np.zeros((NB, 32))
    d = 0.0
    frags = [None] * NB
    for q in range(NB - 1, -1, -1):
        f = R[q].copy()
        f[31] = d                                                   # 2^-9 (r_{q+1} + ... + r_{NB-1})
        frags[q] = f
        acc = np.zeros(32)
        for o in range(NT - 1, -1, -1):
            if q + o < NB:
                acc += T[o][1] @ frags[q + o]
        d = acc[31]
        G[q] = acc
    return G[:, :SPAN].reshape(-1)[:N]


def dense(h, N):
    """T_c = K_h cumsum as a dense (N, N) matrix (pybold/convolution.py:105-132 + pybold/linear.py:15-43)"""
    return orc.toeplitz_from_kernel(h, N, N).dot(np.tril(np.ones((N, N))))


@pytest.mark.parametrize("N,K,NT", [(300, 30, 2), (310, 33, 2), (129, 1, 2), (155, 2, 2), (156, 27, 2), (240, 27, 2),
                                    (300, 34, 3), (300, 48, 3), (304, 64, 3), (187, 40, 3)])
def test_block_algebra_equals_the_dense_operators(N, K, NT):
    rng = np.random.RandomState(N + K)
    h = rng.randn(K) * 0.3
    c = np.cumsum(h)
    S = c[-1]
    NB = -(-N // SPAN)
    T = tiles(c, S, NT)
    w, y, r = rng.randn(N), rng.randn(N) + 50.0, rng.randn(N)       # (a DC baseline in y: it never enters the sums)
    Tc = dense(h, N)
    res, total = forward(w, y, T, NB, NT, N)
    assert np.abs(res - (Tc @ w - y)).max() < 1e-10 * (np.abs(Tc @ w).max() + np.abs(y).max())
    assert abs(total - 2.0 ** -SSHIFT * w.sum()) < 1e-12 * np.abs(w).sum()      # the last sum row: the whole series
    g = adjoint(r, T, NB, NT, N)
    assert np.abs(g - Tc.T @ r).max() < 1e-10 * np.abs(Tc.T @ r).max()


def test_limits_on_the_hrf_length():
    """Two near tiles reach lags 0 .. 61 and the far field must be constant from lag 32 on (K <= 33); one tap more and
    the identity breaks -- which is what the dispatch's MFMA_K2 = 33 / MFMA_K3 = 64 encode."""
    rng = np.random.RandomState(3)
    for K, NT, ok in [(33, 2, True), (34, 2, False), (64, 3, True), (65, 3, False)]:
        N = 300
        h = rng.randn(K) * 0.3
        c = np.cumsum(h)
        T = tiles(c, c[-1], NT)
        w, y = rng.randn(N), rng.randn(N)
        res, _ = forward(w, y, T, -(-N // SPAN), NT, N)
        err = np.abs(res - (dense(h, N) @ w - y)).max()
        assert (err < 1e-9) == ok, (K, NT, err)


def test_why_telescoping_block_against_block_needs_three_blocks():
    """The first attempt of round 4: x_q - x_{q-1} = sum_l e[l] w[t - l] with e[l] = c[l] - c[l - 32], whose support is
    K + 31 lags -- row 0 of a block then reaches 60 samples back, into block q-2: two near tiles are not enough for
    any K > 2 (the sum-slot form is what was built instead)."""
    K = 30
    c = np.cumsum(np.ones(K))
    cc = lambda l: 0.0 if l < 0 else (c[l] if l < K else c[-1])
    e = np.array([cc(l) - cc(l - 32) for l in range(96)])
    support = np.flatnonzero(e).max() + 1
    assert support == K + 31 and support > 32 + 1                   # row 0 of a block sees lags 0 .. 32 of two aligned blocks
