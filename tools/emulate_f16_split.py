"""NumPy emulation of the FISTA loop with its two FIRs done as split 16-bit products
(the arithmetic a matrix-pipe FIR would use), on the golden inputs, 500 iterations.

Question (VERDICT r2 item 5 ii): does eps <= 1e-5 hold if both FIRs are computed as
  x = hh*zh + hh*zl + hl*zh      (h, z each split in two float16 parts, products exact,
                                  accumulated in float32)
with the iterate and the update in float64 as in the shipped kernels?

Variants:
  f32        scans + FIRs in float32 (the shipped arithmetic)           -> reference point
  f16x2      2-way float16 split of window and taps, 3 products
  f16x2_2p   the same with only 2 products (hh*zh + hh*zl: taps not split)  -> expected to fail
  bf16x3     3-way bfloat16 split, 6 products
  bf16x2     2-way bfloat16 split, 3 products                            -> expected to fail
The series is scaled per voxel by a power of two so that max|y| ~ 2^10 (float16 range), as a
kernel would do once at load time (exact: the problem is scale-covariant, threshold included).

Usage: python tools/emulate_f16_split.py        (CPU only; prints a table)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import pybold_oracle as orc  # noqa: E402  (tools/ may use the checker)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def to_f16_rtz(x32):
    """float32 -> float16, round toward zero, saturating (v_cvt_pkrtz_f16_f32)."""
    x32 = np.asarray(x32, dtype=np.float32)
    h = x32.astype(np.float16)
    h = np.where(np.isinf(h), np.sign(x32) * np.float16(65504.0), h).astype(np.float16)
    over = np.abs(h.astype(np.float32)) > np.abs(x32)
    h = np.where(over, np.nextafter(h, np.float16(0.0)), h).astype(np.float16)
    return h


def to_bf16(x32):
    """float32 -> bfloat16 (kept as float32 with the low 16 bits cleared), RNE."""
    u = np.asarray(x32, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def split(x32, kind, parts):
    """list of float32 arrays (each exactly representable in the 16-bit format) summing ~ x32"""
    out, rem = [], np.asarray(x32, dtype=np.float32)
    for _ in range(parts):
        p = to_f16_rtz(rem).astype(np.float32) if kind == "f16" else to_bf16(rem)
        out.append(p)
        rem = (rem - p).astype(np.float32)          # exact in float32
    return out


def fir_f32acc(taps_parts, win_parts, pairs, causal):
    """sum over the listed (tap part, window part) pairs of the K-tap FIR, float32 accumulate
    (sequential over taps like a matrix instruction's k-loop)."""
    n = win_parts[0].shape[-1]
    acc = np.zeros(win_parts[0].shape, dtype=np.float32)
    for (i, j) in pairs:
        h, z = taps_parts[i], win_parts[j]
        for m in range(len(h)):
            if h[m] == 0.0:
                continue
            if causal:
                acc[..., m:] = (acc[..., m:] + h[m] * z[..., : n - m]).astype(np.float32)
            else:
                acc[..., : n - m] = (acc[..., : n - m] + h[m] * z[..., m:]).astype(np.float32)
    return acc


def run(Y, hrf, lbda, step, n_iter, mode):
    V, N = Y.shape
    betas = orc.momentum_sequence(n_iter)
    # per-voxel power-of-two scale: max|y| -> [2^9, 2^10)
    sc = 2.0 ** (10 - np.ceil(np.log2(np.abs(Y).max(axis=1))))
    Ys = (Y * sc[:, None]).astype(np.float32)
    th = lbda * step * sc[:, None]
    W = np.zeros((V, N))
    if mode == "f32":
        hp = [hrf.astype(np.float32)]
        pairs = [(0, 0)]
        kind, parts = None, 1
    else:
        kind = "f16" if mode.startswith("f16") else "bf16"
        parts = 3 if mode == "bf16x3" else 2
        hs = 2.0 ** (13 - np.ceil(np.log2(np.abs(hrf).max())))     # taps scaled into range too
        hp = split((hrf * hs).astype(np.float32), kind, parts)
        if mode == "f16x2_2p":
            pairs = [(0, 0), (0, 1)]
        elif parts == 2:
            pairs = [(0, 0), (0, 1), (1, 0)]
        else:
            pairs = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
    zmax = 0.0
    for k in range(n_iter):
        z = np.cumsum(W.astype(np.float32), axis=1, dtype=np.float32)
        zmax = max(zmax, float(np.abs(z).max()))
        if kind is None:
            x = fir_f32acc(hp, [z], pairs, True)
        else:
            x = (fir_f32acc(hp, split(z, kind, parts), pairs, True) / np.float32(hs)).astype(np.float32)
        r = (x - Ys).astype(np.float32)
        if kind is None:
            c = fir_f32acc(hp, [r], pairs, False)
        else:
            c = (fir_f32acc(hp, split(r, kind, parts), pairs, False) / np.float32(hs)).astype(np.float32)
        g = np.cumsum(c[:, ::-1], axis=1, dtype=np.float32)[:, ::-1]
        U = W - step * g.astype(np.float64)
        D = np.clip(U, -th, th)
        W = U - (1.0 + betas[k]) * D
    return W / sc[:, None], zmax


def main():
    g = np.load(os.path.join(GOLD, "grid.npz"))
    hrf = g["hrf"]
    n_iter = 500
    rows = []
    for mode in ("f32", "f16x2", "f16x2_2p", "bf16x3", "bf16x2"):
        worst = {"dz": 0.0, "z": 0.0, "x": 0.0}
        zmax_all = 0.0
        for lb in ("0.1", "1", "10"):
            Y = np.stack([g[f"y_s{s}"] for s in range(4)])
            step = 1.0 / float(g["lip_s0"])
            # every seed has its own Lipschitz constant in the golden file; run them one by one
            for s in range(4):
                step = 1.0 / float(g[f"lip_s{s}"])
                W, zmax = run(Y[s:s + 1], hrf, float(lb), step, n_iter, mode)
                zmax_all = max(zmax_all, zmax)
                ref = g[f"dz_s{s}_l{lb}_n500"]
                zr, xr = orc.fista_outputs(ref[None, :], hrf)
                zz, xx = orc.fista_outputs(W, hrf)
                worst["dz"] = max(worst["dz"], np.linalg.norm(W[0] - ref) / np.linalg.norm(ref))
                worst["z"] = max(worst["z"], np.linalg.norm(zz - zr) / np.linalg.norm(zr))
                worst["x"] = max(worst["x"], np.linalg.norm(xx - xr) / np.linalg.norm(xr))
        rows.append((mode, worst["dz"], worst["z"], worst["x"], zmax_all))
    print("max relative L2 error vs the reference goldens (lambda 0.1/1/10 x seeds 0-3, 500 iterations)")
    print(f"{'FIR arithmetic':<12} {'diff_z':>10} {'z':>10} {'x':>10}   max|z| (scaled units)")
    for m, a, b, c, zm in rows:
        print(f"{m:<12} {a:10.2e} {b:10.2e} {c:10.2e}   {zm:8.1f}")


if __name__ == "__main__":
    main()
