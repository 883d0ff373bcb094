// Register-resident fused FISTA kernel, "pair" form: TWO problems per 16-lane DPP
// row, every per-sample quantity held as a float2 (problem A, problem B).
//
// Same recurrence, data layout and numerics as fista_fast.h (float64 iterate and
// update, float32 scans/FIR); what changes is the packing of v_pk_fma_f32:
//   fista_fast.h  packs over TAPS   acc = (sum even taps, sum odd taps) -> needs a
//                 combine add per output and a shifted tap copy for alignment;
//   here          packs over VOXELS acc = (x_A[j], x_B[j]) += h[m] * (z_A, z_B)[q-m]
//                 with the tap broadcast from an SGPR (op_sel): no combine, no
//                 alignment cases, scan chains and offsets are packed adds.
// Instruction budget per voxel-iteration at S=19, K=30: ~800 against ~905.
// Used for solves without a stop rule and with shared taps (cost trace optional); the
// other modes stay on fista_fast.h.  An odd problem count pads the last row with a copy
// of the last problem (computed, never stored).
#pragma once
#include "common.h"
#include "fista_fast.h"

namespace pb {

// taps as aligned pairs (h[2i], h[2i+1]); tap m is broadcast to both problems with an
// op_sel swizzle of pair m/2, so the 30 taps cost 30 SGPRs
template <int KT>
struct TapsF {
  static constexpr int NP = (KT + 1) / 2;
  f2 pr[NP];
};

template <int KT>
inline TapsF<KT> make_taps_f(const double* taps, int K) {
  TapsF<KT> t;
  auto h = [&](int m) -> float { return (m < K) ? (float)taps[m] : 0.0f; };
  for (int i = 0; i < TapsF<KT>::NP; ++i) t.pr[i] = f2{h(2 * i), h(2 * i + 1)};
  return t;
}

template <int CTRL>
__device__ __forceinline__ f2 dpp_zero2(f2 v) {
  return f2{dpp_zero<CTRL>(v.x), dpp_zero<CTRL>(v.y)};
}

// SKIP0: tap 0 is exactly zero (true of every SPM HRF: the gamma densities vanish at
// t = 0, pybold/hrf_model.py:25-31) and is left out of both FIRs at compile time.
template <int S, int KT, bool WITH_J = false, bool SKIP0 = false>
__global__ __launch_bounds__(256, 2) void fista_pair_kernel(FistaArgs a, TapsF<KT> taps) {
  constexpr int H = KT - 1;
  constexpr int D = (H + S - 1) / S;
  constexpr int G = 5;                      // outputs per accumulator group
  static_assert(D <= 15, "halo spans more than one DPP row");

  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int sub = threadIdx.x & 15;
  const int row = gid >> 4;
  const int base = sub * S;
  const int pA0 = 2 * row + a.p0, pB0 = 2 * row + 1 + a.p0;
  const bool liveA = pA0 < a.P, liveB = pB0 < a.P;
  const int pA = liveA ? pA0 : a.P - 1;
  const int pB = liveB ? pB0 : a.P - 1;

  double wA[S], wB[S];
  // LDS, per 16-lane row: [S*16] float2 then [S*16] float.  During the solve it holds y
  // (both problems) and the padding mask, which two problems per row leave no registers
  // for: each lane re-reads only what it wrote itself (one ds_read_b64 + ds_read_b32 per
  // sample and iteration, on the otherwise idle LDS pipe).  In the prologue and epilogue
  // the same space transposes rows between the coalesced global layout (lane = sample mod
  // 16: every cache line is requested once) and the strip layout of the solver (lane owns
  // S consecutive samples); with 8 rows in flight per wave, strided per-lane global
  // accesses overflow the 4 MB L2 and re-fetch lines (measured 4x the bytes).  A row is
  // only ever touched by its own 16 lanes, i.e. by one wave: LDS operations of a wave
  // execute in order, so wave-level fences are enough and no workgroup barrier is needed.
  extern __shared__ __attribute__((aligned(16))) char pair_smem[];
  const int rslot = (threadIdx.x >> 4) * S * 16;
  f2* ly = reinterpret_cast<f2*>(pair_smem) + (rslot + sub);
  float* lm = reinterpret_cast<float*>(pair_smem + (size_t)16 * S * 16 * sizeof(f2)) + (rslot + sub);
  double* stage_d = reinterpret_cast<double*>(reinterpret_cast<f2*>(pair_smem) + rslot);
  float* stage_f = reinterpret_cast<float*>(pair_smem + (size_t)16 * S * 16 * sizeof(f2)) + rslot;
  // cost trace only: per-row scratch {lbda_A, lbda_B, ||w_A||_1, ||w_B||_1} behind the mask
  // region, so that none of them is live in registers across the FIRs (the register peak)
  float* lj = reinterpret_cast<float*>(pair_smem + (size_t)16 * S * 16 * (sizeof(f2) + sizeof(float))) +
              (threadIdx.x >> 4) * 4;
  auto lds_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  {
    const float* yA = a.y + (int64_t)(pA / a.y_rep) * a.ldy;
    const float* yB = a.y + (int64_t)(pB / a.y_rep) * a.ldy;
    const double* rA = a.w + (int64_t)pA * a.ldw;
    const double* rB = a.w + (int64_t)pB * a.ldw;
    // w of A, then of B: coalesced read -> LDS (natural order) -> strips
    auto load_w = [&](const double* row, double* strip) {
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const int i = k * 16 + sub;
        stage_d[i] = (i < a.N && !a.cold) ? row[i] : 0.0;
      }
      lds_sync();
#pragma unroll
      for (int j = 0; j < S; ++j) strip[j] = stage_d[base + j];
      lds_sync();
    };
    load_w(rA, wA);
    load_w(rB, wB);
    // y of A and B: the float region stages one row at a time, the float2 region gets
    // the final (A, B) pairs in strip layout
    float ya[S], yb[S];
    auto load_y = [&](const float* row, float* strip) {
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const int i = k * 16 + sub;
        stage_f[i] = (i < a.N) ? row[i] : 0.0f;
      }
      lds_sync();
#pragma unroll
      for (int j = 0; j < S; ++j) strip[j] = stage_f[base + j];
      lds_sync();
    };
    load_y(yA, ya);
    load_y(yB, yb);
#pragma unroll
    for (int j = 0; j < S; ++j) {
      ly[j * 16] = f2{ya[j], yb[j]};
      lm[j * 16] = (base + j < a.N) ? 1.0f : 0.0f;
    }
  }
  const double lbA = a.lbda_vec ? a.lbda_vec[pA] : a.lbda;
  const double lbB = a.lbda_vec ? a.lbda_vec[pB] : a.lbda;
  const double thA = lbA * a.step, thB = lbB * a.step;
  const double nstep = -a.step;

  if constexpr (WITH_J) {
    lj[0] = (float)lbA;
    lj[1] = (float)lbB;
  }
  // ---- forward pass: r = h * cumsum(w) - y for both problems ------------------------
  auto forward = [&](f2 (&r)[S]) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");          // keep the LDS reads inside the loop
    // ---- z = cumsum(w) for both problems -----------------------------------
    f2 z[S];
    z[0] = f2{(float)wA[0], (float)wB[0]};
    if constexpr (WITH_J) {
      // ||w||_1 of the iterate this pass starts from (|.| folds into the add as a source
      // modifier); reduced over the row now and parked in LDS until the residual is known
      float l1a = fabsf(z[0].x), l1b = fabsf(z[0].y);
#pragma unroll
      for (int j = 1; j < S; ++j) {
        const f2 wj = f2{(float)wA[j], (float)wB[j]};
        z[j] = z[j - 1] + wj;
        l1a += fabsf(wj.x);
        l1b += fabsf(wj.y);
      }
      lj[2] = row_allsum(l1a);
      lj[3] = row_allsum(l1b);
    } else {
#pragma unroll
      for (int j = 1; j < S; ++j) z[j] = z[j - 1] + f2{(float)wA[j], (float)wB[j]};
    }
    {
      const f2 off = f2{row_from_below<1>(row_prefix_incl(z[S - 1].x)),
                        row_from_below<1>(row_prefix_incl(z[S - 1].y))};
#pragma unroll
      for (int j = 0; j < S; ++j) z[j] += off;
    }

    // ---- window of z: halo [0, H) from the lanes below, own samples [H, H+S) ----
    f2 Z[H + S];
    static_for<0, S>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      Z[H + j] = z[j];
    });
    static_for<1, D + 1>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      static_for<0, S>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int e = H - d * S + j;
        if constexpr (e >= 0) Z[e] = dpp_zero2<DPP_ROW_SHR + d>(z[j]);
      });
    });

    // ---- r = h * z - y ------------------------------------------------------
    // Outputs are produced in groups of G with the tap loop outside, so that G
    // independent accumulator chains sit next to each other in program order (a
    // dependent v_pk_fma_f32 straight after its producer costs a wait state).
    f2 ypre[G];                               // -y and mask of the group about to start
    float mpre[G];
#pragma unroll
    for (int q = 0; q < G; ++q) {
      ypre[q] = ly[(q < S ? q : S - 1) * 16];
      mpre[q] = lm[(q < S ? q : S - 1) * 16];
    }
    static_for<0, (S + G - 1) / G>([&](auto gc) {
      constexpr int j0 = decltype(gc)::value * G;
      constexpr int gn = (S - j0 < G) ? S - j0 : G;
      f2 acc[gn];
      float mcur[gn];
#pragma unroll
      for (int q = 0; q < gn; ++q) {
        acc[q] = -ypre[q];
        mcur[q] = mpre[q];
      }
      if constexpr (j0 + G < S) {             // issue the next group's LDS reads now
#pragma unroll
        for (int q = 0; q < G; ++q) {
          const int jn = (j0 + G + q < S) ? j0 + G + q : S - 1;
          ypre[q] = ly[jn * 16];
          mpre[q] = lm[jn * 16];
        }
      }
      static_for<(SKIP0 ? 1 : 0), KT>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const f2 tp = taps.pr[m / 2];
        const f2 tb = (m % 2 == 0) ? tp.xx : tp.yy;
        static_for<0, gn>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          acc[q] = __builtin_elementwise_fma(tb, Z[H + j0 + q - m], acc[q]);
        });
      });
#pragma unroll
      for (int q = 0; q < gn; ++q) r[j0 + q] = acc[q] * f2{mcur[q], mcur[q]};
    });
  };

  // ---- adjoint pass and update: w <- prox step from the residual r -------------------
  auto backward = [&](const f2 (&r)[S], const double beta) __attribute__((always_inline)) {
    // ---- window of r: own samples [0, S), halo [S, S+H) from the lanes above ----
    f2 R[S + H];
    static_for<0, S>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      R[j] = r[j];
    });
    static_for<1, D + 1>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      static_for<0, S>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int e = d * S + j;
        if constexpr (e < S + H) R[e] = dpp_zero2<DPP_ROW_SHL + d>(r[j]);
      });
    });

    // ---- g = reverse-cumsum(K^T r) -------------------------------------------
    f2 g[S];
    static_for<0, (S + G - 1) / G>([&](auto gc) {
      constexpr int j0 = decltype(gc)::value * G;
      constexpr int gn = (S - j0 < G) ? S - j0 : G;
      f2 acc[gn];
#pragma unroll
      for (int q = 0; q < gn; ++q) acc[q] = f2{0.f, 0.f};
      static_for<(SKIP0 ? 1 : 0), KT>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const f2 tp = taps.pr[m / 2];
        const f2 tb = (m % 2 == 0) ? tp.xx : tp.yy;
        static_for<0, gn>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          acc[q] = __builtin_elementwise_fma(tb, R[j0 + q + m], acc[q]);
        });
      });
#pragma unroll
      for (int q = 0; q < gn; ++q) g[j0 + q] = acc[q];
    });
#pragma unroll
    for (int j = S - 2; j >= 0; --j) g[j] += g[j + 1];
    {
      const f2 off = f2{row_from_above<1>(row_suffix_incl(g[0].x)),
                        row_from_above<1>(row_suffix_incl(g[0].y))};
#pragma unroll
      for (int j = 0; j < S; ++j) g[j] += off;
    }

    // ---- gradient step, prox, momentum (float64), per problem ------------------
    // staged over all samples so that no instruction directly follows its producer
    // (program order is kept in this translation unit)
    const double nb1 = -(1.0 + beta);
    {
      double uA[S], uB[S], dA[S], dB[S];
#pragma unroll
      for (int j = 0; j < S; ++j) {
        uA[j] = fma(nstep, (double)g[j].x, wA[j]);
        uB[j] = fma(nstep, (double)g[j].y, wB[j]);
      }
#pragma unroll
      for (int j = 0; j < S; ++j) {
        dA[j] = fmax(uA[j], -thA);
        dB[j] = fmax(uB[j], -thB);
      }
#pragma unroll
      for (int j = 0; j < S; ++j) {
        dA[j] = fmin(dA[j], thA);
        dB[j] = fmin(dB[j], thB);
      }
#pragma unroll
      for (int j = 0; j < S; ++j) {
        wA[j] = fma(nb1, dA[j], uA[j]);
        wB[j] = fma(nb1, dB[j], uB[j]);
      }
    }
  };

  // This translation unit keeps program order (Makefile: PAIRFLAGS), so loads are placed by
  // hand well ahead of their use: the momentum factor at the top of an iteration, y and the
  // mask one output group ahead inside the FIR.
  if constexpr (!WITH_J) {
    for (int it = 0; it < a.n_iter; ++it) {
      const double beta = a.betas[it];
      f2 r[S];
      forward(r);
      backward(r, beta);
    }
  } else {
    // cost of iterate k+1 = residual of the forward pass of iteration k+1: the loop is
    // rotated (forward pass at the bottom, one peeled in front) so that it has a single
    // exit and the iterate stays in the same registers around the back edge
    f2 r[S];
    forward(r);
    for (int it = 0; it < a.n_iter; ++it) {
      const double beta = a.betas[it];
      backward(r, beta);
      forward(r);
      f2 sq = r[0] * r[0];                      // ||r||^2 of the new iterate
#pragma unroll
      for (int j = 1; j < S; ++j) sq = __builtin_elementwise_fma(r[j], r[j], sq);
      const float cA = fmaf(0.5f, row_allsum(sq.x), lj[0] * lj[2]);
      const float cB = fmaf(0.5f, row_allsum(sq.y), lj[1] * lj[3]);
      if (sub == 0) {
        if (liveA) a.J[(int64_t)pA * a.ldj + it] = cA;
        if (liveB) a.J[(int64_t)pB * a.ldj + it] = cB;
      }
    }
  }

  // epilogue: strips -> LDS -> coalesced stores (the y/mask contents are dead now)
  auto store_w = [&](const double* strip, double* row, bool live) {
    lds_sync();
#pragma unroll
    for (int j = 0; j < S; ++j) stage_d[base + j] = strip[j];
    lds_sync();
    if (live) {
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const int i = k * 16 + sub;
        if (i < a.N) row[i] = stage_d[i];
      }
    }
  };
  store_w(wA, a.w + (int64_t)pA * a.ldw, liveA);
  store_w(wB, a.w + (int64_t)pB * a.ldw, liveB);
  if (a.n_done && sub == 0) {
    if (liveA) a.n_done[pA] = a.n_iter;
    if (liveB) a.n_done[pB] = a.n_iter;
  }
}

template <int S, int KT>
int launch_pair(const FistaArgs& a, const double* taps, int K, bool with_j, hipStream_t st) {
  const auto tf = make_taps_f<KT>(taps, K);
  const int64_t rows = ((int64_t)(a.P - a.p0) + 1) / 2;
  const dim3 grid((unsigned)((rows * 16 + 255) / 256)), block(256);
  const size_t lds = (size_t)16 * S * 16 * (sizeof(f2) + sizeof(float)) + (with_j ? 16 * 4 * sizeof(float) : 0);
  const bool skip0 = KT > 1 && tf.pr[0].x == 0.0f;      // leading tap exactly zero
  if (with_j && skip0) hipLaunchKernelGGL((fista_pair_kernel<S, KT, true, true>), grid, block, lds, st, a, tf);
  else if (with_j) hipLaunchKernelGGL((fista_pair_kernel<S, KT, true, false>), grid, block, lds, st, a, tf);
  else if (skip0) hipLaunchKernelGGL((fista_pair_kernel<S, KT, false, true>), grid, block, lds, st, a, tf);
  else hipLaunchKernelGGL((fista_pair_kernel<S, KT, false, false>), grid, block, lds, st, a, tf);
  return 0;
}

}  // namespace pb
