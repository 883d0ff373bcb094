// Register-resident fused FISTA kernel in float64 END TO END (y, scans, FIRs, iterate, cost
// trace, stop rules): the reference's own arithmetic, for the calls where its decisions
// matter to the last digit -- the 1-D calls of the API and the noise-driven lambda search,
// whose stop rule sits on a knife edge (Numerics).
//
// One problem per wave64, S consecutive samples per lane (S = 5: series of up to 320 scans;
// S = 10: 640), everything in VGPRs:
//   z = cumsum(w)          lane-local prefix + wave scan (DPP row_shr / row_bcast on both halves)
//   x = h * z              K-tap causal FIR, v_fma_f64 (full rate on gfx950); the K-1 halo
//                          samples come from the lanes below through wave_shr:1 chains
//   g = revcumsum(K^T r)   halo from the lanes above, suffix = total - inclusive prefix
//   u = w - s g ; w = u - (1+beta) clamp(u, -th, th)
// Cost trace and both stop rules as in fista_fast.h, but the window rule (wind = 6) keeps
// u_{k-1} and the last three increments in float64 REGISTERS (S is small), so its criterion
// is float64 too.  ~530 instructions per problem-iteration: a 500-iteration single-voxel
// solve takes ~0.5 ms instead of 4 ms on the LDS kernel (generic.h), a machine-filling batch
// runs at ~6x its rate.
//
// Reference: pybold/bold_signal.py:62-97 (deconv), :259-276 (_loops_deconv),
// pybold/linear.py:73-113, pybold/convolution.py:105-132.
#pragma once
#include "../../include/pybold_hip.h"
#include "common.h"
#include "fista_fast.h"

namespace pb {

template <int KT>
struct TapsD {
  double h[KT];
};

template <int KT>
inline TapsD<KT> make_taps_d(const double* taps, int K) {
  TapsD<KT> t;
  for (int m = 0; m < KT; ++m) t.h[m] = (m < K) ? taps[m] : 0.0;
  return t;
}

template <int CTRL, int ROW_MASK = 0xf, bool BOUND = true>
__device__ __forceinline__ double dpp_f64(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, BOUND);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, BOUND);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// inclusive prefix sum over the 64 lanes of the wave
__device__ __forceinline__ double wave_prefix_incl_f64(double t) {
  t += dpp_f64<DPP_ROW_SHR + 1>(t);
  t += dpp_f64<DPP_ROW_SHR + 2>(t);
  t += dpp_f64<DPP_ROW_SHR + 4>(t);
  t += dpp_f64<DPP_ROW_SHR + 8>(t);
  t += dpp_f64<DPP_ROW_BCAST15, 0xa, false>(t);     // rows 1, 3 += total of the row before
  t += dpp_f64<DPP_ROW_BCAST31, 0xc, false>(t);     // rows 2, 3 += total of the first half
  return t;
}

__device__ __forceinline__ double readlane_f64(double v, int lane);

// sum over the lanes ABOVE this one, built from additions only (in-row suffix scan + the totals of the rows above):
// where every lane above holds 0 the result is EXACTLY 0, as in the reference's flip-cumsum-flip
// (pybold/linear.py:43).  `total - inclusive prefix` leaves a rounding residue of ~1e-16 |total| there instead, and
// the last sample of a series has gradient exactly 0 (every SPM HRF has h[0] = 0): with a NEGATIVE threshold the
// reference's prox is discontinuous at 0 (sign(u) |th|), so that residue grew into a 1e-3 error (round 5).
__device__ __forceinline__ double wave_suffix_excl_f64(double v) {
  double s = v;
  s += dpp_f64<DPP_ROW_SHL + 1>(s);
  s += dpp_f64<DPP_ROW_SHL + 2>(s);
  s += dpp_f64<DPP_ROW_SHL + 4>(s);
  s += dpp_f64<DPP_ROW_SHL + 8>(s);                 // lane 0 of a row: the row's total
  const double t3 = readlane_f64(s, 48), t2 = readlane_f64(s, 32), t1 = readlane_f64(s, 16);
  const int row = (threadIdx.x & 63) >> 4;
  const double t23 = t2 + t3;
  const double above = row == 3 ? 0.0 : (row == 2 ? t3 : (row == 1 ? t23 : t1 + t23));
  return dpp_f64<DPP_ROW_SHL + 1>(s) + above;        // lanes above within the row (0 for the row's last lane) + rows above
}

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)b, lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

template <int S, int KT, bool WITH_J, int STOP>
__global__ __launch_bounds__(256) void fista_exact_kernel(FistaArgs a, TapsD<KT> taps) {
  constexpr int H = KT - 1;
  constexpr int D = (H + S - 1) / S;        // neighbour lanes that contribute halo
  constexpr int WIND = 6;
  static_assert(D <= 63, "halo spans more than the wave");

  const int lane = threadIdx.x & 63;
  // slots of this launch's list (fista_fast.h: launch_slots; the batch itself without a device-side list).  Round 5: the
  // ill-conditioned series of a partitioned float32 call run here (y float32 in HBM, a.y64 == nullptr; cost trace to a.J)
  int s0, s1;
  launch_slots(a, s0, s1);
  const int slot = (int)((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) + s0;
  if (slot >= s1 && a.range) return;               // (one problem per wave: uniform; an empty candidate launch leaves here)
  bool live;
  const int p = slot_to_problem(a, slot, s1, live);
  const int base = lane * S;

  double y[S], w[S], mk[S];
  {
    const double* yrow = a.y64 ? a.y64 + (int64_t)(p / a.y_rep) * a.ldy : nullptr;
    const float* yrow32 = a.y64 ? nullptr : a.y + (int64_t)(p / a.y_rep) * a.ldy;
    const double* wrow = a.w + (int64_t)p * a.ldw;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const bool ok = base + j < a.N;
      y[j] = ok ? (yrow ? yrow[base + j] : (double)yrow32[base + j]) : 0.0;
      w[j] = (ok && !a.cold) ? wrow[base + j] : 0.0;
      mk[j] = ok ? 1.0 : 0.0;
    }
  }
  const double lb = a.lbda_vec ? a.lbda_vec[p] : a.lbda;
  const double th = lb * a.step;
  const double nstep = -a.step;

  // window rule state (wind = 6): u_{k-1} and the increments delta_{k-1}, delta_{k-2}, delta_{k-3}
  double uprev[STOP == 2 ? S : 1], d1[STOP == 2 ? S : 1], d2[STOP == 2 ? S : 1], d3[STOP == 2 ? S : 1];
  if constexpr (STOP == 2) {
#pragma unroll
    for (int j = 0; j < S; ++j) uprev[j] = d1[j] = d2[j] = d3[j] = 0.0;
  }
  bool active = live;
  int done = 0;
  double* Jrow = (WITH_J && a.J64) ? a.J64 + (int64_t)p * a.ldj : nullptr;
  float* Jrow32 = (WITH_J && !a.J64 && a.J) ? a.J + (int64_t)p * a.ldj : nullptr;

  int n_stop = a.n_iter;
  for (int it = 0;; ++it) {
    if (!WITH_J && it >= n_stop) break;
    // ---- z = cumsum(w) ----------------------------------------------------
    double z[S];
    z[0] = w[0];
#pragma unroll
    for (int j = 1; j < S; ++j) z[j] = z[j - 1] + w[j];
    {
      const double off = dpp_f64<DPP_WAVE_SHR1>(wave_prefix_incl_f64(z[S - 1]));
#pragma unroll
      for (int j = 0; j < S; ++j) z[j] += off;
    }
    // ---- window of z: own samples at [H, H+S), halo below -------------------
    double Z[H + S];
    static_for<0, S>([&](auto jc) { Z[H + decltype(jc)::value] = z[decltype(jc)::value]; });
    {
      double sh[S];
#pragma unroll
      for (int j = 0; j < S; ++j) sh[j] = z[j];
      static_for<1, D + 1>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
#pragma unroll
        for (int j = 0; j < S; ++j) sh[j] = dpp_f64<DPP_WAVE_SHR1>(sh[j]);      // lane - d
        static_for<0, S>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int e = H - d * S + j;
          if constexpr (e >= 0) Z[e] = sh[j];
        });
      });
    }
    // ---- r = h * z - y -------------------------------------------------------
    double r[S];
    static_for<0, S>([&](auto jc) { r[decltype(jc)::value] = -y[decltype(jc)::value]; });
    static_for<0, KT>([&](auto mc) {          // tap-major: S independent chains
      constexpr int m = decltype(mc)::value;
      static_for<0, S>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        r[j] = fma(taps.h[m], Z[H + j - m], r[j]);
      });
    });
#pragma unroll
    for (int j = 0; j < S; ++j) r[j] *= mk[j];

    // ---- cost of the iterate this pass started from -------------------------
    if constexpr (WITH_J) {
      if (it > 0) {
        double sq = 0.0, l1 = 0.0;
#pragma unroll
        for (int j = 0; j < S; ++j) {
          sq = fma(r[j], r[j], sq);
          l1 += fabs(w[j]);
        }
        const double cost = seg_allsum_f64<64>(fma(0.5, sq, lb * l1));
        if (live && lane == 0 && (STOP == 0 || it <= done)) {
          if (Jrow) Jrow[it - 1] = cost;
          else if (Jrow32) Jrow32[it - 1] = (float)cost;
        }
      }
      if (it >= n_stop) break;
    }

    // ---- window of r: own samples at [0, S), halo above ----------------------
    double R[S + H];
    static_for<0, S>([&](auto jc) { R[decltype(jc)::value] = r[decltype(jc)::value]; });
    {
      double sh[S];
#pragma unroll
      for (int j = 0; j < S; ++j) sh[j] = r[j];
      static_for<1, D + 1>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
#pragma unroll
        for (int j = 0; j < S; ++j) sh[j] = dpp_f64<DPP_WAVE_SHL1>(sh[j]);      // lane + d
        static_for<0, S>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          constexpr int e = d * S + j;
          if constexpr (e < S + H) R[e] = sh[j];
        });
      });
    }
    // ---- g = revcumsum(K^T r) ---------------------------------------------------
    double g[S];
#pragma unroll
    for (int j = 0; j < S; ++j) g[j] = 0.0;
    static_for<0, KT>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      static_for<0, S>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        g[j] = fma(taps.h[m], R[j + m], g[j]);
      });
    });
#pragma unroll
    for (int j = S - 2; j >= 0; --j) g[j] += g[j + 1];
    {
      const double off = wave_suffix_excl_f64(g[0]);         // sum of the lanes above, exactly 0 above the last sample
#pragma unroll
      for (int j = 0; j < S; ++j) g[j] += off;
    }

    // ---- gradient step, prox, momentum; stop rules ------------------------------
    const double beta = a.betas[it];
    const double nb1 = -(1.0 + beta);
    if constexpr (STOP == 0) {
#pragma unroll
      for (int j = 0; j < S; ++j) {
        const double u = fma(nstep, g[j], w[j]);
        const double d = prox_excess_ref(u, th);
        w[j] = fma(nb1, d, u);
      }
    } else {
      double num = 0.0, den = 0.0, floor_eps = 1.0e-10;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        const double u = fma(nstep, g[j], w[j]);
        const double d = prox_excess_ref(u, th);
        const double wn = fma(nb1, d, u);
        if constexpr (STOP == 1) {            // _loops_deconv rule (pybold/bold_signal.py:267-273)
          const double diff = wn - u;
          num = fma(diff, diff, num);
          den = fma(wn, wn, den);
        } else {
          // deconv window rule, wind = 6 (:82-95), on [u_{k-4} .. u_k, w_{k+1}]:
          //   3 (new - old) = delta_{k-3} + 2 delta_{k-2} + 3 delta_{k-1} + 2 delta_k + e
          //   3 new         = 3 u_k - delta_k + e          (see fista_fast.h)
          const double dk = u - uprev[j];
          const double e = wn - u;
          const double diff = fma(2.0, dk, fma(3.0, d1[j], fma(2.0, d2[j], d3[j]))) + e;
          const double sn = fma(3.0, u, e - dk);
          num = fma(diff, diff, num);
          den = fma(sn, sn, den);
          d3[j] = d2[j];
          d2[j] = d1[j];
          d1[j] = dk;
          uprev[j] = u;
        }
        w[j] = wn;
      }
      if constexpr (STOP == 2) floor_eps = 3.0e-10;
      num = seg_allsum_f64<64>(num);
      den = seg_allsum_f64<64>(den);
      if (active) {
        done = it + 1;
        constexpr int first_test = (STOP == 1) ? 3 : WIND + 1;
        if (it >= first_test && sqrt(num) / (sqrt(den) + floor_eps) < a.tol) {
          active = false;                     // wave-uniform: one problem per wave
          n_stop = it + 1;
        }
      }
    }
  }

  if (live) {
    double* wrow = a.w + (int64_t)p * a.ldw;
#pragma unroll
    for (int j = 0; j < S; ++j)
      if (base + j < a.N) wrow[base + j] = w[j];
    if (a.n_done && lane == 0) a.n_done[p] = (STOP == 0) ? a.n_iter : done;
  }
}

template <int S, int KT>
int launch_exact(const FistaArgs& a, const double* taps, int K, bool with_j, int stop, hipStream_t st) {
  const auto td = make_taps_d<KT>(taps, K);
  const dim3 grid((unsigned)((launch_count(a) + 3) / 4)), block(256);
  if (stop == PB_STOP_NONE) {
    if (with_j) hipLaunchKernelGGL((fista_exact_kernel<S, KT, true, 0>), grid, block, 0, st, a, td);
    else hipLaunchKernelGGL((fista_exact_kernel<S, KT, false, 0>), grid, block, 0, st, a, td);
  } else if (stop == PB_STOP_LOOPS) {
    if (with_j) hipLaunchKernelGGL((fista_exact_kernel<S, KT, true, 1>), grid, block, 0, st, a, td);
    else hipLaunchKernelGGL((fista_exact_kernel<S, KT, false, 1>), grid, block, 0, st, a, td);
  } else {
    if (with_j) hipLaunchKernelGGL((fista_exact_kernel<S, KT, true, 2>), grid, block, 0, st, a, td);
    else hipLaunchKernelGGL((fista_exact_kernel<S, KT, false, 2>), grid, block, 0, st, a, td);
  }
  return 0;
}

}  // namespace pb
