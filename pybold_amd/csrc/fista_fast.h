// Register-resident fused FISTA kernel for gfx950 (MI355X).
//
// One problem (voxel, lambda) per 16-lane DPP row, S = ceil(N/16) consecutive
// samples per lane, four problems per wave64, every iteration of the
// recurrence executed inside one launch with all state in VGPRs:
//
//   z = cumsum(w)            lane-local prefix + 4-step row scan (DPP row_shr)
//   x = h * z                K-tap causal FIR; the K-1 halo samples come from
//                            the lanes below through DPP row_shr with
//                            zero fill = the Toeplitz zero padding
//   r = x - y
//   c = K^T r                halo from the lanes above (DPP row_shl)
//   g = reverse-cumsum(c)    lane-local suffix + row scan (DPP row_shl)
//   u = w - s g ; w = u - (1+beta) clamp(u, -th, th)      (float64)
//
// Reference: pybold/bold_signal.py:62-72 (deconv), :259-276 (_loops_deconv),
// pybold/linear.py:73-113 (H.op / H.adj), pybold/convolution.py:105-132.
//
// Arithmetic: the iterate w and its update are float64 (2 VGPRs per sample);
// scans, FIR and residual are float32.  Measured on the golden inputs this
// gives 8e-8 relative L2 error on diff_z after 500 iterations, against
// 1.1e-5 for an all-float32 state (DESIGN.md "Numerics").
//
// The two FIRs are ~80 % of the instruction stream; every VALU instruction on
// gfx950 costs 4 cycles per wave64 and v_pk_fma_f32 retires two FMAs in one,
// so they are written on float2 pairs.  Packing is over TAPS, not outputs:
//   acc.lo += h[m] * Z[q],  acc.hi += h[m+1] * Z[q-1]
// with the tap pair in an SGPR pair (kernel argument) and the window pair an
// aligned VGPR pair swapped by op_sel; outputs whose pair would be misaligned
// use a second, one-tap-shifted copy of the taps (loop-invariant, held in VGPRs:
// two SGPR copies would spill the scalar file).
#pragma once
#include "common.h"

namespace pb {

struct FistaArgs {
  const float* y;         // [ceil(P/y_rep)][ldy]
  const double* y64;      // the same in float64 (all-float64 LDS kernel only) or nullptr
  int64_t ldy;
  double* w;              // [P][ldw] in: warm start, out: final iterate
  int64_t ldw;
  const double* lbda_vec; // [P] or nullptr
  const double* betas;    // [n_iter]
  float* J;               // [P][ldj] or nullptr
  double* J64;            // float64 cost trace (all-float64 LDS kernel only) or nullptr
  int64_t ldj;
  int32_t* n_done;        // [P] or nullptr
  const double* taps_pp;  // [P][ldt] per-problem HRF taps (K of them) or nullptr
  int64_t ldt;
  const double* step_vec; // [P] per-problem step (goes with taps_pp) or nullptr
  int step_shared;        // 1: taps_pp / step_vec hold ONE HRF / step used by every problem
  double step;
  double lbda;
  double tol;
  int y_rep;
  int P;                  // problems [p0, P) belong to this launch (row indices are global)
  int p0;
  int N;
  int n_iter;
  int stop_mode;
  int K;                  // number of taps actually used (<= KT)
  int cold;               // 1: the iterate starts from 0, a.w is written only
  int wind = 6;           // window rule: stored iterates (register-resident forms: 4, 6 or 8)
  int ybits = 14;         // matrix-pipe form: every series is scaled so that max |y| lies in [2^(ybits-1), 2^ybits)
  int rho_guard = 1;      // matrix-pipe form: hand sparse solutions (th / max|w| > MFMA_RHO_MAX) back to the vector forms
  int only_flagged = 0;   // 1: solve only the problems with n_done[p] < 0 (left by the certificate
                          //    form of the pair kernel, fista_pair_ffa.h), skip the others
  // Lists built on the device (path.h): perm[0 .. *n_dense) the front class in ascending order, the middle class behind
  // it, ill-conditioned problems descending from perm[P-1].  perm_side = 1 (and 3): slot s -> problem perm[s]; 2: the
  // middle class on its own (slot s -> perm[*n_dense + s]); counts and ranges are read on the device, so a grid covers a
  // host bound and the waves beyond the list leave at once.  0: slot = problem.
  const int32_t* perm = nullptr;
  const int32_t* n_dense = nullptr;
  int perm_side = 0;
  // Device-side plan (round 5, plan.h): range[0], range[1] = the slots [s0, s1) of its list this launch solves, read on
  // the device; the grid covers `grid_slots` slots (the worst case the host can bound) and the waves beyond s1 - s0
  // leave at once.  nullptr: the slots [p0, length of the list) as before.
  const int32_t* range = nullptr;
  int grid_slots = 0;
};

// problems the host sizes the grid for
inline int64_t launch_count(const FistaArgs& a) { return a.range ? (int64_t)a.grid_slots : (int64_t)(a.P - a.p0); }

// slots of this launch's list (the whole batch without a partition)
__device__ __forceinline__ int list_length(const FistaArgs& a) {
  if (a.perm_side == 0) return a.P;
  if (a.perm_side == 3) return a.P;              // (a list of its own, e.g. the handed-back problems: bounded by `range`)
  const int nd = *a.n_dense;
  return a.perm_side == 1 ? nd : a.P - nd;
}
// first slot and end of the slots this launch solves
__device__ __forceinline__ void launch_slots(const FistaArgs& a, int& s0, int& s1) {
  if (a.range) {
    s0 = a.range[0];
    s1 = a.range[1];
  } else {
    s0 = a.p0;
    s1 = list_length(a);
  }
}
// slot -> problem; `live` = the slot is inside the list (a dead slot maps to a valid problem whose data may be read)
__device__ __forceinline__ int slot_to_problem(const FistaArgs& a, int slot, int n_list, bool& live) {
  live = slot < n_list;
  if (a.perm_side == 0) return live ? slot : a.P - 1;
  return live ? a.perm[a.perm_side == 2 ? *a.n_dense + slot : slot] : 0;     // (sides 1 and 3: positions of the list array)
}

// Tap pairs as kernel arguments (read with scalar loads, kept in SGPRs).
//   even[a] = (h[2a],   h[2a+1])
//   odd[a]  = (h[2a-1], h[2a])        with h[-1] = h[>=K] = 0
template <int KT>
struct TapPairs {
  static constexpr int NE = (KT + 1) / 2;
  static constexpr int NO = KT / 2 + 1;
  f2 even[NE];
  f2 odd[NO];
};

template <int KT>
inline TapPairs<KT> make_tap_pairs(const double* taps, int K) {
  TapPairs<KT> t;
  auto h = [&](int m) -> float { return (m >= 0 && m < K) ? (float)taps[m] : 0.0f; };
  for (int a = 0; a < TapPairs<KT>::NE; ++a) t.even[a] = f2{h(2 * a), h(2 * a + 1)};
  for (int a = 0; a < TapPairs<KT>::NO; ++a) t.odd[a] = f2{h(2 * a - 1), h(2 * a)};
  return t;
}

// Window of H halo samples + S own samples (+ padding), stored as aligned pairs.
template <int NPAIR>
struct Window {
  f2 p[NPAIR];
  template <int E>
  __device__ __forceinline__ void set(float v) {
    static_assert(E >= 0 && E < 2 * NPAIR, "window index");
    if constexpr (E % 2 == 0) p[E / 2].x = v; else p[E / 2].y = v;
  }
};

// PP = per-problem taps and step (blind deconvolution with one HRF per voxel):
// both tap-pair copies are then loaded from a.taps_pp into VGPRs.
// LPV = lanes per problem: 16 (one DPP row, four problems per wave; series up to 16*S
// scans) or 64 (one problem per wave for long series, up to 64*S scans: halo through
// wave_shr/shl:1 chains, scans with the row_bcast steps).
// Weights of the window rule for `wind` = W stored iterates [u_{k-W+2} .. u_k, w_{k+1}]
// (pybold/bold_signal.py:82-95: old = mean of the first W - W/2, new = mean of the last W/2).
// With delta_i = u_i - u_{i-1}, e = w_{k+1} - u_k, NN = W/2, NO = W - NN:
//   NN (new - old) = sum_{t=0}^{W-3} num(t) delta_{k-t} + e,   num(t) = c_t NN / NO - a_t
//   NN new         = NN u_k + sum_t den(t) delta_{k-t} + e,    den(t) = -a_t
//   a_t = max(NN - 2 - t, 0),  c_t = W - 1 - max(NN - 1, t + 1)
// (W = 6: 2, 3, 2, 1 and -1, 0, 0, 0).  W - 2 increments are live: the newest plus W - 3 stored.
template <int W>
struct WinCoef {
  static constexpr int NN = W / 2, NO = W - W / 2, NR = W - 2;
  static constexpr int a(int t) { return NN - 2 - t > 0 ? NN - 2 - t : 0; }
  static constexpr int c(int t) { return W - 1 - (NN - 1 > t + 1 ? NN - 1 : t + 1); }
  static constexpr float num(int t) { return (float)(c(t) * NN) / (float)NO - (float)a(t); }
  static constexpr float den(int t) { return -(float)a(t); }
};

template <int S, int KT, bool WITH_J, int STOP, bool PP = false, int LPV = 16, int WIND = 6>
__global__ __launch_bounds__(256) void fista_fast_kernel(FistaArgs a, TapPairs<KT> taps) {
  static_assert(WIND >= 4 && WIND <= 8, "register-resident window rule: 4 <= wind <= 8");
  using WC = WinCoef<WIND>;
  constexpr int NR = WC::NR;                // ring slots: increments delta_k .. delta_{k-NR+1}
  static_assert(LPV == 16 || LPV == 64, "a problem occupies one DPP row or one wave");
  constexpr int H = KT - 1;                 // halo length
  constexpr int D = (H + S - 1) / S;        // neighbour lanes that contribute halo
  constexpr int NPAIR = (H + S + 2) / 2;    // window pairs (>= H+S+1 elements)
  static_assert(D <= 15, "halo spans more than one DPP row");
  using TP = TapPairs<KT>;

  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int sub = threadIdx.x & (LPV - 1);  // lane within the problem's segment
  int s0, s1;
  launch_slots(a, s0, s1);
  // the whole wave lies beyond this launch's slots (a candidate launch of a device-side plan that got nothing): leave
  if ((int)((gid - (threadIdx.x & 63)) / LPV) + s0 >= s1) return;
  bool live;
  const int p = slot_to_problem(a, (int)(gid / LPV) + s0, s1, live);
  const int base = sub * S;
  // re-solve pass behind the certificate / matrix-pipe kernels: only flagged problems; a wave
  // none of whose rows is flagged leaves at once (no barrier anywhere in this kernel)
  if (a.only_flagged) {
    live = live && a.n_done[p] < 0;
    if (__builtin_amdgcn_ballot_w64(live) == 0) return;
  }

  // ---- load the problem: y strip (fp32), w strip (fp64) -------------------
  float y[S];
  double w[S];
  {
    const float* yrow = a.y + (int64_t)(p / a.y_rep) * a.ldy;
    const double* wrow = a.w + (int64_t)p * a.ldw;
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const bool ok = live && (base + j < a.N);
      y[j] = ok ? yrow[base + j] : 0.0f;
      w[j] = (ok && !a.cold) ? wrow[base + j] : 0.0;
    }
  }
  // 1.0 for the real samples of this lane, 0.0 for the padding behind sample N-1
  // (kept as floats in VGPRs: hoisted lane masks would spill the SGPR file)
  float mk[S];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    mk[j] = (base + j < a.N) ? 1.0f : 0.0f;
    asm volatile("" : "+v"(mk[j]));
  }
  const double lb = a.lbda_vec ? a.lbda_vec[p] : a.lbda;
  const double stp = PP ? a.step_vec[a.step_shared ? 0 : p] : a.step;
  const double th = lb * stp;
  const double nstep = -stp;
  const float lbf = (float)lb;

  // The one-tap-shifted copy of the taps lives in VGPRs (two SGPR copies would
  // spill); the empty asm makes the values opaque so they are not rematerialised.
  f2 odd_v[TP::NO];
  f2 even_v[PP ? TP::NE : 1];
  if constexpr (PP) {
    const double* tp = a.taps_pp + (int64_t)p * a.ldt;
    auto h = [&](int m) -> float { return (m >= 0 && m < a.K) ? (float)tp[m] : 0.0f; };
#pragma unroll
    for (int t = 0; t < TP::NE; ++t) even_v[t] = f2{h(2 * t), h(2 * t + 1)};
#pragma unroll
    for (int t = 0; t < TP::NO; ++t) odd_v[t] = f2{h(2 * t - 1), h(2 * t)};
  } else {
#pragma unroll
    for (int t = 0; t < TP::NO; ++t) {
      odd_v[t] = taps.odd[t];
      asm volatile("" : "+v"(odd_v[t]));
    }
  }
  auto even_tap = [&](auto tc) -> f2 {
    constexpr int t = decltype(tc)::value;
    if constexpr (PP) return even_v[t]; else return taps.even[t];
  };

  // window rule: the criterion is a function of the current u_k, w_{k+1} and of the last
  // wind - 2 INCREMENTS delta_i = u_i - u_{i-1} only (WinCoef above).  u_{k-1} stays in float64
  // registers; the increments, small numbers, are stored as float32 in an LDS ring
  // [slot][sample][lane] (relative accuracy 6e-8 of the increments, i.e. of the criterion
  // itself).  Each lane reads back only what it wrote: no barrier.
  double uprev[STOP == 2 ? S : 1];
  float* ring = nullptr;
  int rp = 0;                               // ring slot that receives delta_k
  if constexpr (STOP == 2) {
    extern __shared__ __attribute__((aligned(16))) char fast_smem[];
    ring = reinterpret_cast<float*>(fast_smem) + ((threadIdx.x / LPV) * NR * S * LPV + sub);
#pragma unroll
    for (int j = 0; j < S; ++j) uprev[j] = 0.0;
    // the older slots are read before the first iterations have written them; each lane owns
    // its slots, so it zeroes them itself and no barrier is needed
#pragma unroll
    for (int q = 0; q < NR * S; ++q) ring[q * LPV] = 0.0f;
  }

  bool active = live;                       // per-problem (row-uniform) early-stop state; idle rows never hold a wave
  int done = 0;
  float* Jrow = WITH_J ? a.J + (int64_t)p * a.ldj : nullptr;

  // n_stop = iterations to execute; with WITH_J one more forward pass follows the
  // last iteration to price its iterate.  Every wave reaches it == n_stop.
  int n_stop = a.n_iter;
  for (int it = 0;; ++it) {
    if (!WITH_J && it >= n_stop) break;
    // ---- z = cumsum(w) ----------------------------------------------------
    float z[S];
    z[0] = (float)w[0];
#pragma unroll
    for (int j = 1; j < S; ++j) z[j] = z[j - 1] + (float)w[j];
    {
      const float off = seg_sum_below<LPV>(z[S - 1]);
#pragma unroll
      for (int j = 0; j < S; ++j) z[j] += off;
    }

    // ---- window of z: own samples at [H, H+S), halo below ------------------
    Window<NPAIR> Z;
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) Z.p[i] = f2{0.f, 0.f};
    static_for<0, S>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      Z.template set<H + j>(z[j]);
    });
    static_for<1, D + 1>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      static_for<0, S>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int e = H - d * S + j;
        if constexpr (e >= 0) Z.template set<e>(seg_from_below<LPV, d>(z[j]));
      });
    });

    // ---- r = h * z - y  (packed over taps) --------------------------------
    float r[S];
    static_for<0, S>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int q = H + j;
      f2 acc = f2{-y[j], 0.f};
      if constexpr (q % 2 == 1) {
        static_for<0, TP::NE>([&](auto ac) {
          constexpr int t = decltype(ac)::value;
          constexpr int pi = (q - 2 * t - 1) / 2;
          acc = __builtin_elementwise_fma(even_tap(ac), Z.p[pi].yx, acc);
        });
      } else {
        static_for<0, TP::NO>([&](auto ac) {
          constexpr int t = decltype(ac)::value;
          constexpr int pi = (q - 2 * t) / 2;
          acc = __builtin_elementwise_fma(odd_v[t], Z.p[pi].yx, acc);
        });
      }
      r[j] = (acc.x + acc.y) * mk[j];
    });

    // ---- cost of the iterate this pass started from -----------------------
    if constexpr (WITH_J) {
      if (it > 0) {
        float sq = 0.f, l1 = 0.f;
#pragma unroll
        for (int j = 0; j < S; ++j) {
          sq = fmaf(r[j], r[j], sq);
          l1 += fabsf((float)w[j]);
        }
        const float cost = seg_allsum<LPV>(fmaf(0.5f, sq, lbf * l1));
        if (live && sub == 0 && (STOP == 0 || it <= done)) Jrow[it - 1] = cost;
      }
      if (it >= n_stop) break;
    }

    // ---- window of r: own samples at [0, S), halo above --------------------
    Window<NPAIR> R;
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) R.p[i] = f2{0.f, 0.f};
    static_for<0, S>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      R.template set<j>(r[j]);
    });
    static_for<1, D + 1>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      static_for<0, S>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int e = d * S + j;
        if constexpr (e < S + H) R.template set<e>(seg_from_above<LPV, d>(r[j]));
      });
    });

    // ---- c = K^T r, g = reverse-cumsum(c) ---------------------------------
    float g[S];
    static_for<0, S>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      f2 acc = f2{0.f, 0.f};
      if constexpr (j % 2 == 0) {
        static_for<0, TP::NE>([&](auto ac) {
          constexpr int t = decltype(ac)::value;
          acc = __builtin_elementwise_fma(even_tap(ac), R.p[(j + 2 * t) / 2], acc);
        });
      } else {
        static_for<0, TP::NO>([&](auto ac) {
          constexpr int t = decltype(ac)::value;
          acc = __builtin_elementwise_fma(odd_v[t], R.p[(j + 2 * t - 1) / 2], acc);
        });
      }
      g[j] = acc.x + acc.y;
    });
#pragma unroll
    for (int j = S - 2; j >= 0; --j) g[j] += g[j + 1];
    {
      const float off = seg_sum_above<LPV>(g[0]);
#pragma unroll
      for (int j = 0; j < S; ++j) g[j] += off;
    }

    // ---- gradient step, prox, momentum (float64) ---------------------------
    //   u = w - s g ; d = clamp(u, -th, th) ; p = u - d ; w' = p + beta (p - u)
    //   = u - (1 + beta) d
    const double beta = a.betas[it];
    const double nb1 = -(1.0 + beta);
    if constexpr (STOP == 0) {
#pragma unroll
      for (int j = 0; j < S; ++j) {
        const double u = fma(nstep, (double)g[j], w[j]);
        const double d = fmin(fmax(u, -th), th);
        w[j] = fma(nb1, d, u);
      }
    } else {
      // One pass per sample: gradient step, prox+momentum, stop-rule partial sums.  The
      // iterate is updated unconditionally: a problem (row) that meets its rule is written
      // out at that moment (rare, divergent store) and its lanes simply keep iterating,
      // results discarded -- cheaper than a per-sample select in every iteration.
      double num = 0.0, den = 0.0;
      double floor_eps = 1.0e-10;
      if constexpr (STOP == 1) {
        // _loops_deconv rule (pybold/bold_signal.py:267-273):
        //   ||w' - u|| / (||w'|| + 1e-10) < tol, tested from the 4th iteration on
#pragma unroll
        for (int j = 0; j < S; ++j) {
          const double u = fma(nstep, (double)g[j], w[j]);
          const double d = fmin(fmax(u, -th), th);
          const double wn = fma(nb1, d, u);
          const double diff = wn - u;
          num = fma(diff, diff, num);
          den = fma(wn, wn, den);
          w[j] = wn;
        }
        num = seg_allsum_f64<LPV>(num);
        den = seg_allsum_f64<LPV>(den);
      } else {
        // deconv window rule (pybold/bold_signal.py:82-95): the stored iterates are
        // [u_{k-W+2} .. u_k, w_{k+1}] (each stored w was overwritten in place by the next gradient
        // step, :65/:72); old = mean of the first W - W/2, new = mean of the last W/2; weights:
        // WinCoef (sums left unscaled by NN, so is the 1e-10 floor).  delta_k and e are
        // differences of float64 values rounded to float32 (relative accuracy 6e-8 each); the two
        // norms are then accumulated as one packed float32 pair.  The criterion is accurate to
        // ~1e-6 relative, well inside what the float32 FIRs of the iterate itself leave of it.
        floor_eps = WC::NN * 1.0e-10;
        const float* rt[NR];                                   // rt[t] -> delta_{k-t}, t >= 1
#pragma unroll
        for (int t = 1; t < NR; ++t) {
          int q = rp - t;
          q = q < 0 ? q + NR : q;
          rt[t] = ring + q * S * LPV;
        }
        float* r0 = ring + rp * S * LPV;                       // delta_k goes here
        rp = (rp + 1 == NR) ? 0 : rp + 1;
        f2 nd = f2{0.f, 0.f};
#pragma unroll
        for (int j = 0; j < S; ++j) {
          const double u = fma(nstep, (double)g[j], w[j]);
          const double d = fmin(fmax(u, -th), th);
          const double wn = fma(nb1, d, u);
          const float dk = (float)(u - uprev[j]);
          const float ef = (float)(wn - u);
          const float uf = (float)u;
          float dsum = WC::num(NR - 1) * rt[NR - 1][j * LPV];  // oldest first (num = 1: exact)
          float dden = 0.0f;
          static_for<1, NR - 1>([&](auto tc) {
            constexpr int t = NR - 1 - decltype(tc)::value;    // NR-2 .. 1
            const float dt = rt[t][j * LPV];
            dsum = fmaf(WC::num(t), dt, dsum);
            if constexpr (WC::den(t) != 0.0f) dden = fmaf(WC::den(t), dt, dden);
          });
          dsum = fmaf(WC::num(0), dk, dsum);
          float tail = ef;
          if constexpr (WC::den(0) != 0.0f) tail = fmaf(WC::den(0), dk, ef);
          if constexpr (WC::NN > 3) tail += dden;
          const f2 v = f2{dsum + ef, fmaf((float)WC::NN, uf, tail)};
          nd = __builtin_elementwise_fma(v, v, nd);
          r0[j * LPV] = dk;
          uprev[j] = u;
          w[j] = wn;
        }
        num = (double)seg_allsum<LPV>(nd.x);
        den = (double)seg_allsum<LPV>(nd.y);
      }
      if (active) {
        done = it + 1;
        constexpr int first_test = (STOP == 1) ? 3 : WIND + 1;   // `idx > wind` (:85)
        static_assert(STOP != 2 || first_test >= NR - 1, "every ring slot is valid at the first test");
        if (it >= first_test && sqrt(num) / (sqrt(den) + floor_eps) < a.tol) {
          active = false;
          if (live) {                           // this problem is finished: write it out now
            double* wrow = a.w + (int64_t)p * a.ldw;
#pragma unroll
            for (int j = 0; j < S; ++j)
              if (base + j < a.N) wrow[base + j] = w[j];
          }
        }
      }
      if (__builtin_amdgcn_ballot_w64(active) == 0) n_stop = it + 1;
    }
  }

  // ---- store the final iterate (problems stopped by their rule were stored then) -----
  if (live) {
    if (STOP == 0 || active) {
      double* wrow = a.w + (int64_t)p * a.ldw;
#pragma unroll
      for (int j = 0; j < S; ++j)
        if (base + j < a.N) wrow[base + j] = w[j];
    }
    if (a.n_done && sub == 0) a.n_done[p] = (STOP == 0) ? a.n_iter : done;
  }
}

}  // namespace pb
