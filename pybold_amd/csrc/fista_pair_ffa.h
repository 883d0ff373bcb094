// Register-resident fused FISTA kernel, "pair" form with 2-parallel fast FIRs.
//
// Same mapping, recurrence and numerics as fista_pair.h (two problems per 16-lane DPP row,
// every per-sample quantity a float2 over the two problems, float64 iterate and update);
// what changes is the arithmetic of the two K-tap FIRs, 70 % of that kernel's instructions.
// Split window and taps into even and odd phases (pair index n, window position q = 2n / 2n+1):
//
//   forward   x(2n)   = A[n] + B[n-1]            A = h0 * Ze,  B = h1 * Zo,
//             x(2n+1) = C[n] - A[n] - B[n]       C = (h0 + h1) * (Ze + Zo)
//   adjoint   c(2n)   = A'[n] + C'[n]            A' = h0 . (Re - Ro),  B' = h1 . (Re[+1] - Ro),
//             c(2n+1) = B'[n] + C'[n]            C' = (h0 + h1) . Ro
//
// (h0[k] = h[2k], h1[k] = h[2k+1]; "*" causal convolution, "." correlation): three
// half-length filters on half-rate sequences instead of four, i.e. 3/4 of the multiply-adds
// for a few packed adds before and after.  At S = 19, K = 30 with the zero leading tap of an
// SPM HRF skipped: 440 + 53 packed instructions per forward FIR instead of 551, 425 + 67 per
// adjoint FIR.
//
// Registers: the summed windows are produced just before their first use and die as the
// outputs advance, so that both float64 iterates still fit (240 VGPRs, two waves per SIMD);
// the padding mask is no longer read from LDS but computed on the fly, on a scalar branch,
// for the few trailing samples of a strip that can be padded at all; y is stored
// pre-combined as the initial values of the accumulator chains.  (WB_LDS = true parks the
// iterate of problem B in LDS between its two uses of an iteration instead: 206 VGPRs, but
// 57 more LDS operations per iteration and 3 % slower; kept for larger S.)
#pragma once
#include "common.h"
#include "fista_fast.h"
#include "fista_pair.h"

namespace pb {

constexpr int PAIR_LJ = 12;          // floats of per-row scratch behind the staging regions
constexpr int PAIR_LC = 16;          // CERT: floats of per-lane scratch behind that

// taps as (h0[k], h1[k]) pairs and the sums hs[k] = h0[k] + h1[k] packed two per SGPR pair
template <int KT>
struct TapsFFA {
  static constexpr int KE = (KT + 1) / 2;
  f2 pr[KE];
  f2 sm[(KE + 1) / 2];
};

template <int KT>
inline TapsFFA<KT> make_taps_ffa(const double* taps, int K) {
  TapsFFA<KT> t;
  auto h = [&](int m) -> float { return (m < K) ? (float)taps[m] : 0.0f; };
  constexpr int KE = TapsFFA<KT>::KE;
  for (int k = 0; k < KE; ++k) t.pr[k] = f2{h(2 * k), h(2 * k + 1)};
  for (int i = 0; i < (KE + 1) / 2; ++i) {
    auto hs = [&](int k) -> float { return k < KE ? h(2 * k) + h(2 * k + 1) : 0.0f; };
    t.sm[i] = f2{hs(2 * i), hs(2 * i + 1)};
  }
  return t;
}

// TAPS_DEV: the HRF and the step are read from device memory (a.taps_pp: K float64 shared by
// every problem, a.step_vec[0]) instead of the kernel arguments -- the shared-HRF blind step,
// whose taps come out of pb_theta_fit without passing through the host.  Uniform loads: the
// taps still end up in SGPRs.
//
// CERT: the deconv window rule (wind = 6, pybold/bold_signal.py:82-95) as a per-iteration
// NO-FIRE CERTIFICATE.  The rule stops when  ||3(new - old)|| / (||3 new|| + 3e-10) < tol  with
//   3 (new - old) = delta_{k-3} + 2 delta_{k-2} + 3 delta_{k-1} + 2 delta_k + e,   3 new = u_{k-1} + u_k + w_{k+1}
// (delta_i = u_i - u_{i-1}, e = w_{k+1} - u_k; fista_fast.h).  Its full evaluation needs the last
// four increments of EVERY sample (no room for them at this density); a proof that it does NOT
// fire needs much less:
//   numerator   >= the same combination on ONE tracked sample per lane (16 of the 16 S samples:
//                  its increment history lives in 5 registers per problem), less its rounding;
//   denominator <= ||w_k|| + 2 ||w_{k+1}|| + 4 th sqrt(N)      since u_i = w_{i+1} + (1+beta_i) d_i
//                  with |d_i| <= th elementwise; the two norms come out of the cumsum pass.
// Squared and split with Cauchy-Schwarz weights p = (0.3133, 0.6467, 0.04) (2 % loose when the two
// norms agree), the test becomes ONE row sum per problem and iteration of lane partials:
//   sum_lanes [ v_lane^2 - tol^2 (||w_k||^2_lane / p1 + 4 ||w_{k+1}||^2_lane / p2) ]  >=  tol^2 c0^2 / p3.
// A problem whose certificate fails at some iteration (the rule may or may not have fired) is
// FLAGGED: its iterate is not stored and n_done[p] = -1; the caller re-solves the flagged
// problems exactly on fista_fast_kernel<..., STOP = 2> (capi.hip).  On the reference's default
// tol = 1e-6 the criterion stays ~0.9/k and the bound is ~4x below it: nothing is flagged.
// Implies the rotated loop of WITH_J; the cost trace itself is written only if a.J != nullptr.
//
// SPLIT: ONE series of 16 S < N <= 32 S scans per row, its first 16 S samples in slot A and the rest
// in slot B of every float2 -- the pair arithmetic unchanged, at the same density per sample
// (the reference's shipped demo is 600 scans, examples/synth_data/deconv.py:46).  What changes is
// where the two halves meet: slot B's halo below comes from slot A's last lanes (a row rotation
// of A merged into the zero-filled shift of B: DPP with bound_ctrl off keeps `old` in the lanes
// whose source is outside the row), slot A's halo above from slot B's first lanes, the prefix
// scan of B starts from A's total and the suffix scan of A from B's.
template <int S, int KT, bool WITH_J = false, bool SKIP0 = false, bool WB_LDS = false, bool TAPS_DEV = false,
          bool CERT = false, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void fista_pair_ffa_kernel(FistaArgs a, TapsFFA<KT> taps_arg) {
  static_assert(!CERT || WITH_J, "the certificate runs in the rotated (cost trace) loop");
  static_assert(!SPLIT || !WB_LDS, "SPLIT keeps both halves in registers");
  constexpr int H = KT - 1;
  constexpr int D = (H + S - 1) / S;
  constexpr int KE = (KT + 1) / 2;          // taps per phase
  constexpr int WL = H + S;                 // window length
  constexpr int NW = (WL + 1) / 2;          // window pairs
#ifndef PB_FFA_GN
#define PB_FFA_GN 2
#endif
  constexpr int GN = PB_FFA_GN;             // pairs per accumulator group (3 chains each)
  static_assert(D <= 15, "halo spans more than one DPP row");

  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int sub = threadIdx.x & 15;
  const int row = gid >> 4;
  const int base = sub * S;
  int s0, n_list;                                 // this launch's slots [s0, n_list) of its list (the batch itself without a partition)
  launch_slots(a, s0, n_list);
  const int pA0 = SPLIT ? row + s0 : 2 * row + s0, pB0 = SPLIT ? pA0 : pA0 + 1;
  // the four rows of this wave all lie beyond the list: leave (wave-level synchronisation only below)
  if ((SPLIT ? 4 : 8) * (gid >> 6) + s0 >= n_list) return;
  bool liveA, liveB;
  // samples of the series held by slot A / slot B, and where slot B starts in the row
  const int nA = SPLIT ? 16 * S : a.N, nB = SPLIT ? a.N - 16 * S : a.N, oB = SPLIT ? 16 * S : 0;
  const int pA = slot_to_problem(a, pA0, n_list, liveA);
  const int pB = slot_to_problem(a, pB0, n_list, liveB);

  extern __shared__ __attribute__((aligned(16))) char pair_smem[];
  const int rslot = (threadIdx.x >> 4) * S * 16;
  f2* ly = reinterpret_cast<f2*>(pair_smem) + (rslot + sub);                       // yc[j] at ly[j*16]
  double* lw = reinterpret_cast<double*>(pair_smem + (size_t)16 * S * 16 * sizeof(f2)) + (rslot + sub);
  double* stage_d = reinterpret_cast<double*>(reinterpret_cast<f2*>(pair_smem) + rslot);
  // second region: float [S][16] staging rows (prologue) -- float64 when it also parks wB
  constexpr size_t R2 = WB_LDS ? sizeof(double) : sizeof(float);
  float* stage_f = reinterpret_cast<float*>(pair_smem + (size_t)16 * S * 16 * sizeof(f2)) +
                   (WB_LDS ? 2 : 1) * rslot;
  // per row: [0,1] lambda, [2,3] ||w||_1, CERT: [10,11] tol^2 c0^2 / p3
  float* lj = reinterpret_cast<float*>(pair_smem + (size_t)16 * S * 16 * (sizeof(f2) + R2)) +
              (threadIdx.x >> 4) * PAIR_LJ;
  auto lds_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  double wA[S];
  double wBr[WB_LDS ? 1 : S];               // problem B's iterate: registers, or parked in LDS
  {
    const float* yA = a.y + (int64_t)(pA / a.y_rep) * a.ldy;
    const float* yB = a.y + (int64_t)(pB / a.y_rep) * a.ldy + oB;
    const double* rA = a.w + (int64_t)pA * a.ldw;
    const double* rB = a.w + (int64_t)pB * a.ldw + oB;
    // coalesced read -> LDS (natural order) -> strips (see fista_pair.h)
    auto load_w = [&](const double* row, double* strip, int nv) {
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const int i = k * 16 + sub;
        stage_d[i] = (i < nv && !a.cold) ? row[i] : 0.0;
      }
      lds_sync();
#pragma unroll
      for (int j = 0; j < S; ++j) strip[j] = stage_d[base + j];
      lds_sync();
    };
    double wB[S];
    load_w(rA, wA, nA);
    load_w(rB, wB, nB);
    float ya[S], yb[S];
    auto load_y = [&](const float* row, float* strip, int nv) {
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const int i = k * 16 + sub;
        stage_f[i] = (i < nv) ? row[i] : 0.0f;
      }
      lds_sync();
#pragma unroll
      for (int j = 0; j < S; ++j) strip[j] = stage_f[base + j];
      lds_sync();
    };
    load_y(yA, ya, nA);
    load_y(yB, yb, nB);
    // y as the initial values of the forward accumulator chains: the output at an even
    // window position starts its A chain from -y_j, the one at an odd position its C chain
    // from -(y_j + y_{j-1}) (y_{j-1} only if that even position is an output of this lane)
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const bool odd_q = ((H + j) & 1) != 0;
      const bool with_prev = odd_q && j >= 1;
      ly[j * 16] = f2{ya[j] + (with_prev ? ya[j - 1] : 0.0f), yb[j] + (with_prev ? yb[j - 1] : 0.0f)};
      if constexpr (WB_LDS) lw[j * 16] = wB[j]; else wBr[j] = wB[j];
    }
  }
  TapsFFA<KT> taps = taps_arg;
  double step = a.step;
  if constexpr (TAPS_DEV) {
    auto uni = [](float v) -> float {       // wave-uniform value -> SGPR
      return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
    };
    auto h = [&](int m) -> float { return (m < a.K) ? (float)a.taps_pp[m] : 0.0f; };
#pragma unroll
    for (int k = 0; k < KE; ++k) taps.pr[k] = f2{uni(h(2 * k)), uni(h(2 * k + 1))};
#pragma unroll
    for (int i = 0; i < (KE + 1) / 2; ++i) {
      const float s0 = (2 * i < KE) ? h(4 * i) + h(4 * i + 1) : 0.0f;
      const float s1 = (2 * i + 1 < KE) ? h(4 * i + 2) + h(4 * i + 3) : 0.0f;
      taps.sm[i] = f2{uni(s0), uni(s1)};
    }
    step = a.step_vec[0];
  }
  const double lbA = a.lbda_vec ? a.lbda_vec[pA] : a.lbda;
  const double lbB = a.lbda_vec ? a.lbda_vec[pB] : a.lbda;
  const double thA = lbA * step, thB = lbB * step;
  const double nstep = -step;
  if constexpr (WITH_J) {
    lj[0] = (float)lbA;
    lj[1] = (float)lbB;
  }
  // CERT: c0 = 4 th sqrt(16 S) >= 4 th sqrt(N), rounded up, + the rule's 3e-10 floor
  constexpr float CP1 = 0.3133f, CP2 = 0.6467f, CP3 = 0.04f;
  const float cert_t2 = ((float)a.tol * 1.001f) * ((float)a.tol * 1.001f);
  if constexpr (CERT) {
    const float cert_sq = 16.0f * 1.0001f * __builtin_sqrtf((float)(SPLIT ? 2 * S : S));
    const float c0A = (float)thA * cert_sq + 3.1e-10f, c0B = (float)thB * cert_sq + 3.1e-10f;
    lj[10] = cert_t2 * c0A * c0A * (1.0001f / CP3);
    lj[11] = cert_t2 * c0B * c0B * (1.0001f / CP3);
  }
  int cert_it = -1;                         // iteration whose certificate the next forward pass closes
  // samples that can be padding in SOME lane: j >= jpad (uniform); sample j of this lane is
  // real iff j < jlim = N - base, i.e. mask = saturate(jlim - j) as a float
  const int jpad = (nB - 15 * S > 0) ? nB - 15 * S : 0;
  const float jlimf = (float)(nB - base);
  // CERT state, per lane, in LDS behind the per-row scratch (slot-major: conflict-free b32
  // accesses; each lane touches only its own words): [0..7] ring of the tracked sample's last
  // four increments (float32, slot = 2 (k mod 4) + problem), [8..11] its u_{k-1} (float64 halves),
  // [12,13] v^2 of this lane, [14,15] this lane's part of ||w_k||^2
  constexpr int JT = S / 2;
  float* lc = reinterpret_cast<float*>(pair_smem + (size_t)16 * S * 16 * (sizeof(f2) + R2)) + 16 * PAIR_LJ +
              threadIdx.x;
  if constexpr (CERT) {
#pragma unroll
    for (int q = 0; q < PAIR_LC; ++q) lc[q * 256] = 0.0f;
  }
  bool flagA = false, flagB = false;

  auto tap0 = [&](auto kc) -> f2 { return taps.pr[decltype(kc)::value].xx; };
  auto tap1 = [&](auto kc) -> f2 { return taps.pr[decltype(kc)::value].yy; };
  auto taps_sum = [&](auto kc) -> f2 {
    constexpr int k = decltype(kc)::value;
    const f2 p = taps.sm[k / 2];
    return (k % 2 == 0) ? p.xx : p.yy;
  };

  // ---- forward pass: r = h * cumsum(w) - y for both problems ------------------------
  auto forward = [&](f2 (&r)[S]) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");          // keep the LDS reads inside the loop
    f2 z[S];
    {
      double wb[S];
#pragma unroll
      for (int j = 0; j < S; ++j) {
        if constexpr (WB_LDS) wb[j] = lw[j * 16]; else wb[j] = wBr[j];
      }
      z[0] = f2{(float)wA[0], (float)wb[0]};
      if constexpr (WITH_J) {
        // CERT: the lane partials and the limit are read here, a cumsum away from their use
        f2 c_vsq, c_wp, c_lim;
        if constexpr (CERT) {
          c_vsq = f2{lc[12 * 256], lc[13 * 256]};
          c_wp = f2{lc[14 * 256], lc[15 * 256]};
          c_lim = f2{lj[10], lj[11]};
          asm volatile("" : "+v"(c_vsq), "+v"(c_wp), "+v"(c_lim));     // issued here, not sunk to the use
        }
        float l1a = fabsf(z[0].x), l1b = fabsf(z[0].y);
        f2 sq = z[0] * z[0];
#pragma unroll
        for (int j = 1; j < S; ++j) {
          const f2 wj = f2{(float)wA[j], (float)wb[j]};
          z[j] = z[j - 1] + wj;
          l1a += fabsf(wj.x);
          l1b += fabsf(wj.y);
          if constexpr (CERT) sq = __builtin_elementwise_fma(wj, wj, sq);
        }
        lj[2] = row_allsum(l1a);
        lj[3] = row_allsum(l1b);
        if constexpr (CERT) {
          // close the certificate of iteration cert_it (the rule is first tested at wind + 1 = 7)
          const f2 t = c_vsq - cert_t2 * ((1.0001f / CP1) * c_wp + (4.0001f / CP2) * sq);
          const float tA = row_allsum(t.x), tB = row_allsum(t.y);
          const bool chk = cert_it >= 7;
          if constexpr (SPLIT) {
            flagA = flagA | (chk & !(tA + tB >= c_lim.x));
            flagB = flagA;
          } else {
            flagA = flagA | (chk & !(tA >= c_lim.x));   // NaN-safe: anything unclear is flagged
            flagB = flagB | (chk & !(tB >= c_lim.y));
          }
          lc[14 * 256] = sq.x;
          lc[15 * 256] = sq.y;
        }
      } else {
#pragma unroll
        for (int j = 1; j < S; ++j) z[j] = z[j - 1] + f2{(float)wA[j], (float)wb[j]};
      }
    }
    {
      f2 off = f2{row_from_below<1>(row_prefix_incl(z[S - 1].x)),
                  row_from_below<1>(row_prefix_incl(z[S - 1].y))};
      if constexpr (SPLIT) off.y += row_allsum(z[S - 1].x);      // slot B continues slot A's sum
#pragma unroll
      for (int j = 0; j < S; ++j) z[j] += off;
    }
    // window of z in phases: Z[2n] = Ze[n], Z[2n+1] = Zo[n]; halo [0, H) from the lanes below
    f2 Z[2 * NW];
    if constexpr (WL % 2 == 1) Z[WL] = f2{0.f, 0.f};
    static_for<0, S>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      Z[H + j] = z[j];
    });
    static_for<1, D + 1>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      static_for<0, S>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int e = H - d * S + j;
        if constexpr (e >= 0) {
          if constexpr (SPLIT) Z[e] = f2{dpp_zero<DPP_ROW_SHR + d>(z[j].x),
                                        dpp_keep<DPP_ROW_SHR + d>(dpp_zero<DPP_ROW_ROR + d>(z[j].x), z[j].y)};
          else Z[e] = dpp_zero2<DPP_ROW_SHR + d>(z[j]);
        }
      });
    });

    constexpr int q_lo = H, q_hi = H + S - 1;
    constexpr int n_lo = q_lo / 2, n_hi = q_hi / 2;
    auto has_even = [](int n) constexpr { return 2 * n >= q_lo && 2 * n <= q_hi; };
    auto has_odd = [](int n) constexpr { return 2 * n + 1 >= q_lo && 2 * n + 1 <= q_hi; };
    f2 ZS[NW];                                // Ze + Zo, filled just before its first use
    static_for<(n_lo - KE + 1 > 0 ? n_lo - KE + 1 : 0), n_lo>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      ZS[i] = Z[2 * i] + Z[2 * i + 1];
    });
    f2 b_prev = f2{0.f, 0.f};
    if constexpr (has_even(n_lo)) {           // B[n_lo - 1] for the first even output
      static_for<0, KE>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (2 * k + 1 < KT && n_lo - 1 - k >= 0)
          b_prev = __builtin_elementwise_fma(tap1(kc), Z[2 * (n_lo - 1 - k) + 1], b_prev);
      });
    }
    static_for<0, (n_hi - n_lo + GN) / GN>([&](auto gc) {
      constexpr int n0 = n_lo + decltype(gc)::value * GN;
      constexpr int gn = (n_hi - n0 + 1 < GN) ? n_hi - n0 + 1 : GN;
      f2 ca[gn], cb[gn], cc[gn];
      static_for<0, gn>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int n = n0 + i;
        ZS[n] = Z[2 * n] + Z[2 * n + 1];
        // chain inits: -y of the even output on A, -(y_odd + y_even) on C
        if constexpr (has_even(n)) ca[i] = -ly[(2 * n - H) * 16]; else ca[i] = f2{0.f, 0.f};
        if constexpr (has_odd(n)) cc[i] = -ly[(2 * n + 1 - H) * 16]; else cc[i] = f2{0.f, 0.f};
        cb[i] = f2{0.f, 0.f};
      });
      static_for<0, KE>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        static_for<0, gn>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int n = n0 + i;
          if constexpr (!(SKIP0 && k == 0))
            ca[i] = __builtin_elementwise_fma(tap0(kc), Z[2 * (n - k)], ca[i]);
          if constexpr (2 * k + 1 < KT && (has_odd(n) || has_even(n + 1)))
            cb[i] = __builtin_elementwise_fma(tap1(kc), Z[2 * (n - k) + 1], cb[i]);
          if constexpr (has_odd(n))
            cc[i] = __builtin_elementwise_fma(taps_sum(kc), ZS[n - k], cc[i]);
        });
      });
      static_for<0, gn>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int n = n0 + i;
        const f2 bp = (i == 0) ? b_prev : cb[i > 0 ? i - 1 : 0];
        if constexpr (has_even(n)) r[2 * n - H] = ca[i] + bp;
        if constexpr (has_odd(n)) r[2 * n + 1 - H] = cc[i] - (ca[i] + cb[i]);
      });
      b_prev = cb[gn - 1];
    });
    // zero the residual behind sample N-1: only trailing samples of a strip can be padding,
    // the others skip the multiply on a scalar branch (the empty asm keeps it a branch and
    // the comparison inside the loop: hoisted, the 19 conditions would spill the SGPR file)
    int jp = jpad;
    float jl = jlimf;
    asm volatile("" : "+s"(jp), "+v"(jl));    // opaque per iteration: nothing below is hoisted
    // every residual exists before the first branch (else the accumulator chains of the later
    // samples are sunk between the branches, one lone dependent chain after the other)
#pragma unroll
    for (int j = 0; j < S; ++j) asm volatile("" : "+v"(r[j]));
    static_for<0, S>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (j >= jp) {
        asm volatile("");
        const float m = __builtin_amdgcn_fmed3f(jl - (float)j, 0.0f, 1.0f);
        r[j] = r[j] * f2{SPLIT ? 1.0f : m, m};
      }
    });
  };

  // ---- adjoint pass and update: w <- prox step from the residual r -------------------
  auto backward = [&](const f2 (&r)[S], const double beta) __attribute__((always_inline)) {
    f2 R[2 * NW];                             // own samples [0, S), halo [S, S+H) from above
    if constexpr (WL % 2 == 1) R[WL] = f2{0.f, 0.f};
    static_for<0, S>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      R[j] = r[j];
    });
    static_for<1, D + 1>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      static_for<0, S>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int e = d * S + j;
        if constexpr (e < S + H) {
          if constexpr (SPLIT) R[e] = f2{dpp_keep<DPP_ROW_SHL + d>(dpp_zero<DPP_ROW_ROR + (16 - d)>(r[j].y), r[j].x),
                                        dpp_zero<DPP_ROW_SHL + d>(r[j].y)};
          else R[e] = dpp_zero2<DPP_ROW_SHL + d>(r[j]);
        }
      });
    });
    // g(2n) = A'[n] + C'[n], g(2n+1) = B'[n] + C'[n]
    constexpr int n_hi = (S - 1) / 2;
    auto has_odd = [](int n) constexpr { return 2 * n + 1 <= S - 1; };
    f2 UA[NW], UB[NW];                        // Re - Ro, Re[+1] - Ro: filled just before first use
    static_for<0, KE - 1>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      UA[i] = R[2 * i] - R[2 * i + 1];
      if constexpr (2 * i + 2 < 2 * NW) UB[i] = R[2 * i + 2] - R[2 * i + 1];
    });
    f2 g[S];
    static_for<0, (n_hi + GN) / GN>([&](auto gc) {
      constexpr int n0 = decltype(gc)::value * GN;
      constexpr int gn = (n_hi - n0 + 1 < GN) ? n_hi - n0 + 1 : GN;
      f2 ca[gn], cb[gn], cc[gn];
      static_for<0, gn>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int e = n0 + i + KE - 1;    // newest window pair this group touches
        if constexpr (e < NW) {
          UA[e] = R[2 * e] - R[2 * e + 1];
          if constexpr (2 * e + 2 < 2 * NW) UB[e] = R[2 * e + 2] - R[2 * e + 1];
          else UB[e] = -R[2 * e + 1];
        }
        ca[i] = f2{0.f, 0.f};
        cb[i] = f2{0.f, 0.f};
        cc[i] = f2{0.f, 0.f};
      });
      static_for<0, KE>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        static_for<0, gn>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          constexpr int n = n0 + i;
          if constexpr (n + k < NW) {
            if constexpr (!(SKIP0 && k == 0))
              ca[i] = __builtin_elementwise_fma(tap0(kc), UA[n + k], ca[i]);
            if constexpr (2 * k + 1 < KT && has_odd(n))
              cb[i] = __builtin_elementwise_fma(tap1(kc), UB[n + k], cb[i]);
            cc[i] = __builtin_elementwise_fma(taps_sum(kc), R[2 * (n + k) + 1], cc[i]);
          }
        });
      });
      static_for<0, gn>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int n = n0 + i;
        g[2 * n] = ca[i] + cc[i];
        if constexpr (has_odd(n)) g[2 * n + 1] = cb[i] + cc[i];
      });
    });
#pragma unroll
    for (int j = S - 2; j >= 0; --j) g[j] += g[j + 1];
    {
      f2 off = f2{row_from_above<1>(row_suffix_incl(g[0].x)),
                  row_from_above<1>(row_suffix_incl(g[0].y))};
      if constexpr (SPLIT) off.x += row_allsum(g[0].y);          // slot A's suffix continues into slot B
#pragma unroll
      for (int j = 0; j < S; ++j) g[j] += off;
    }

    // ---- gradient step, prox, momentum (float64), per problem ------------------
    const double nb1 = -(1.0 + beta);
    {
      // CERT: the tracked sample's history is read here, the float64 update away from its use
      f2 c_d1, c_d2, c_d3;
      unsigned c_u[4];
      if constexpr (CERT) {
        const float* r1 = lc + ((cert_it + 3) & 3) * 512;
        const float* r2 = lc + ((cert_it + 2) & 3) * 512;
        const float* r3 = lc + ((cert_it + 1) & 3) * 512;
        c_d1 = f2{r1[0], r1[256]};
        c_d2 = f2{r2[0], r2[256]};
        c_d3 = f2{r3[0], r3[256]};
#pragma unroll
        for (int q = 0; q < 4; ++q) c_u[q] = __builtin_bit_cast(unsigned, lc[(8 + q) * 256]);
        asm volatile("" : "+v"(c_d1), "+v"(c_d2), "+v"(c_d3), "+v"(c_u[0]), "+v"(c_u[1]), "+v"(c_u[2]), "+v"(c_u[3]));
      }
      double uA[S], uB[S], dA[S], dB[S];
#pragma unroll
      for (int j = 0; j < S; ++j) {
        if constexpr (WB_LDS) uB[j] = lw[j * 16]; else uB[j] = wBr[j];
      }
#pragma unroll
      for (int j = 0; j < S; ++j) {
        uA[j] = fma(nstep, (double)g[j].x, wA[j]);
        uB[j] = fma(nstep, (double)g[j].y, uB[j]);
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
      double wnB_t = 0.0;
#pragma unroll
      for (int j = 0; j < S; ++j) {
        wA[j] = fma(nb1, dA[j], uA[j]);
        const double wnB = fma(nb1, dB[j], uB[j]);
        if constexpr (WB_LDS) lw[j * 16] = wnB; else wBr[j] = wnB;
        if (CERT && j == JT) wnB_t = wnB;
      }
      if constexpr (CERT) {
        // the window combination on the tracked sample of this lane, both problems (float32 from
        // float64 differences; its rounding, and that of the stored increments, is below 2^-21 M)
        auto st_d = [&](int q, double v) {
          const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
          lc[q * 256] = __builtin_bit_cast(float, (unsigned)b);
          lc[(q + 1) * 256] = __builtin_bit_cast(float, (unsigned)(b >> 32));
        };
        float* r0 = lc + (cert_it & 3) * 512;
        const f2 d1 = c_d1, d2 = c_d2, d3 = c_d3;
        const double upA = __builtin_bit_cast(double, ((unsigned long long)c_u[1] << 32) | c_u[0]);
        const double upB = __builtin_bit_cast(double, ((unsigned long long)c_u[3] << 32) | c_u[2]);
        const f2 dk = f2{(float)(uA[JT] - upA), (float)(uB[JT] - upB)};
        const f2 e = f2{(float)(wA[JT] - uA[JT]), (float)(wnB_t - uB[JT])};
        const f2 two = f2{2.f, 2.f}, three = f2{3.f, 3.f};
        f2 v = __builtin_elementwise_fma(two, dk, e);
        v = __builtin_elementwise_fma(three, d1, v);
        v = __builtin_elementwise_fma(two, d2, v) + d3;
        f2 m = __builtin_elementwise_fma(two, __builtin_elementwise_abs(dk), __builtin_elementwise_abs(e));
        m = __builtin_elementwise_fma(three, __builtin_elementwise_abs(d1), m);
        m = __builtin_elementwise_fma(two, __builtin_elementwise_abs(d2), m) + __builtin_elementwise_abs(d3);
        f2 vs = __builtin_elementwise_fma(f2{-0x1p-21f, -0x1p-21f}, m, __builtin_elementwise_abs(v));
        vs = __builtin_elementwise_max(vs, f2{0.f, 0.f});
        vs = vs * vs;
        lc[12 * 256] = vs.x;
        lc[13 * 256] = vs.y;
        r0[0] = dk.x;
        r0[256] = dk.y;
        st_d(8, uA[JT]);
        st_d(10, uB[JT]);
      }
    }
  };

  if constexpr (!WITH_J) {
    for (int it = 0; it < a.n_iter; ++it) {
      const double beta = a.betas[it];
      f2 r[S];
      forward(r);
      backward(r, beta);
    }
  } else {
    f2 r[S];
    forward(r);
    for (int it = 0; it < a.n_iter; ++it) {
      const double beta = a.betas[it];
      cert_it = it;
      backward(r, beta);
      forward(r);
      f2 sq = r[0] * r[0];
#pragma unroll
      for (int j = 1; j < S; ++j) sq = __builtin_elementwise_fma(r[j], r[j], sq);
      const float cA = fmaf(0.5f, row_allsum(sq.x), lj[0] * lj[2]);
      const float cB = fmaf(0.5f, row_allsum(sq.y), lj[1] * lj[3]);
      // (a flagged problem stops tracing: it was flagged no later than the iteration its rule
      // fires at, so the re-solve rewrites everything written here and nothing beyond its stop)
      if (sub == 0 && (!CERT || a.J != nullptr)) {
        if constexpr (SPLIT) {
          if (liveA && !flagA) a.J[(int64_t)pA * a.ldj + it] = cA + cB;
        } else {
          if (liveA && !flagA) a.J[(int64_t)pA * a.ldj + it] = cA;
          if (liveB && !flagB) a.J[(int64_t)pB * a.ldj + it] = cB;
        }
      }
    }
  }

  // epilogue: strips -> LDS -> coalesced stores (problem B's strip is read back first)
  double wBf[S];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    if constexpr (WB_LDS) wBf[j] = lw[j * 16]; else wBf[j] = wBr[j];
  }
  auto store_w = [&](const double* strip, double* row, bool live, int nv) {
    lds_sync();
#pragma unroll
    for (int j = 0; j < S; ++j) stage_d[base + j] = strip[j];
    lds_sync();
    if (live) {
#pragma unroll
      for (int k = 0; k < S; ++k) {
        const int i = k * 16 + sub;
        if (i < nv) row[i] = stage_d[i];
      }
    }
  };
  store_w(wA, a.w + (int64_t)pA * a.ldw, liveA && !flagA, nA);
  store_w(wBf, a.w + (int64_t)pB * a.ldw + oB, liveB && !flagB, nB);
  if (a.n_done && sub == 0) {
    if (liveA) a.n_done[pA] = flagA ? -1 : a.n_iter;
    if (liveB && !SPLIT) a.n_done[pB] = flagB ? -1 : a.n_iter;
  }
}

// shared HRF and step in device memory (a.taps_pp, a.step_vec[0]); no cost trace
template <int S, int KT>
int launch_pair_ffa_dev(const FistaArgs& a, hipStream_t st) {
  const TapsFFA<KT> none{};
  const int64_t rows = (launch_count(a) + 1) / 2;
  const dim3 grid((unsigned)((rows * 16 + 255) / 256)), block(256);
  const size_t lds = (size_t)16 * S * 16 * (sizeof(f2) + sizeof(float)) + 16 * PAIR_LJ * sizeof(float);
  hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, false, false, false, true>), grid, block, lds, st, a, none);
  return 0;
}

// one series of 16 S < N <= 32 S scans per row (SPLIT): plain, cost trace, certificate
template <int S, int KT>
int launch_pair_ffa_split(const FistaArgs& a, const double* taps, int K, bool with_j, bool cert, hipStream_t st) {
  if (a.N <= 16 * S || a.N > 32 * S || (cert && !a.n_done)) return 1;
  const auto tf = make_taps_ffa<KT>(taps, K);
  const int64_t rows = launch_count(a);
  const dim3 grid((unsigned)((rows * 16 + 255) / 256)), block(256);
  const size_t lds = (size_t)16 * S * 16 * (sizeof(f2) + sizeof(float)) +
                     (16 * PAIR_LJ + (cert ? 256 * PAIR_LC : 0)) * sizeof(float);
  const bool skip0 = KT > 1 && tf.pr[0].x == 0.0f;
  if (cert) {
    if (skip0) hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, true, true, false, false, true, true>), grid, block, lds, st, a, tf);
    else hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, true, false, false, false, true, true>), grid, block, lds, st, a, tf);
  } else if (with_j) {
    if (skip0) hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, true, true, false, false, false, true>), grid, block, lds, st, a, tf);
    else hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, true, false, false, false, false, true>), grid, block, lds, st, a, tf);
  } else {
    if (skip0) hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, false, true, false, false, false, true>), grid, block, lds, st, a, tf);
    else hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, false, false, false, false, false, true>), grid, block, lds, st, a, tf);
  }
  return 0;
}

// window rule (wind = 6) as a no-fire certificate; flagged problems come back with n_done = -1
template <int S, int KT>
int launch_pair_ffa_cert(const FistaArgs& a, const double* taps, int K, hipStream_t st) {
  if (!a.n_done) return 1;
  const auto tf = make_taps_ffa<KT>(taps, K);
  const int64_t rows = (launch_count(a) + 1) / 2;
  const dim3 grid((unsigned)((rows * 16 + 255) / 256)), block(256);
  const size_t lds = (size_t)16 * S * 16 * (sizeof(f2) + sizeof(float)) + (16 * PAIR_LJ + 256 * PAIR_LC) * sizeof(float);
  const bool skip0 = KT > 1 && tf.pr[0].x == 0.0f;
  if (skip0) hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, true, true, false, false, true>), grid, block, lds, st, a, tf);
  else hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, true, false, false, false, true>), grid, block, lds, st, a, tf);
  return 0;
}

template <int S, int KT>
int launch_pair_ffa(const FistaArgs& a, const double* taps, int K, bool with_j, hipStream_t st) {
  const auto tf = make_taps_ffa<KT>(taps, K);
  const int64_t rows = (launch_count(a) + 1) / 2;
  const dim3 grid((unsigned)((rows * 16 + 255) / 256)), block(256);
  const size_t lds = (size_t)16 * S * 16 * (sizeof(f2) + sizeof(float)) + 16 * PAIR_LJ * sizeof(float);
  const bool skip0 = KT > 1 && tf.pr[0].x == 0.0f;      // leading tap exactly zero
  if (with_j && skip0) hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, true, true>), grid, block, lds, st, a, tf);
  else if (with_j) hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, true, false>), grid, block, lds, st, a, tf);
  else if (skip0) hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, false, true>), grid, block, lds, st, a, tf);
  else hipLaunchKernelGGL((fista_pair_ffa_kernel<S, KT, false, false>), grid, block, lds, st, a, tf);
  return 0;
}

}  // namespace pb
