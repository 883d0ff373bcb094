// The matrix-pipe form of the fused FISTA kernel with ONE series split over the FOUR waves of a workgroup -- one wave on
// each SIMD of a compute unit: series of 641 .. 1 280 scans (21 .. 40 blocks of 32 samples; HCP-length runs,
// examples/icassp_2019/validation.py:41-48), which fit neither one wave (fista_mfma.h) nor two (fista_mfma2.h).
//
// Wave k owns blocks k A .. k A + A - 1 of the same 16 problems, A = ceil(ceil(N / 32) / 4) for every wave: the passes
// of an iteration run in all four waves at once between workgroup barriers, so a pass lasts what the LONGEST share
// lasts and blocks of padding behind the series (all of them in the last wave) cost nothing.  The coupling is
// fista_mfma2.h's, between neighbours and summed over everything further away:
//   forward  (r = T_c w - y, ascending): wave k > 0 needs the float16 fragment of wave k-1's LAST block and
//            S * sum(w over every block before that one) -- one scalar per problem: wave j publishes the float64 sums of
//            its updated iterate without (E_j) and with (T_j) its last block, wave k adds T_0 .. T_{k-2} + E_{k-1};
//   adjoint  (g = T_c^T r, descending): wave k < 3 needs the residual fragment of wave k+1's FIRST block (in LDS anyway)
//            and S * sum(r over every block behind that one): sums without (RE_j) and with (RT_j) the first block.
// Everything is produced at the end of the pass before; two workgroup barriers per iteration.  The operator (near band
// + constant far field, two near tiles, K <= 33), the scaling, the float16 split, the guards, the cost trace and the
// no-fire certificate of the window rule are fista_mfma2.h's / fista_mfma.h's; capi.hip re-solves what is handed back.
//
// Reference: pybold/bold_signal.py:62-72, pybold/linear.py:73-113, pybold/convolution.py:105-132.
#pragma once
#include "fista_mfma2.h"

namespace pb {

constexpr int MFMA4_WAVES = 4;

// bytes of dynamic LDS of one workgroup (four waves of A blocks): residual fragments, last-block fragments, cumulative
// taps, the float64 sums of the iterate, residual sums, scale, guards, cost-trace parts, certificate state
constexpr size_t mfma4_lds_bytes(int A) {
  return ((size_t)MFMA4_WAVES * A * 2 * 64 + MFMA4_WAVES * 2 * 2 * 64) * sizeof(u4) +
         ((size_t)MFMA4_WAVES * 2 * 256 + MFMA4_WAVES * 2 * 64) * sizeof(double) +
         (size_t)MFMA4_WAVES * (96 + 2 * 256 + 64 + 2 * 64 + 3 * 64) * sizeof(float) + (size_t)7 * 256 * sizeof(float);
}

// One wave's share, KW = wave index.  HAS_L / HAS_R: there is a wave to the left / right; LASTW: the wave holding the end
// of the series (any of its blocks may be padding).  (Four bodies per kernel: with the index at run time the two middle
// waves could share one, at the price of a dozen address registers -- and of scratch, at ten blocks per wave.)
template <int NBW, int KW, bool TAPS_DEV, bool WITH_J = false, bool CERT = false, bool LOOPS = false, int NT = 2>
__device__ __forceinline__ void mfma4_role(const FistaArgs& a, const MfmaTaps& tp, char* smem) {
  constexpr int k = KW;
  constexpr bool HAS_L = KW > 0, HAS_R = KW < MFMA4_WAVES - 1, LASTW = KW == MFMA4_WAVES - 1;
  static_assert(!CERT || (WITH_J && !TAPS_DEV), "the certificate runs in the rotated (cost trace) loop");
  static_assert(!LOOPS || (!WITH_J && !TAPS_DEV && !CERT), "the _loops_deconv rule rides the plain variant");
  static_assert(NT == 2 || NT == 3, "two near tiles (K <= 33) or three (K <= 65)");
  static_assert(NBW > NT && NBW <= 10, "more blocks per wave than near tiles (a tile reaches the neighbour only), ten at most");
  constexpr int NW = MFMA4_WAVES;
  constexpr int LCW = 32 * NT, LCS = 96;             // cumulative taps kept: lags 0 .. 32 NT - 1 (LCS: a wave's slice of the area)
  constexpr int NX = NT - 1;                       // blocks of a neighbour the near tiles reach into
  constexpr int NBT = NW * NBW;
  const int qoff = k * NBW;                        // this wave's first block within the series
  // (the lane from the exec mask, not from threadIdx: nothing of the kernel's entry state stays live across the roles)
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int v = lane & 15, g = lane >> 4;
  int s0, s1;                                      // this launch's slots of its list (fista_fast.h: launch_slots)
  launch_slots(a, s0, s1);
  bool live;
  const int p = slot_to_problem(a, (int)blockIdx.x * 16 + v + s0, s1, live);
  const int tb = 8 * g;
  const int nrem = a.N - 32 * qoff - tb;           // sample j of block q of this lane exists iff 32 q + j < nrem

  // ---- LDS ----------------------------------------------------------------------------------------------------------
  u4* const lbase = reinterpret_cast<u4*>(smem);
  u4* const lrf = lbase + k * (NBW * 2 * 64) + lane;               // this wave's residual fragments
  u4* const lrf_next = lbase + (k + 1) * (NBW * 2 * 64) + lane;    // the right neighbour's (its block 0)
  u4* const xwb = lbase + NW * (NBW * 2 * 64) + lane;              // [NW][2][2][64]: fragments (hi, lo) of wave j's last block and (three near tiles) the one before
  // The sums cross the cuts as the four LANE PARTS of each problem (slot 4 v + g), added up by the wave that reads them: a
  // lane-crossing sum at the end of a pass is two dependent ds_bpermute round trips with nothing to overlap them; at the
  // start of the next pass the reads hide behind the float16 split of the first block (fista_mfma2.h).
  double* const xsp = reinterpret_cast<double*>(lbase + NW * (NBW * 2 * 64) + NW * 2 * 2 * 64);    // [NW][2][16][4]: E_j, T_j lane parts
  double* const xl = xsp + NW * 2 * 256 + lane;                    // [NW][2][64]: _loops_deconv rule, each wave's ||d||^2, ||w'||^2
  float* const fbase = reinterpret_cast<float*>(xsp + NW * 2 * 256 + NW * 2 * 64);
  float* const lc = fbase + k * LCS;                               // [NW][96] cumulative taps, one copy per wave
  float* const xrp = fbase + NW * LCS;                             // [NW][2][16][4]: RE_j, RT_j lane parts
  float* const xm = fbase + NW * LCS + NW * 512 + lane;            // [NW][64] max |y| of each share
  float* const xg = fbase + NW * LCS + NW * 576 + lane;            // [NW][2][64] guard, largest |w| of each share
  float* const xj = fbase + NW * LCS + NW * 704 + lane;            // [NW][3][64] cost-trace parts: ||r||^2, ||w||_1, certificate
  float* const lt = fbase + NW * LCS + NW * 896 + 64 * k + lane;   // [7][256] certificate state of every lane (fista_mfma.h)
  if constexpr (CERT) {
#pragma unroll
    for (int q = 0; q < 7; ++q) lt[q * 256] = 0.0f;
  }
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto wg_sync = [] {                                // all waves: everything written to LDS before is visible after
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };

  // ---- cumulative taps (as fista_mfma.h) ------------------------------------------------------------------------------
  double step = a.step, g_scale = tp.g_scale;
  float y_scale = tp.y_scale;
  if constexpr (TAPS_DEV) {
    double run = 0.0, run2 = 0.0;
    for (int kk = 0; kk <= lane && kk < a.K; ++kk) run += (double)(float)a.taps_pp[kk];
    float cm = fabsf((float)run);
    if constexpr (NT == 3) {                       // lags 64 .. 95, every lane for lag 64 + (lane & 31): the store below stays unmasked
      for (int kk = 0; kk <= 64 + (lane & 31) && kk < a.K; ++kk) run2 += (double)(float)a.taps_pp[kk];
      cm = fmaxf(cm, fabsf((float)run2));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) cm = fmaxf(cm, __shfl_xor(cm, o, 64));
    int e = 0;
    if (cm > 0.0f) (void)frexpf(cm, &e);
    const int sa = 3 - e;
    lc[lane] = (float)ldexp(run, sa);
    if constexpr (NT == 3) lc[64 + (lane & 31)] = (float)ldexp(run2, sa);
    g_scale = ldexp(1.0, -2 * sa);
    y_scale = ldexpf(1.0f, sa);
    step = a.step_vec[0];
  } else {
    lc[lane] = tp.c[lane];
    // (every lane, the upper half twice: a `lane < 32` mask is one more exec-masked region for the compiler to park registers in)
    if constexpr (NT == 3) lc[64 + (lane & 31)] = tp.c[64 + (lane & 31)];
  }
  wave_sync();

  // ---- operator tiles (identical in every wave) ----------------------------------------------------------------------
  Frag An[2][NT], Bn[2][NT], Ff;
  {
    const int rho = lane & 15, kg = lane >> 4, gp = rho >> 2, i = rho & 3;
    // (branch-free: an unconditional load from a clamped address, then a select -- under the per-lane `lag < 0 ? 0 : load`
    // the compiler parked live registers in accumulator registers inside the exec-masked load: tools/isa_spill_lint.py)
    auto cval = [&](int lag) -> float {
      const float c = lc[lag < 0 ? 0 : (lag > LCW - 1 ? LCW - 1 : lag)];
      return lag < 0 ? 0.0f : c;
    };
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int o = 0; o < NT; ++o) {
        float fa[8], fb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          fa[j] = cval(32 * o + 8 * (gp - kg) + 4 * r + i - j);
          fb[j] = cval(32 * o + 8 * (kg - gp) + j - 4 * r - i);
        }
        An[r][o] = split8(fa);
        Bn[r][o] = split8(fb);
      }
    float ff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) ff[j] = lc[LCW - 1];
    Ff = split8(ff);
  }
  const float s_far = lc[LCW - 1];                 // 2^a S: the far-field gain (every lag >= K-1)

  // ---- this wave's share of the problem; the scale comes from the WHOLE series ----------------------------------------
  const double lb = a.lbda_vec ? a.lbda_vec[p] : a.lbda;
  float ysn[NBW][8];
  double w[NBW][8];
  float sigma = 1.0f, inv_sigma = 1.0f;
  bool degenerate = false;                         // an all-zero series with a warm start: no scale to work at (fista_mfma.h)
  {
    const float* yrow = a.y + (int64_t)(p / a.y_rep) * a.ldy + 32 * qoff + tb;
    float m = 0.0f;
#pragma unroll
    for (int q = 0; q < NBW; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // (branch-free loads: a clamped address and a select -- fista_mfma2.h says why; only the last wave holds padding)
        float yv;
        if (LASTW && q >= NBW - 4) {
          const bool ok = 32 * q + j < nrem;
          const float yl = yrow[ok ? 32 * q + j : -tb];        // (sample 0 of the last wave's share always exists)
          yv = ok ? yl : 0.0f;
        } else {
          yv = yrow[32 * q + j];
        }
        ysn[q][j] = yv;
        m = fmaxf(m, fabsf(yv));
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    xm[k * 64] = m;
    wg_sync();
    m = fmaxf(fmaxf(xm[0], xm[64]), fmaxf(xm[128], xm[192]));
    {
      const bool okm = m > 0.0f && m < 3.0e38f;
      int e = 0;
      (void)frexpf(okm ? m : 1.0f, &e);
      const float sg = ldexpf(1.0f, a.ybits - e) / y_scale, isg = ldexpf(1.0f, e - a.ybits) * y_scale;
      sigma = okm ? sg : 1.0f;
      inv_sigma = okm ? isg : 1.0f;
      degenerate = !okm && !a.cold;                // (the same in every wave: m is the maximum over the whole series)
    }
    const float ys = -sigma * y_scale;
    const double* wrow = a.w + (int64_t)p * a.ldw + 32 * qoff + tb;
#pragma unroll
    for (int q = 0; q < NBW; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ysn[q][j] *= ys;
        w[q][j] = 0.0;
      }
    if (!a.cold) {                                // (wave-uniform: a scalar branch; the loads inside are branch-free)
#pragma unroll
      for (int q = 0; q < NBW; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (LASTW && q >= NBW - 4) {
            const bool ok = 32 * q + j < nrem;
            const double wl = wrow[ok ? 32 * q + j : -tb] * (double)sigma;
            w[q][j] = ok ? wl : 0.0;
          } else {
            w[q][j] = wrow[32 * q + j] * (double)sigma;
          }
        }
    }
  }
  const double th = lb * step * (double)sigma;
  const double nstep = -step * g_scale;
  float guard = 0.0f, wlast = 0.0f;
  double ldsq0 = 0.0, ldsq1 = 0.0, lwsq0 = 0.0, lwsq1 = 0.0;     // LOOPS: this lane's parts of ||d||^2 and ||w_{k+1}||^2
  bool lactive = true;                           // LOOPS: this problem has not met its rule yet
  // cost trace: 0.5 ||r''||^2 / (2^a sigma)^2 + lbda ||w'||_1 / sigma
  const float jq = 0.5f * (inv_sigma / y_scale) * (inv_sigma / y_scale), jl = (float)lb * inv_sigma;
  float jsq = 0.0f, jl1 = 0.0f;
  // CERT: lane group g tracks sample 3 of block CQ[g] of this wave's share
  constexpr int CQ0 = NBW / 8, CQ1 = (3 * NBW) / 8, CQ2 = (5 * NBW) / 8, CQ3 = (7 * NBW) / 8;
  const int cq_mine = ((2 * g + 1) * NBW) >> 3;    // = CQ0 .. CQ3 of lane group g (arithmetic: the chained selects became branches writing an accumulator register under partial exec masks)
  double cu = 0.0, cw = 0.0;
  float jw2 = 0.0f, cvsq = 0.0f;
  bool cflag = false;
  int cert_it = -1;
  constexpr float CP1 = 0.3133f, CP2 = 0.6467f, CP3 = 0.04f;
  const float cert_t2 = ((float)a.tol * 1.001f) * ((float)a.tol * 1.001f);
  const float cert_c0 = (float)th * (4.0f * 1.0001f) * __builtin_sqrtf(32.0f * NBT) + 3.1e-10f * sigma;
  const float cert_lim = cert_t2 * cert_c0 * cert_c0 * (1.0001f / CP3);

#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int o = 0; o < NT; ++o)
      asm volatile("" : "+a"(An[r][o].hi), "+a"(An[r][o].lo), "+a"(Bn[r][o].hi), "+a"(Bn[r][o].lo));
  asm volatile("" : "+a"(Ff.hi), "+a"(Ff.lo));
#pragma unroll
  for (int q = 0; q < NBW; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(ysn[q][j]));

  auto mfma_part = [](const Frag& A, const Frag& B, f4 acc, int part) __attribute__((always_inline)) -> f4 {
    return part == 0   ? __builtin_amdgcn_mfma_f32_16x16x32_f16(A.hi, B.hi, acc, 0, 0, 0)
           : part == 1 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(A.hi, B.lo, acc, 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_16x16x32_f16(A.lo, B.hi, acc, 0, 0, 0);
  };
  // what the right neighbour needs of the current iterate -- the fragments of this wave's last NX blocks (x = 0: the last) ...
  auto publish_block = [&](auto xc_) {
    constexpr int x = decltype(xc_)::value;
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = (float)w[NBW - 1 - x][j];
    const Frag f = split8(xv);
    xwb[((2 * k + x) * 2) * 64] = __builtin_bit_cast(u4, f.hi);
    xwb[((2 * k + x) * 2 + 1) * 64] = __builtin_bit_cast(u4, f.lo);
  };
  // ... and the float64 sums of this wave's blocks without / with the last NX ones (lane parts)
  auto publish_sums = [&](double se, double sl) {
    xsp[(2 * k) * 256 + 4 * v + g] = se;
    xsp[(2 * k + 1) * 256 + 4 * v + g] = se + sl;
  };

  // ---- forward: r = T_c w - y over this wave's blocks (ascending) ---------------------------------------------------
  auto forward = [&]() __attribute__((always_inline)) {
    f4 carry = f4{0.f, 0.f, 0.f, 0.f};
    Frag wfX[NX];                                  // the left neighbour's last NX blocks ([0]: the last)
    f2v rs2 = f2v{0.f, 0.f}, rs0 = f2v{0.f, 0.f};  // sums of the residual samples of blocks 1 .. / of block 0
    if constexpr (HAS_L) {
      // E_{k-1} + T_{k-2} + ... + T_0 (every lane of a problem adds the same parts in the same order)
      const double* pe = xsp + (2 * (k - 1)) * 256 + 4 * v;
      double s = (pe[0] + pe[1]) + (pe[2] + pe[3]);
#pragma unroll
      for (int j = k - 2; j >= 0; --j) {
        const double* pt = xsp + (2 * j + 1) * 256 + 4 * v;
        s += (pt[0] + pt[1]) + (pt[2] + pt[3]);
      }
      const float c = (float)(s * (double)s_far);
      carry = f4{c, c, c, c};
#pragma unroll
      for (int x = 0; x < NX; ++x) {
        wfX[x].hi = __builtin_bit_cast(h8, xwb[((2 * (k - 1) + x) * 2) * 64]);
        wfX[x].lo = __builtin_bit_cast(h8, xwb[((2 * (k - 1) + x) * 2 + 1) * 64]);
      }
    }
    Frag wf[NBW + 1];
    f4 acc[NBW + 1][2];
    unsigned ph[NBW + 1][4], pl[NBW + 1][4];
    unsigned rh[NBW][4], rl[NBW][4];
    if constexpr (WITH_J) { jsq = 0.0f; jl1 = 0.0f; }
    if constexpr (CERT) jw2 = 0.0f;
    // (the last wave's padding masks are made in the pass that uses them: hoisted out of the solve loop, the 32 compare
    // results would be 64 scalar registers held across it)
    int nr = nrem;
    if constexpr (LASTW) asm volatile("" : "+v"(nr));
    auto prep_pair = [&](auto qc, auto pc) {
      constexpr int q = decltype(qc)::value, pp = decltype(pc)::value;
      float x0, x1;
      asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x0) : "v"(w[q][2 * pp]));
      asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x1) : "v"(w[q][2 * pp + 1]));
      if constexpr (WITH_J) jl1 += fabsf(x0) + fabsf(x1);
      if constexpr (CERT) jw2 = fmaf(x1, x1, fmaf(x0, x0, jw2));
      split_pair(x0, x1, ph[q][pp], pl[q][pp]);
      if constexpr (pp == 3) {
        wf[q].hi = __builtin_bit_cast(h8, u4{ph[q][0], ph[q][1], ph[q][2], ph[q][3]});
        wf[q].lo = __builtin_bit_cast(h8, u4{pl[q][0], pl[q][1], pl[q][2], pl[q][3]});
      }
    };
    auto cinit = [&](auto qc, auto rc, const f4& cy) {
      constexpr int q = decltype(qc)::value, r = decltype(rc)::value;
      acc[q][r] = f4{cy[0] + ysn[q][4 * r + 0], cy[1] + ysn[q][4 * r + 1], cy[2] + ysn[q][4 * r + 2],
                     cy[3] + ysn[q][4 * r + 3]};
    };
    auto finish_pair = [&](auto qc, auto pc) {
      constexpr int q = decltype(qc)::value, pp = decltype(pc)::value;
      float x0 = acc[q][pp >> 1][(2 * pp) & 3], x1 = acc[q][pp >> 1][(2 * pp + 1) & 3];
      if constexpr (LASTW && q >= NBW - 4) {       // padding behind sample N-1 (the series ends in one of these blocks)
        x0 = (32 * q + 2 * pp < nr) ? x0 : 0.0f;
        x1 = (32 * q + 2 * pp + 1 < nr) ? x1 : 0.0f;
      }
      if constexpr (HAS_L) {
        if constexpr (q >= NX) rs2 += f2v{x0, x1};     // (the neighbour's near tiles reach this wave's first NX blocks)
        else rs0 += f2v{x0, x1};
      }
      if constexpr (WITH_J) jsq = fmaf(x1, x1, fmaf(x0, x0, jsq));
      split_pair(x0, x1, rh[q][pp], rl[q][pp]);
      if constexpr (pp == 3) {
        lrf[(2 * q) * 64] = u4{rh[q][0], rh[q][1], rh[q][2], rh[q][3]};
        lrf[(2 * q + 1) * 64] = u4{rl[q][0], rl[q][1], rl[q][2], rl[q][3]};
      }
    };
    static_for<0, 4>([&](auto pc) { prep_pair(std::integral_constant<int, 0>{}, pc); });
    cinit(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, carry);
    cinit(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, carry);
    static_for<0, NBW>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      f4 cn = carry;                               // carry of block q+1
      static_for<0, 3 + 6 * NT>([&](auto sc) {
        constexpr int sl = decltype(sc)::value;
        if constexpr (sl < 3) {                      // carry of block q+1: + S (sum of block q+1-NT)
          if constexpr (q + 1 < NBW) {
            if constexpr (q >= NT - 1) cn = mfma_part(Ff, wf[q >= NT - 1 ? q - (NT - 1) : 0], cn, sl);
            else if constexpr (HAS_L) cn = mfma_part(Ff, wfX[q < NT - 1 ? NT - 2 - q : 0], cn, sl);   // (block q+1-NT < 0: the left neighbour's)
          }
        } else {
          constexpr int c = sl - 3, r = c & 1, kk = c >> 1, o = kk / 3;        // near tile o: block q-o
          if constexpr (q >= o) acc[q][r] = mfma_part(An[r][o], wf[q >= o ? q - o : 0], acc[q][r], kk - 3 * o);
          else if constexpr (HAS_L) acc[q][r] = mfma_part(An[r][o], wfX[q < o ? o - q - 1 : 0], acc[q][r], kk - 3 * o);
        }
        if constexpr (sl < 4) {
          if constexpr (HAS_R && q + 1 >= NBW - NX && !WITH_J) {      // this wave's last NX blocks: split when they were updated
            // (with the cost trace their samples are converted again: ||w||_1 needs them)
            if constexpr (sl == 0) {
              constexpr int x = NBW - 1 - (q + 1);
              wf[q + 1].hi = __builtin_bit_cast(h8, xwb[((2 * k + x) * 2) * 64]);
              wf[q + 1].lo = __builtin_bit_cast(h8, xwb[((2 * k + x) * 2 + 1) * 64]);
            }
          } else if constexpr (q + 1 < NBW) prep_pair(std::integral_constant<int, q + 1>{}, sc);
        } else if constexpr (sl < 8) {
          if constexpr (q >= 1) finish_pair(std::integral_constant<int, q - 1>{}, std::integral_constant<int, sl - 4>{});
        } else if constexpr (sl == 10 || sl == 11) {
          if constexpr (q + 1 < NBW) cinit(std::integral_constant<int, q + 1>{}, std::integral_constant<int, sl - 10>{}, cn);
        }
      });
      carry = cn;
    });
    static_for<0, 4>([&](auto pc) { finish_pair(std::integral_constant<int, NBW - 1>{}, pc); });
    if constexpr (HAS_L) {                         // what the waves to the left need of this wave's residual
      const float re = rs2[0] + rs2[1], r0 = rs0[0] + rs0[1];
      xrp[(2 * k) * 256 + 4 * v + g] = re;
      xrp[(2 * k + 1) * 256 + 4 * v + g] = re + r0;
    }
  };

  // ---- adjoint and update: g = T_c^T r over this wave's blocks (descending) -----------------------------------------
  auto backward = [&](const double beta) __attribute__((always_inline)) {
    const double nb1 = -(1.0 + beta);
    f4 carry = f4{0.f, 0.f, 0.f, 0.f};
    Frag rfX[NX];                                  // the right neighbour's first NX blocks
    double sum0 = 0.0, sum1 = 0.0, suml = 0.0;     // sums of the updated iterate over blocks 0 .. NBW-2 / over block NBW-1
    if constexpr (HAS_R) {
      const float* pe = xrp + (2 * (k + 1)) * 256 + 4 * v;          // S (RE_{k+1} + RT_{k+2} + ...)
      float c = (pe[0] + pe[1]) + (pe[2] + pe[3]);
#pragma unroll
      for (int j = k + 2; j < NW; ++j) {
        const float* pt = xrp + (2 * j + 1) * 256 + 4 * v;
        c += (pt[0] + pt[1]) + (pt[2] + pt[3]);
      }
      c *= s_far;
      carry = f4{c, c, c, c};
#pragma unroll
      for (int x = 0; x < NX; ++x) {
        rfX[x].hi = __builtin_bit_cast(h8, lrf_next[(2 * x) * 64]);
        rfX[x].lo = __builtin_bit_cast(h8, lrf_next[(2 * x + 1) * 64]);
      }
    }
    f4 acc[NBW + 1][2];
    Frag rf[NBW + 2];
    auto fetch = [&](auto qc) {
      constexpr int q = decltype(qc)::value;
      rf[q].hi = __builtin_bit_cast(h8, lrf[(2 * q) * 64]);
      rf[q].lo = __builtin_bit_cast(h8, lrf[(2 * q + 1) * 64]);
    };
    auto update = [&](auto qc, auto jc) {
      constexpr int q = decltype(qc)::value, j = decltype(jc)::value;
      const double gj = (double)acc[q][j >> 2][j & 3];
      const double u = fma(nstep, gj, w[q][j]);
      const double d = fmin(fmax(u, -th), th);
      w[q][j] = fma(nb1, d, u);
      if constexpr (LOOPS) {
        if constexpr ((j & 1) == 0) { ldsq0 = fma(d, d, ldsq0); lwsq0 = fma(w[q][j], w[q][j], lwsq0); }
        else { ldsq1 = fma(d, d, ldsq1); lwsq1 = fma(w[q][j], w[q][j], lwsq1); }
      }
      if constexpr (HAS_R) {
        if constexpr (q >= NBW - NX) suml += w[q][j];
        else if constexpr ((j & 1) == 0) sum0 += w[q][j];
        else sum1 += w[q][j];
      }
      if constexpr (CERT && j == 3 && (q == CQ0 || q == CQ1 || q == CQ2 || q == CQ3)) {
        cu = (cq_mine == q) ? u : cu;
        cw = (cq_mine == q) ? w[q][j] : cw;
      }
    };
    fetch(std::integral_constant<int, NBW - 1>{});
    static_for<0, NBW>([&](auto qq) {
      constexpr int q = NBW - 1 - decltype(qq)::value;
      f4 cn = carry;                               // carry of block q-1
      if constexpr (q >= 1) fetch(std::integral_constant<int, q - 1>{});
      static_for<0, 3 + 6 * NT>([&](auto sc) {
        constexpr int sl = decltype(sc)::value;
        if constexpr (sl < 3) {                      // carry of block q-1: + S (sum of block q-1+NT)
          if constexpr (q >= 1) {
            if constexpr (q - 1 + NT < NBW) cn = mfma_part(Ff, rf[q - 1 + NT < NBW ? q - 1 + NT : 0], cn, sl);
            else if constexpr (HAS_R) cn = mfma_part(Ff, rfX[q - 1 + NT >= NBW ? q - 1 + NT - NBW : 0], cn, sl);
          }
        } else {
          constexpr int c = sl - 3, r = c & 1, kk = c >> 1, o = kk / 3;        // near tile o: block q+o
          if constexpr (kk == 0) acc[q][r] = mfma_part(Bn[r][0], rf[q], carry, 0);
          else if constexpr (q + o < NBW) acc[q][r] = mfma_part(Bn[r][o], rf[q + o < NBW ? q + o : 0], acc[q][r], kk - 3 * o);
          else if constexpr (HAS_R) acc[q][r] = mfma_part(Bn[r][o], rfX[q + o >= NBW ? q + o - NBW : 0], acc[q][r], kk - 3 * o);
        }
        if constexpr ((sl & 1) == 0 && sl < 16 && q + 1 < NBW)
          update(std::integral_constant<int, q + 1>{}, std::integral_constant<int, sl / 2>{});
      });
      carry = cn;
      // a block is complete once the block before it has run: the fragments the neighbour needs go out at once
      if constexpr (HAS_R && q == NBW - 2) publish_block(std::integral_constant<int, 0>{});
      if constexpr (HAS_R && NX == 2 && q == NBW - 3) publish_block(std::integral_constant<int, 1>{});
    });
    static_for<0, 8>([&](auto jc) { update(std::integral_constant<int, 0>{}, jc); });
    if constexpr (HAS_R) publish_sums(sum0 + sum1, suml);
    if constexpr (CERT) {                          // the window combination on this lane's tracked sample (fista_mfma.h)
      const float d1 = lt[((cert_it + 3) & 3) * 256], d2 = lt[((cert_it + 2) & 3) * 256], d3 = lt[((cert_it + 1) & 3) * 256];
      const unsigned ulo = __builtin_bit_cast(unsigned, lt[4 * 256]), uhi = __builtin_bit_cast(unsigned, lt[5 * 256]);
      const double up = __builtin_bit_cast(double, ((unsigned long long)uhi << 32) | ulo);
      const float dk = (float)(cu - up), e = (float)(cw - cu);
      const float vv = fmaf(2.0f, d2, fmaf(3.0f, d1, fmaf(2.0f, dk, e))) + d3;
      const float mm = fmaf(2.0f, fabsf(d2), fmaf(3.0f, fabsf(d1), fmaf(2.0f, fabsf(dk), fabsf(e)))) + fabsf(d3);
      const float vs = fmaxf(fmaf(-0x1p-21f, mm, fabsf(vv)), 0.0f);
      cvsq = vs * vs;
      lt[(cert_it & 3) * 256] = dk;
      const unsigned long long ub = __builtin_bit_cast(unsigned long long, cu);
      lt[4 * 256] = __builtin_bit_cast(float, (unsigned)ub);
      lt[5 * 256] = __builtin_bit_cast(float, (unsigned)(ub >> 32));
    }
  };
  auto range_check = [&]() {
    unsigned mb = 0;
#pragma unroll
    for (int q = 0; q < NBW; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) mb = max(mb, __builtin_bit_cast(unsigned, (float)w[q][j]) & 0x7fffffffu);
    const float m = mb >= 0x7f800000u ? 65504.0f : __builtin_bit_cast(float, mb);
    wlast = m;
    unsigned e = 0;
#pragma unroll
    for (int q = 0; q < NBW; ++q) {
      const u4 h = lrf[(2 * q) * 64];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        e = max(e, h[c] & 0x7fffu);
        e = max(e, (h[c] >> 16) & 0x7fffu);
      }
    }
    guard = __builtin_fmaxf(guard, __builtin_fmaxf(m, e >= 0x7800u ? 65504.0f : 0.0f));
  };

  // ---- iterations: both passes in all waves at once, one barrier per phase boundary -----------------------------------
  if constexpr (HAS_R) {                           // the start iterate's contribution to the waves on the right
    publish_block(std::integral_constant<int, 0>{});
    if constexpr (NX == 2) publish_block(std::integral_constant<int, 1>{});
    double se = 0.0, sl = 0.0;
#pragma unroll
    for (int q = 0; q < NBW; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (q >= NBW - NX) sl += w[q][j];
        else se += w[q][j];
      }
    publish_sums(se, sl);
  }
  wg_sync();
  if constexpr (!WITH_J) {
    for (int it = 0; it < a.n_iter; ++it) {
      const double beta = a.betas[it];
      forward();
      wg_sync();                                   // residual fragments and their sums are out
      if constexpr (LOOPS) { ldsq0 = ldsq1 = lwsq0 = lwsq1 = 0.0; }
      backward(beta);
      if constexpr (LOOPS) {                       // this wave's share of the rule's two norms (fista_mfma.h)
        double num = ldsq0 + ldsq1, den = lwsq0 + lwsq1;
        num += __shfl_xor(num, 16, 64);
        den += __shfl_xor(den, 16, 64);
        num += __shfl_xor(num, 32, 64);
        den += __shfl_xor(den, 32, 64);
        xl[(2 * k) * 64] = num;
        xl[(2 * k + 1) * 64] = den;
      }
      if (PB_MFMA_CHECKS && (((it & 7) == 7) || (it == a.n_iter - 1) || (it == 0 && !a.cold))) range_check();
      wg_sync();                                   // the updated iterate's fragments and sums are out
      if constexpr (LOOPS) {
        // every wave adds the shares in the same order: the same verdict everywhere (the branches below are workgroup-uniform)
        const double num = (xl[0 * 64] + xl[2 * 64]) + (xl[4 * 64] + xl[6 * 64]);
        const double den = (xl[1 * 64] + xl[3 * 64]) + (xl[5 * 64] + xl[7 * 64]);
        // (the criterion in EVERY lane, pinned: under `lactive &&` the compiler evaluated it in an exec-masked region and
        // parked live registers in accumulator registers there -- the pattern tools/isa_spill_lint.py refuses)
        double crit = (1.0 + beta) * sqrt(num) / (sqrt(den) + 1.0e-10 * (double)sigma);
        asm volatile("" : "+v"(crit));
        const bool fire = lactive && it >= 3 && crit < a.tol;
        if (__builtin_amdgcn_ballot_w64(fire) != 0) {         // (rare: at most once per problem)
          range_check();                                      // this moment's operands, for the problems that finish now
          float gq = guard, wq = wlast;
          gq = fmaxf(gq, __shfl_xor(gq, 16, 64));
          gq = fmaxf(gq, __shfl_xor(gq, 32, 64));
          wq = fmaxf(wq, __shfl_xor(wq, 16, 64));
          wq = fmaxf(wq, __shfl_xor(wq, 32, 64));
          xg[(k * 2 + 0) * 64] = gq;
          xg[(k * 2 + 1) * 64] = wq;
          wg_sync();
          bool in_range = true;
          float wm = 0.0f;
#pragma unroll
          for (int j = 0; j < NW; ++j) {
            in_range = in_range && (xg[(j * 2 + 0) * 64] < 60000.0f);
            wm = fmaxf(wm, xg[(j * 2 + 1) * 64]);
          }
          const bool badq = !in_range || (a.rho_guard && wm > 0.0f && (float)th > MFMA_RHO_MAX * wm) || degenerate;
          if (fire) {
            lactive = false;
            if (live && !badq) {
              // (addresses made HERE, from laundered values: hoisted out of the loop they cost registers through the whole solve)
              double* wrow = a.w + (int64_t)p * a.ldw + 32 * qoff + tb;
              int nrv = nrem;
              asm volatile("" : "+v"(wrow), "+v"(nrv));
#pragma unroll
              for (int q = 0; q < NBW; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                  if (!(LASTW && q >= NBW - 4) || 32 * q + j < nrv) wrow[32 * q + j] = w[q][j] * (double)inv_sigma;
            }
            if (!HAS_L && live && a.n_done && g == 0) a.n_done[p] = badq ? -1 : it + 1;
          }
          if (__builtin_amdgcn_ballot_w64(lactive && live) == 0) return;       // every problem of the workgroup has finished (all waves agree)
          wg_sync();                                 // (the guard area is read again at the next finish)
        }
      }
    }
  } else {
    // rotated: the cost of iterate k+1 comes from the residual of the NEXT forward pass (one pass in front)
    forward();
    if constexpr (CERT) lt[6 * 256] = jw2;         // this lane's part of ||w_0||^2
    wg_sync();
    for (int it = 0; it < a.n_iter; ++it) {
      const double beta = a.betas[it];
      cert_it = it;
      backward(beta);
      wg_sync();
      forward();
      float sq = jsq, l1 = jl1;                     // this lane's samples -> this wave's share of the problem
      sq += __shfl_xor(sq, 16, 64);
      l1 += __shfl_xor(l1, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      l1 += __shfl_xor(l1, 32, 64);
      float t = 0.0f;
      if constexpr (CERT) {
        // this share's part of  sum v^2 - tol^2 (||w_k||^2 / p1 + 4 ||w_{k+1}||^2 / p2)
        t = cvsq - cert_t2 * ((1.0001f / CP1) * lt[6 * 256] + (4.0001f / CP2) * jw2);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        lt[6 * 256] = jw2;
      }
      xj[(k * 3 + 0) * 64] = sq;
      xj[(k * 3 + 1) * 64] = l1;
      if constexpr (CERT) xj[(k * 3 + 2) * 64] = t;
      wg_sync();                                   // (the barrier of the forward pass: the shares of the cost meet here)
      // (every wave adds the four shares in the same order: the same cost and the same verdict everywhere)
      sq = (xj[0 * 64] + xj[3 * 64]) + (xj[6 * 64] + xj[9 * 64]);
      l1 = (xj[1 * 64] + xj[4 * 64]) + (xj[7 * 64] + xj[10 * 64]);
      if (!HAS_L && live && g == 0 && (!CERT || (a.J != nullptr && !cflag))) a.J[(int64_t)p * a.ldj + it] = fmaf(jq, sq, jl * l1);
      if constexpr (CERT) {
        t = (xj[2 * 64] + xj[5 * 64]) + (xj[8 * 64] + xj[11 * 64]);
        cflag = cflag | ((it >= 7) & !(t >= cert_lim));      // NaN-safe; the same verdict in every wave
      }
      if (PB_MFMA_CHECKS && (((it & 7) == 7) || (it == a.n_iter - 1) || (it == 0 && !a.cold))) range_check();
    }
  }

  // ---- guards over the whole series, store ---------------------------------------------------------------------------
  guard = fmaxf(guard, __shfl_xor(guard, 16, 64));
  guard = fmaxf(guard, __shfl_xor(guard, 32, 64));
  wlast = fmaxf(wlast, __shfl_xor(wlast, 16, 64));
  wlast = fmaxf(wlast, __shfl_xor(wlast, 32, 64));
  xg[(k * 2 + 0) * 64] = guard;
  xg[(k * 2 + 1) * 64] = wlast;
  wg_sync();
  {
    bool in_range = true;                          // NaN in any share: out of range
    float gm = 0.0f, wm = 0.0f;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const float gj = xg[(j * 2 + 0) * 64];
      in_range = in_range && (gj < 60000.0f);
      gm = fmaxf(gm, gj);
      wm = fmaxf(wm, xg[(j * 2 + 1) * 64]);
    }
    guard = in_range ? gm : 65504.0f;
    wlast = wm;
  }
  const bool bad = !(guard < 60000.0f) || degenerate || (a.rho_guard && wlast > 0.0f && (float)th > MFMA_RHO_MAX * wlast) || (CERT && cflag);
  if (live && !bad && (!LOOPS || lactive)) {
    double* wrow = a.w + (int64_t)p * a.ldw + 32 * qoff + tb;
#pragma unroll
    for (int q = 0; q < NBW; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (!(LASTW && q >= NBW - 4) || 32 * q + j < nrem) wrow[32 * q + j] = w[q][j] * (double)inv_sigma;
      }
  }
  if (!HAS_L && live && a.n_done && g == 0 && (!LOOPS || lactive)) a.n_done[p] = bad ? -1 : a.n_iter;
}

// one workgroup = four waves = 16 problems; the wave index picks the share (scalar branches: each wave runs one role)
template <int A, bool TAPS_DEV = false, bool WITH_J = false, bool CERT = false, bool LOOPS = false, int NT = 2>
__global__ __launch_bounds__(256) void fista_mfma4_kernel(FistaArgs a, MfmaTaps tp) {
  extern __shared__ __attribute__((aligned(16))) char mf4_smem[];
  if (a.range) {                                   // a candidate launch of a device-side plan: workgroups beyond its slots leave
    if ((int)blockIdx.x * 16 + a.range[0] >= a.range[1]) return;      // (all waves: before any barrier)
  }
  const int k = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (k == 0) mfma4_role<A, 0, TAPS_DEV, WITH_J, CERT, LOOPS, NT>(a, tp, mf4_smem);
  else if (k == 1) mfma4_role<A, 1, TAPS_DEV, WITH_J, CERT, LOOPS, NT>(a, tp, mf4_smem);
  else if (k == 2) mfma4_role<A, 2, TAPS_DEV, WITH_J, CERT, LOOPS, NT>(a, tp, mf4_smem);
  else mfma4_role<A, 3, TAPS_DEV, WITH_J, CERT, LOOPS, NT>(a, tp, mf4_smem);
}

// Plain solves, with or without the cost trace, the window rule (wind = 6) as a no-fire certificate, the _loops_deconv rule
// in full (no cost trace); HRFs of up to 33 taps -- 34 .. 65 with three near tiles --; 128 (A - 1) < N <= 128 A: the series ends in one of the last wave's last four blocks.  Shared HRF in device memory
// (the blind step's z-step): plain only.
template <int A>
int launch_mfma4(const FistaArgs& a, const double* taps, int K, bool with_j, hipStream_t st) {
  if (a.N > 128 * A || a.N <= 128 * (A - 1) || K < 1 || K > 65) return 1;
  const bool three = K > 33;                       // three near tiles: plain solves and the cost trace only
  const bool cert = a.stop_mode == PB_STOP_WINDOW, loops = a.stop_mode == PB_STOP_LOOPS;
  if (!a.n_done) return 1;
  if ((with_j || cert || loops) && a.taps_pp) return 1;
  if (loops && with_j) return 1;                   // (the _loops_deconv rule: plain variant, as on the one-wave form)
  const int64_t groups = (launch_count(a) + 15) / 16;
  const dim3 grid((unsigned)groups), block(256);
  const size_t lds = mfma4_lds_bytes(A);
  auto go = [&](void (*kernel)(FistaArgs, MfmaTaps)) {
    // (more than 64 KB of dynamic LDS has to be asked for, per kernel and device: a host-side call of about a microsecond)
    if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, grid, block, lds, st, a, a.taps_pp ? MfmaTaps{} : make_mfma_taps(taps, K));
  };
  if (three && a.taps_pp) go(fista_mfma4_kernel<A, true, false, false, false, 3>);
  else if (three) go(loops ? fista_mfma4_kernel<A, false, false, false, true, 3> : cert ? fista_mfma4_kernel<A, false, true, true, false, 3>
                     : (with_j ? fista_mfma4_kernel<A, false, true, false, false, 3> : fista_mfma4_kernel<A, false, false, false, false, 3>));
  else if (a.taps_pp) go(fista_mfma4_kernel<A, true>);
  else if (loops) go(fista_mfma4_kernel<A, false, false, false, true>);
  else if (cert) go(fista_mfma4_kernel<A, false, true, true>);
  else if (with_j) go(fista_mfma4_kernel<A, false, true, false>);
  else go(fista_mfma4_kernel<A, false>);
  return 0;
}

}  // namespace pb
