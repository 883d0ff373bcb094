// The matrix-pipe form of the fused FISTA kernel (fista_mfma.h) with ONE series split over the TWO waves of a
// workgroup: wave 0 ("left") owns blocks 0 .. NBA-1 of 32 samples, wave 1 ("right") blocks NBA .. NBA+NBB-1 of the
// same 16 problems (NBB = NBA or NBA + 1: the left wave has the extra work of the exchange -- a float64 sum of its
// updated iterate -- so an odd block goes to the right wave and that work hides in the left wave's slack).  Two uses:
//   * series of 321 .. 640 scans (11 .. 20 blocks): they do not fit one wave (the float64 iterate alone would be
//     304+ registers) -- the reference's own shipped demo is 600 scans (examples/synth_data/deconv.py:46);
//   * small batches and remainders of 129 .. 320 scans: a pass of the one-wave form lasts what 16 problems x NB
//     blocks last on ONE SIMD whatever the batch size; split over two SIMDs an iteration takes about half as long
//     (8 192 problems per pass instead of 16 384).
//
// The operator is the one of fista_mfma.h (T_c = near band + constant far field, two near tiles, K <= 33), so the
// two halves are coupled through very little, and both passes of an iteration run in BOTH waves at the same time:
//   forward  (x = T_c w - y, ascending blocks): the right wave needs the float16 fragment of the left wave's LAST
//            block (its near tile reaches one block back) and the far field of everything before it,
//            S * sum(w over blocks 0 .. NBA-2) -- one scalar per problem;
//   adjoint  (g = T_c^T r, descending blocks): the left wave needs the residual fragment of the right wave's FIRST
//            block (already in LDS: every residual fragment waits there for the adjoint pass) and
//            S * sum(r over blocks NBA+1 ..) -- one scalar per problem.
// Those four items are produced at the END of the pass before: the left wave adds up its updated iterate in
// float64 while it updates it (the order of a sum does not matter, so the descending adjoint pass can make the
// ascending pass's prefix) and splits its last block -- the first one it updates -- at once; the right wave adds
// up its residual samples as it finishes them.  Two workgroup barriers per iteration (phase boundaries), 2.3 KB of
// LDS exchanged.  Everything else -- scaling, float16 split, guards, exact re-solve by capi.hip -- is fista_mfma.h's.
//
// Reference: pybold/bold_signal.py:62-72, pybold/linear.py:73-113, pybold/convolution.py:105-132.
#pragma once
#include "fista_mfma.h"

namespace pb {

// bytes of dynamic LDS of one workgroup (two waves; nbm = blocks of the larger half): residual fragments, the exchange
// areas (fragment, far fields, scale, guards, cost-trace parts), the taps, the certificate's per-lane state
constexpr size_t mfma2_lds_bytes(int nbm) {
  return ((size_t)2 * nbm * 2 * 64 + 2 * 64) * sizeof(u4) +
         (size_t)(2 * 64 + 64 + 64 + 2 * 64 + 4 * 64 + 6 * 64 + 7 * 128) * sizeof(float) + (size_t)2 * 2 * 64 * sizeof(double) +
         (size_t)256 * sizeof(double) + (size_t)256 * sizeof(float) + (size_t)2 * 64 * sizeof(u4) + (size_t)2 * 96 * sizeof(float);
}

// ROLE 0: the left wave (blocks 0 .. NBA-1), ROLE 1: the right wave (blocks NBA .. NBA+NBB-1; padding in its last block)
// WITH_J: cost trace (the loop rotated as in fista_mfma.h; each wave adds up its half, the halves meet in LDS at the
//   barrier that is there anyway).  CERT: the window rule (wind = 6) as the no-fire certificate of fista_mfma.h, four
//   tracked samples per WAVE (eight per problem); implies WITH_J; a problem that cannot be cleared is handed back.
// LOOPS: the _loops_deconv stop rule in full, as on the one-wave form (fista_mfma.h): each wave adds up its half of the two
//   norms next to the update, the halves meet in LDS at the barrier that ends the iteration; plain variant only.
// NT: near tiles -- 2 (K <= 33), or 3 (K <= 65: the tiles reach TWO blocks across the cut).
template <int NBA, int NBB, bool TAPS_DEV, int ROLE, bool WITH_J = false, bool CERT = false, bool LOOPS = false, int NT = 2>
__device__ __forceinline__ void mfma2_role(const FistaArgs& a, const MfmaTaps& tp, char* smem) {
  static_assert(!CERT || (WITH_J && !TAPS_DEV), "the certificate runs in the rotated (cost trace) loop");
  static_assert(!LOOPS || (!WITH_J && !TAPS_DEV && !CERT), "the _loops_deconv rule rides the plain variant");
  static_assert(NBB >= NBA && NBB <= NBA + 1 && NBA >= 2 && NBB <= 10, "right half = the larger one; two blocks at least per wave");
  constexpr int NBM = NBB;                         // blocks of the larger half: the size of a wave's fragment area
  static_assert(NT == 2 || (NT == 3 && NBA > 3), "three near tiles: four blocks at least per wave");
  constexpr int LCW = 32 * NT, NX = NT - 1;        // cumulative taps kept: lags 0 .. 32 NT - 1; NX: blocks of the neighbour the near tiles reach into
  constexpr int NBW = ROLE == 0 ? NBA : NBB;       // blocks of this wave
  constexpr int QOFF = ROLE == 0 ? 0 : NBA;        // its first block within the series
  constexpr int NBT = NBA + NBB;
  const int lane = threadIdx.x & 63;
  const int v = lane & 15, g = lane >> 4;
  int s0, s1;                                      // this launch's slots of its list (fista_fast.h: launch_slots)
  launch_slots(a, s0, s1);
  bool live;
  const int p = slot_to_problem(a, (int)blockIdx.x * 16 + v + s0, s1, live);
  const int tb = 8 * g;

  // ---- LDS: residual fragments of both waves, the exchange areas, the taps --------------------------------
  u4* const lbase = reinterpret_cast<u4*>(smem);
  u4* const lrf = lbase + ROLE * (NBM * 2 * 64) + lane;            // this wave's residual fragments
  u4* const lrf_right = lbase + (NBM * 2 * 64) + lane;             // the right wave's (its block 0: what the left wave reads)
  u4* const xw = lbase + 2 * (NBM * 2 * 64) + lane;                // [2][64]: fragment (hi, lo) of the left wave's last block
  float* const fbase = reinterpret_cast<float*>(lbase + 2 * (NBM * 2 * 64) + 2 * 64);
  // (areas below are laid out for LCW = 64; with three near tiles the taps live behind everything else, lc3)
  // (fbase + 2 LCW .. + 128: round 4's far-field scalars; since round 5 they cross as lane parts, xcp / xrp below)
  float* const xm = fbase + 128 + 128;                         // [2][64] max |y| of each half
  float* const xg = fbase + 128 + 256;                         // [2][2][64] guard, largest |w| of each half
  float* const xj = fbase + 128 + 512;                         // [2][3][64] cost-trace parts of each half: ||r||^2, ||w||_1, certificate
  float* const lt = fbase + 128 + 896 + threadIdx.x;           // [7][128] certificate state of every lane (as fista_mfma.h)
  double* const xl = reinterpret_cast<double*>(fbase + 128 + 896 + 7 * 128) + lane;     // [2][2][64] _loops_deconv rule: ||d||^2, ||w'||^2 of each half
  // The two far-field scalars cross the cut as the four LANE PARTS of each problem (slot 4 v + g), added up by the wave that
  // reads them: a lane-crossing sum at the END of a pass is two dependent ds_bpermute round trips with nothing left to
  // overlap them (round 5: ~250 cycles per pass, 10 % of an iteration of five-block waves); at the START of the next pass the
  // reads hide behind the float16 split of the first block.
  double* const xcp = reinterpret_cast<double*>(fbase + 128 + 896 + 7 * 128) + 2 * 2 * 64;   // [16][4] left -> right: sum(w, blocks 0 .. NBA-2), lane parts
  float* const xrp = reinterpret_cast<float*>(xcp + 256);                                         // [16][4] right -> left: sum(r, blocks NBA+1 ..), lane parts
  u4* const xw2 = reinterpret_cast<u4*>(xrp + 256) + lane;                                        // [2][64] three near tiles: the left wave's block before its last
  float* const lc = NT == 2 ? fbase + ROLE * 64 : reinterpret_cast<float*>(xw2 - lane + 2 * 64) + ROLE * 96;   // cumulative taps, one copy per wave
  if constexpr (CERT) {
#pragma unroll
    for (int q = 0; q < 7; ++q) lt[q * 128] = 0.0f;
  }
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto wg_sync = [] {                                // both waves: everything written to LDS before is visible after
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };

  // ---- cumulative taps (as fista_mfma.h) ---------------------------------------------------------------------
  double step = a.step, g_scale = tp.g_scale;
  float y_scale = tp.y_scale;
  if constexpr (TAPS_DEV) {
    double run = 0.0, run2 = 0.0;
    for (int k = 0; k <= lane && k < a.K; ++k) run += (double)(float)a.taps_pp[k];
    float cm = fabsf((float)run);
    if constexpr (NT == 3) {                       // lags 64 .. 95, every lane for lag 64 + (lane & 31): the store below stays unmasked
      for (int k = 0; k <= 64 + (lane & 31) && k < a.K; ++k) run2 += (double)(float)a.taps_pp[k];
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

  // ---- operator tiles (identical in both waves) ----------------------------------------------------------------
  Frag An[2][NT], Bn[2][NT], Ff;
  {
    const int rho = lane & 15, kg = lane >> 4, gp = rho >> 2, i = rho & 3;
    auto cval = [&](int lag) -> float { return lag < 0 ? 0.0f : lc[lag > LCW - 1 ? LCW - 1 : lag]; };
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

  // ---- this wave's part of the problem; the scale comes from the WHOLE series ---------------------------------
  const double lb = a.lbda_vec ? a.lbda_vec[p] : a.lbda;
  float ysn[NBW][8];
  double w[NBW][8];
  float sigma = 1.0f, inv_sigma = 1.0f;
  bool degenerate = false;                         // an all-zero series with a warm start: no scale to work at (fista_mfma.h)
  {
    const float* yrow = a.y + (int64_t)(p / a.y_rep) * a.ldy;
    float m = 0.0f;
#pragma unroll
    for (int q = 0; q < NBW; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // (branch-free loads: a clamped address and a select.  With a per-lane `if` around each load this compiler
        // placed a VGPR -> AGPR spill INSIDE one of the exec-masked regions; on a cold start that mask is empty, the
        // spill never happened and its reload -- an LDS address -- was garbage: the certificate variant at 17+ blocks)
        // only the LAST block of the series can hold padding (32 (NBT-1) < N): every other load is unconditional
        const int t = 32 * (QOFF + q) + tb + j;
        float yv;
        if (QOFF + q == NBT - 1) {
          const float yl = yrow[t < a.N ? t : a.N - 1];
          yv = (t < a.N) ? yl : 0.0f;
        } else {
          yv = yrow[t];
        }
        ysn[q][j] = yv;
        m = fmaxf(m, fabsf(yv));
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    xm[ROLE * 64 + lane] = m;
    wg_sync();
    m = fmaxf(m, xm[(1 - ROLE) * 64 + lane]);
    {                                            // (branch-free: see the note on the loads above)
      const bool okm = m > 0.0f && m < 3.0e38f;
      int e = 0;
      (void)frexpf(okm ? m : 1.0f, &e);
      const float sg = ldexpf(1.0f, a.ybits - e) / y_scale, isg = ldexpf(1.0f, e - a.ybits) * y_scale;
      sigma = okm ? sg : 1.0f;
      inv_sigma = okm ? isg : 1.0f;
      degenerate = !okm && !a.cold;                // (the same in both waves: m is the maximum over the whole series)
    }
    const float ys = -sigma * y_scale;
    const double* wrow = a.w + (int64_t)p * a.ldw;
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
          const int t = 32 * (QOFF + q) + tb + j;
          if (QOFF + q == NBT - 1) {
            const double wl = wrow[t < a.N ? t : a.N - 1] * (double)sigma;
            w[q][j] = (t < a.N) ? wl : 0.0;
          } else {
            w[q][j] = wrow[t] * (double)sigma;
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
  // CERT: lane group g tracks sample 3 of block CQ[g] of this wave's half
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
  // left wave: what the right wave needs of the current iterate -- the fragments of its last NX blocks (x = 0: the last) ...
  auto publish_block = [&](auto xc_) {
    constexpr int x = decltype(xc_)::value;
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = (float)w[NBW - 1 - x][j];
    const Frag f = split8(xv);
    u4* const dst = x == 0 ? xw : xw2;
    dst[0] = __builtin_bit_cast(u4, f.hi);
    dst[64] = __builtin_bit_cast(u4, f.lo);
  };
  // ... and the far field of the blocks before it (float64 sum over this lane's samples -> the problem's four lanes)
  auto publish_far_field = [&](double s) { xcp[4 * v + g] = s; };

  // ---- forward: r = T_c w - y over this wave's blocks (ascending) ---------------------------------------------
  auto forward = [&]() __attribute__((always_inline)) {
    f4 carry = f4{0.f, 0.f, 0.f, 0.f};
    Frag wfX[NX];                                  // right wave: the left wave's last NX blocks ([0]: the last)
    f2v rs2 = f2v{0.f, 0.f};                       // right wave: sum of the residual samples of its blocks 1 ..
    if constexpr (ROLE == 1) {
      const double* pc = xcp + 4 * v;               // (every lane of a problem adds the same four parts in the same order)
      const float c = (float)(((pc[0] + pc[1]) + (pc[2] + pc[3])) * (double)s_far);
      carry = f4{c, c, c, c};
      wfX[0].hi = __builtin_bit_cast(h8, xw[0]);
      wfX[0].lo = __builtin_bit_cast(h8, xw[64]);
      if constexpr (NX == 2) {
        wfX[1].hi = __builtin_bit_cast(h8, xw2[0]);
        wfX[1].lo = __builtin_bit_cast(h8, xw2[64]);
      }
    }
    Frag wf[NBW + 1];
    f4 acc[NBW + 1][2];
    unsigned ph[NBW + 1][4], pl[NBW + 1][4];
    unsigned rh[NBW][4], rl[NBW][4];
    if constexpr (WITH_J) { jsq = 0.0f; jl1 = 0.0f; }
    if constexpr (CERT) jw2 = 0.0f;
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
      if constexpr (QOFF + q == NBT - 1) {         // padding behind sample N-1 (last block of the series only)
        x0 = (32 * (QOFF + q) + tb + 2 * pp < a.N) ? x0 : 0.0f;
        x1 = (32 * (QOFF + q) + tb + 2 * pp + 1 < a.N) ? x1 : 0.0f;
      }
      if constexpr (ROLE == 1 && q >= NX) rs2 += f2v{x0, x1};     // (the left wave's near tiles reach this wave's first NX blocks)
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
            else if constexpr (ROLE == 1) cn = mfma_part(Ff, wfX[q < NT - 1 ? NT - 2 - q : 0], cn, sl);   // (block q+1-NT < 0: the left wave's)
          }
        } else {
          constexpr int c = sl - 3, r = c & 1, k = c >> 1, o = k / 3;          // near tile o: block q-o
          if constexpr (q >= o) acc[q][r] = mfma_part(An[r][o], wf[q >= o ? q - o : 0], acc[q][r], k - 3 * o);
          else if constexpr (ROLE == 1) acc[q][r] = mfma_part(An[r][o], wfX[q < o ? o - q - 1 : 0], acc[q][r], k - 3 * o);
        }
        if constexpr (sl < 4) {
          if constexpr (ROLE == 0 && q + 1 >= NBW - NX && !WITH_J) {  // the left wave's last NX blocks: split when they were updated (xw)
            // (with the cost trace their samples are converted again: ||w||_1 needs them)
            if constexpr (sl == 0) {
              const u4* const src = (NBW - 1 - (q + 1)) == 0 ? xw : xw2;
              wf[q + 1].hi = __builtin_bit_cast(h8, src[0]);
              wf[q + 1].lo = __builtin_bit_cast(h8, src[64]);
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
    if constexpr (ROLE == 1) {                     // what the left wave's adjoint pass needs of the far blocks
      xrp[4 * v + g] = rs2[0] + rs2[1];
    }
  };

  // ---- adjoint and update: g = T_c^T r over this wave's blocks (descending) -----------------------------------
  auto backward = [&](const double beta) __attribute__((always_inline)) {
    const double nb1 = -(1.0 + beta);
    f4 carry = f4{0.f, 0.f, 0.f, 0.f};
    Frag rfX[NX];                                  // left wave: the right wave's first NX blocks
    double sum0 = 0.0, sum1 = 0.0;                 // left wave: sum of the updated iterate over blocks 0 .. NBW-2
    if constexpr (ROLE == 0) {
      const float* pr = xrp + 4 * v;
      const float c = ((pr[0] + pr[1]) + (pr[2] + pr[3])) * s_far;
      carry = f4{c, c, c, c};
#pragma unroll
      for (int x = 0; x < NX; ++x) {
        rfX[x].hi = __builtin_bit_cast(h8, lrf_right[(2 * x) * 64]);
        rfX[x].lo = __builtin_bit_cast(h8, lrf_right[(2 * x + 1) * 64]);
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
      if constexpr (ROLE == 0 && q <= NBW - 1 - NX) {
        if constexpr ((j & 1) == 0) sum0 += w[q][j]; else sum1 += w[q][j];
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
            else if constexpr (ROLE == 0) cn = mfma_part(Ff, rfX[q - 1 + NT >= NBW ? q - 1 + NT - NBW : 0], cn, sl);
          }
        } else {
          constexpr int c = sl - 3, r = c & 1, k = c >> 1, o = k / 3;          // near tile o: block q+o
          if constexpr (k == 0) acc[q][r] = mfma_part(Bn[r][0], rf[q], carry, 0);
          else if constexpr (q + o < NBW) acc[q][r] = mfma_part(Bn[r][o], rf[q + o < NBW ? q + o : 0], acc[q][r], k - 3 * o);
          else if constexpr (ROLE == 0) acc[q][r] = mfma_part(Bn[r][o], rfX[q + o >= NBW ? q + o - NBW : 0], acc[q][r], k - 3 * o);
        }
        if constexpr ((sl & 1) == 0 && sl < 16 && q + 1 < NBW)
          update(std::integral_constant<int, q + 1>{}, std::integral_constant<int, sl / 2>{});
      });
      carry = cn;
      // the left wave's last block is complete once the block before it has run: its fragment goes out at once
      if constexpr (ROLE == 0 && q == NBW - 2) publish_block(std::integral_constant<int, 0>{});
      if constexpr (ROLE == 0 && NX == 2 && q == NBW - 3) publish_block(std::integral_constant<int, 1>{});
    });
    static_for<0, 8>([&](auto jc) { update(std::integral_constant<int, 0>{}, jc); });
    if constexpr (ROLE == 0) publish_far_field(sum0 + sum1);
    if constexpr (CERT) {                          // the window combination on this lane's tracked sample (fista_mfma.h)
      const float d1 = lt[((cert_it + 3) & 3) * 128], d2 = lt[((cert_it + 2) & 3) * 128], d3 = lt[((cert_it + 1) & 3) * 128];
      const unsigned ulo = __builtin_bit_cast(unsigned, lt[4 * 128]), uhi = __builtin_bit_cast(unsigned, lt[5 * 128]);
      const double up = __builtin_bit_cast(double, ((unsigned long long)uhi << 32) | ulo);
      const float dk = (float)(cu - up), e = (float)(cw - cu);
      const float v = fmaf(2.0f, d2, fmaf(3.0f, d1, fmaf(2.0f, dk, e))) + d3;
      const float m = fmaf(2.0f, fabsf(d2), fmaf(3.0f, fabsf(d1), fmaf(2.0f, fabsf(dk), fabsf(e)))) + fabsf(d3);
      const float vs = fmaxf(fmaf(-0x1p-21f, m, fabsf(v)), 0.0f);
      cvsq = vs * vs;
      lt[(cert_it & 3) * 128] = dk;
      const unsigned long long ub = __builtin_bit_cast(unsigned long long, cu);
      lt[4 * 128] = __builtin_bit_cast(float, (unsigned)ub);
      lt[5 * 128] = __builtin_bit_cast(float, (unsigned)(ub >> 32));
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

  // ---- iterations: both passes in both waves at once, one barrier per phase boundary ----------------------------
  if constexpr (ROLE == 0) {                       // the start iterate's contribution to the right wave's first pass
    publish_block(std::integral_constant<int, 0>{});
    if constexpr (NX == 2) publish_block(std::integral_constant<int, 1>{});
    double s = 0.0;
#pragma unroll
    for (int q = 0; q <= NBW - 1 - NX; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += w[q][j];
    publish_far_field(s);
  }
  wg_sync();
  if constexpr (!WITH_J) {
    for (int it = 0; it < a.n_iter; ++it) {
      const double beta = a.betas[it];
      forward();
      wg_sync();                                   // residual fragments and their far field are out
      if constexpr (LOOPS) { ldsq0 = ldsq1 = lwsq0 = lwsq1 = 0.0; }
      backward(beta);
      if constexpr (LOOPS) {                       // this wave's half of the rule's two norms (fista_mfma.h)
        double num = ldsq0 + ldsq1, den = lwsq0 + lwsq1;
        num += __shfl_xor(num, 16, 64);
        den += __shfl_xor(den, 16, 64);
        num += __shfl_xor(num, 32, 64);
        den += __shfl_xor(den, 32, 64);
        xl[(2 * ROLE) * 64] = num;
        xl[(2 * ROLE + 1) * 64] = den;
      }
      if (PB_MFMA_CHECKS && (((it & 7) == 7) || (it == a.n_iter - 1) || (it == 0 && !a.cold))) range_check();
      wg_sync();                                   // the updated iterate's fragment and far field are out
      if constexpr (LOOPS) {
        // both waves add the halves in the same order: the same verdict in both (the branches below are workgroup-uniform)
        const double num = xl[0 * 64] + xl[2 * 64], den = xl[1 * 64] + xl[3 * 64];
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
          xg[(ROLE * 2 + 0) * 64 + lane] = gq;
          xg[(ROLE * 2 + 1) * 64 + lane] = wq;
          wg_sync();
          const float go = xg[((1 - ROLE) * 2 + 0) * 64 + lane], wo = xg[((1 - ROLE) * 2 + 1) * 64 + lane];
          const bool in_range = gq < 60000.0f && go < 60000.0f;
          const float wm = fmaxf(wq, wo);
          const bool badq = !in_range || (a.rho_guard && wm > 0.0f && (float)th > MFMA_RHO_MAX * wm) || degenerate;
          if (fire) {
            lactive = false;
            if (live && !badq) {
              // (addresses made HERE, from laundered values: hoisted out of the loop they cost registers through the whole solve)
              double* wrow = a.w + (int64_t)p * a.ldw;
              int tbv = tb;
              asm volatile("" : "+v"(wrow), "+v"(tbv));
#pragma unroll
              for (int q = 0; q < NBW; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const int t = 32 * (QOFF + q) + tbv + j;
                  if (t < a.N) wrow[t] = w[q][j] * (double)inv_sigma;
                }
            }
            if (ROLE == 0 && live && a.n_done && g == 0) a.n_done[p] = badq ? -1 : it + 1;
          }
          if (__builtin_amdgcn_ballot_w64(lactive && live) == 0) return;       // every problem of the workgroup has finished (both waves agree)
          wg_sync();                                 // (the guard area is read again at the next finish)
        }
      }
    }
  } else {
    // rotated: the cost of iterate k+1 comes from the residual of the NEXT forward pass (one pass in front)
    forward();
    if constexpr (CERT) lt[6 * 128] = jw2;         // this lane's part of ||w_0||^2
    wg_sync();
    for (int it = 0; it < a.n_iter; ++it) {
      const double beta = a.betas[it];
      cert_it = it;
      backward(beta);
      wg_sync();
      forward();
      float sq = jsq, l1 = jl1;                     // this lane's samples -> this wave's half of the problem
      sq += __shfl_xor(sq, 16, 64);
      l1 += __shfl_xor(l1, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      l1 += __shfl_xor(l1, 32, 64);
      float t = 0.0f;
      if constexpr (CERT) {
        // this half's part of  sum v^2 - tol^2 (||w_k||^2 / p1 + 4 ||w_{k+1}||^2 / p2)
        t = cvsq - cert_t2 * ((1.0001f / CP1) * lt[6 * 128] + (4.0001f / CP2) * jw2);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        lt[6 * 128] = jw2;
      }
      xj[(ROLE * 3 + 0) * 64 + lane] = sq;
      xj[(ROLE * 3 + 1) * 64 + lane] = l1;
      if constexpr (CERT) xj[(ROLE * 3 + 2) * 64 + lane] = t;
      wg_sync();                                   // (the barrier of the forward pass: the halves of the cost meet here)
      sq += xj[((1 - ROLE) * 3 + 0) * 64 + lane];
      l1 += xj[((1 - ROLE) * 3 + 1) * 64 + lane];
      if (ROLE == 0 && live && g == 0 && (!CERT || (a.J != nullptr && !cflag))) a.J[(int64_t)p * a.ldj + it] = fmaf(jq, sq, jl * l1);
      if constexpr (CERT) {
        t += xj[((1 - ROLE) * 3 + 2) * 64 + lane];
        cflag = cflag | ((it >= 7) & !(t >= cert_lim));      // NaN-safe; the same verdict in both waves
      }
      if (PB_MFMA_CHECKS && (((it & 7) == 7) || (it == a.n_iter - 1) || (it == 0 && !a.cold))) range_check();
    }
  }

  // ---- guards over the whole series, store ---------------------------------------------------------------------
  guard = fmaxf(guard, __shfl_xor(guard, 16, 64));
  guard = fmaxf(guard, __shfl_xor(guard, 32, 64));
  wlast = fmaxf(wlast, __shfl_xor(wlast, 16, 64));
  wlast = fmaxf(wlast, __shfl_xor(wlast, 32, 64));
  xg[(ROLE * 2 + 0) * 64 + lane] = guard;
  xg[(ROLE * 2 + 1) * 64 + lane] = wlast;
  wg_sync();
  {
    const float go = xg[((1 - ROLE) * 2 + 0) * 64 + lane], wo = xg[((1 - ROLE) * 2 + 1) * 64 + lane];
    guard = (guard < 60000.0f && go < 60000.0f) ? fmaxf(guard, go) : 65504.0f;      // NaN on either side: out of range
    wlast = fmaxf(wlast, wo);
  }
  const bool bad = !(guard < 60000.0f) || degenerate || (a.rho_guard && wlast > 0.0f && (float)th > MFMA_RHO_MAX * wlast) || (CERT && cflag);
  if (live && !bad && (!LOOPS || lactive)) {
    double* wrow = a.w + (int64_t)p * a.ldw;
#pragma unroll
    for (int q = 0; q < NBW; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = 32 * (QOFF + q) + tb + j;
        if (t < a.N) wrow[t] = w[q][j] * (double)inv_sigma;
      }
  }
  if (ROLE == 0 && live && a.n_done && g == 0 && (!LOOPS || lactive)) a.n_done[p] = bad ? -1 : a.n_iter;
}

// one workgroup = two waves = 16 problems; the wave index picks the half (a scalar branch: each wave runs one role)
template <int NBA, int NBB, bool TAPS_DEV = false, bool WITH_J = false, bool CERT = false, bool LOOPS = false, int NT = 2>
__global__ __launch_bounds__(128) void fista_mfma2_kernel(FistaArgs a, MfmaTaps tp) {
  extern __shared__ __attribute__((aligned(16))) char mf2_smem[];
  if (a.range) {                                   // a candidate launch of a device-side plan: workgroups beyond its slots leave
    if ((int)blockIdx.x * 16 + a.range[0] >= a.range[1]) return;      // (both waves: before any barrier)
  }
  if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) mfma2_role<NBA, NBB, TAPS_DEV, 0, WITH_J, CERT, LOOPS, NT>(a, tp, mf2_smem);
  else mfma2_role<NBA, NBB, TAPS_DEV, 1, WITH_J, CERT, LOOPS, NT>(a, tp, mf2_smem);
}

// Plain solves, with or without the cost trace, the window rule (wind = 6) as a no-fire certificate, the _loops_deconv
// rule in full (no cost trace); HRFs of up to 33 taps, 32 (NBA+NBB-1) < N <= 32 (NBA+NBB).  The shared-HRF z-step (taps in device memory): plain only.
template <int NBA, int NBB>
int launch_mfma2(const FistaArgs& a, const double* taps, int K, bool with_j, hipStream_t st) {
  constexpr int NB = NBA + NBB;
  if (a.N > 32 * NB || a.N <= 32 * (NB - 1) || K < 1 || K > 65) return 1;
  const bool three = K > 33;                       // three near tiles: plain solves and the cost trace, four blocks at least per wave
  const bool cert = a.stop_mode == PB_STOP_WINDOW, loops = a.stop_mode == PB_STOP_LOOPS;
  if (!a.n_done) return 1;
  if ((with_j || cert || loops) && a.taps_pp) return 1;
  if (loops && with_j) return 1;                   // (the _loops_deconv rule: plain variant, as on the one-wave form)
  if (three && NBA <= 3) return 1;
  const int64_t groups = (launch_count(a) + 15) / 16;
  const dim3 grid((unsigned)groups), block(128);
  const size_t lds = mfma2_lds_bytes(NBB);
  if (a.taps_pp) {                                 // shared HRF and step in device memory (the blind step's z-step)
    const MfmaTaps none{};
    if constexpr (NBA > 3) {
      if (three) {
        hipLaunchKernelGGL((fista_mfma2_kernel<NBA, NBB, true, false, false, false, 3>), grid, block, lds, st, a, none);
        return 0;
      }
    }
    hipLaunchKernelGGL((fista_mfma2_kernel<NBA, NBB, true>), grid, block, lds, st, a, none);
    return 0;
  }
  const MfmaTaps tp = make_mfma_taps(taps, K);
  if constexpr (NBA > 3) {
    if (three) {
      if (loops) hipLaunchKernelGGL((fista_mfma2_kernel<NBA, NBB, false, false, false, true, 3>), grid, block, lds, st, a, tp);
      else if (cert) hipLaunchKernelGGL((fista_mfma2_kernel<NBA, NBB, false, true, true, false, 3>), grid, block, lds, st, a, tp);
      else if (with_j) hipLaunchKernelGGL((fista_mfma2_kernel<NBA, NBB, false, true, false, false, 3>), grid, block, lds, st, a, tp);
      else hipLaunchKernelGGL((fista_mfma2_kernel<NBA, NBB, false, false, false, false, 3>), grid, block, lds, st, a, tp);
      return 0;
    }
  }
  if (loops) hipLaunchKernelGGL((fista_mfma2_kernel<NBA, NBB, false, false, false, true>), grid, block, lds, st, a, tp);
  else if (cert) hipLaunchKernelGGL((fista_mfma2_kernel<NBA, NBB, false, true, true>), grid, block, lds, st, a, tp);
  else if (with_j) hipLaunchKernelGGL((fista_mfma2_kernel<NBA, NBB, false, true, false>), grid, block, lds, st, a, tp);
  else hipLaunchKernelGGL((fista_mfma2_kernel<NBA, NBB, false>), grid, block, lds, st, a, tp);
  return 0;
}

}  // namespace pb
