// Register-resident fused FISTA kernel with both operators on the MATRIX pipe (gfx950 MFMA).
//
// The two K-tap FIRs are two thirds of the instructions of the VALU kernels (fista_pair_ffa.h) and
// those kernels are VALU-issue bound.  v_pk_fma_f32 does not overlap with MFMA on gfx950, but
// unpacked vector instructions (float64 update, conversions) do (profiles/r3_mfma16_coissue.txt):
// a 16-bit matrix instruction holds the vector issue port for 8 of its 16 cycles only.  So here the
// WHOLE linear operator moves to v_mfma_f32_16x16x32_f16, scans included:
//
//   x = K_h (cumsum w) = T_c w,    T_c[t][s] = c[t - s],  c = cumsum(h)  (c[m] = S = sum h for m >= K-1)
//   g = cumsum^T K_h^T r = T_c^T r,   r = x - y
//
// (The reference's other form, g = T_c^T x - T_c^T y with the second term precomputed, was measured
// and is WORSE here: the 22-bit rounding of x then enters at the scale of y instead of the scale of
// the residual -- 3e-6 instead of 2.5e-7 on diff_z, tools/r3_mfma_precision.py.)
//
// T_c is lower triangular Toeplitz with a CONSTANT far field, so per block of samples
//   x_q = C0 w_q + C1 w_{q-1} + S 1 1^T (w_0 + ... + w_{q-2})
// two near tiles (three for 34..64 taps) and the far field.  Round 3 carried the far field as a running "carry"
// accumulated by one more tile product per block (3 of the 15 matrix instructions of a block and pass) and added to
// the accumulators' start values on the vector pipe.  Round 4 (this file) lets it ride INSIDE the near products: a
// block is 31 samples + one SUM SLOT (slot 31: lane group 3, j = 7).  As an OUTPUT row of tile 0 the slot adds up the
// block (entries 2^-9, and 1 at [31][31]): D_{q+1} = 2^-9 sum(w_q) + D_q, the running prefix sum, which lands in the
// very lane and register that hold the slot -- as an INPUT column of the last near tile (entries S 2^9) it feeds
// S sum_{b <= q-NT} w_b to the 31 real rows of block q.  So a block costs 12 matrix instructions per pass instead of 15
// (228 per iteration at 300 scans instead of 276), the accumulators start from -y itself (no carry to add), and the
// price is one select + one addition + one more v_fma_mix per block and pass (the slot's value enters the float16
// split together with the samples; what the 22-bit split drops of it is carried along and added back: the prefix sums
// keep float32 precision, as the round-3 carry did).  The adjoint mirrors it (suffix sums of the residual; the slot of
// the fetched fragment is patched with two v_perm_b32).  No cross-lane instruction is left in the loop: the cumulative
// sums ride in the matrix accumulators.  The products of a block run OLDEST tile first, so that the fragment of the
// block itself -- whose sum slot needs the finished accumulators of the block before -- is needed last.
//
// Precision: operands are split in two float16 parts (x = hi + lo, 22 bits; three products
// hi.hi + hi.lo + lo.hi, float32 accumulation in the matrix unit), the iterate and its update
// stay float64 as in the other kernels.  Emulated on the golden inputs (tools/emulate_f16_split.py,
// DESIGN 3): 2e-7 on diff_z after 500 iterations against 4e-8 for float32 operators; tolerance 1e-5.
// float16 range: every voxel is scaled by a power of two so that max|y| lies in [2^13, 2^14) (the
// problem is scale-covariant, threshold included: exact), taps by a power of two given by the host.
//
// Mapping: one wave = 16 problems; lane (v = lane & 15, g = lane >> 4) owns slots k = 8 g + j (j < 8) of every
// block q < NB of problem v, slot k < 31 holding sample t = 31 q + k -- exactly the B-operand layout of the 16x16x32
// instruction (k = 8 g + j), and, with the tile ROWS permuted (row 4 g + i of row-half r <-> slot
// 8 g + 4 r + i: the tile is data, any row order is free), exactly its D layout too:
// operands and results never change lanes.  One wave per SIMD (the iterate alone is 160 VGPRs):
// the matrix pipe and the vector pipe of the SAME wave overlap.
//
// Reference: pybold/bold_signal.py:62-72, pybold/linear.py:73-113, pybold/convolution.py:105-132.
#pragma once
#include "../../include/pybold_hip.h"
#include "common.h"
#include "fista_fast.h"
#ifndef PB_MFMA_CHECKS
#define PB_MFMA_CHECKS 1
#endif
// Development switches (A/B builds, tools/r3_mfma_ab.py).  PB_MFMA_SB: a scheduling barrier behind
// every PB_MFMA_SBK-th slot of the hand-pipelined loops.  Measured for 98 304 problems x 500
// iterations: barrier per slot 10.85 ms, per block 10.54 ms, none 10.47 ms (the source order is
// kept well enough by data dependences; the scheduler fills the wait states itself): none.
#ifndef PB_MFMA_SBK
#define PB_MFMA_SBK 15
#endif
// (Round 3's PB_MFMA_YMAT experiment -- -y through the matrix pipe with an identity tile: 145 vector instructions fewer,
// 40 matrix instructions more, 2.5 % slower, profiles/r3_mfma_y_through_matrix_pipe_ab.txt -- left with the carry tile.)
#ifndef PB_MFMA_SB
#define PB_MFMA_SB
#endif
// PB_MFMA_REM (on): what the 22-bit split of a prefix / suffix sum drops is carried along and added back to the next sum
// (one v_fma_mix_f32 + one addition per block and pass), so that the sums keep float32 precision.  A/B build: 0.
#ifndef PB_MFMA_REM
#define PB_MFMA_REM 1
#endif
// PB_MFMA_FETCH_SLOT: the residual fragment of block q-1 is fetched from LDS in this slot of block q (-1: at the start
// of block q-1 itself, three slots before its sum slot is rebuilt -- the LDS latency then shows: 9.98 ms against 9.88 with
// slot 8 for 98 304 problems x 500 iterations, three alternations on one box).
#ifndef PB_MFMA_FETCH_SLOT
#define PB_MFMA_FETCH_SLOT 8
#endif
#ifndef PB_MFMA_CHECKS
#define PB_MFMA_CHECKS 1
#endif

namespace pb {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u3 __attribute__((ext_vector_type(3)));

constexpr float MFMA_RHO_MAX = 0.02f;
constexpr int MFMA_SPAN = 31;       // samples per block of 32 slots; slot 31 (lane group 3, j = 7) is the block's sum slot
constexpr int MFMA_SSHIFT = 9;      // the sums ride scaled by 2^-9: below the largest sample for up to 512 of them

struct MfmaTaps {
  float c[96];      // 2^a * cumsum(h)[m], m < 96 (constant from m = K-1 on; K <= 33 uses 64 of them)
  double g_scale;   // 2^(-2a): the gradient comes out scaled by 2^(2a)
  float y_scale;    // 2^a
};

inline MfmaTaps make_mfma_taps(const double* taps, int K) {
  MfmaTaps t;
  double c[96], run = 0.0, cmax = 0.0;
  for (int m = 0; m < 96; ++m) {
    if (m < K) run += (double)(float)taps[m];
    c[m] = run;
    cmax = fabs(run) > cmax ? fabs(run) : cmax;
  }
  int e = 0;
  if (cmax > 0.0) frexp(cmax, &e);          // cmax = f 2^e, f in [0.5, 1)
  const int a = 3 - e;                       // max |c| 2^a in [4, 8)
  for (int m = 0; m < 96; ++m) t.c[m] = (float)ldexp(c[m], a);
  t.g_scale = ldexp(1.0, -2 * a);
  t.y_scale = (float)ldexp(1.0, a);
  return t;
}

struct Frag {
  h8 hi, lo;
};

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
// two float32 -> packed float16, round to nearest even (v_cvt_pk_f16_f32).  The LOW parts are
// rounded, not truncated: a truncated split shrinks every operand by ~2^-23 on average, a bias
// that adds up coherently over samples and iterations (measured: 14x the error of float32
// operators along a regularisation path, tools/r3_mfma_precision.py).
__device__ __forceinline__ unsigned pk_rne(float x0, float x1) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f2v{x0, x1}, h2));
}

// (x0, x1) -> packed hi = RTZ(x) and packed lo = RNE(x - hi); x - hi is exact in float32.
// (v_fma_mixlo_f16 + v_fma_mixhi_f16 would write the rounded differences straight into the two
// halves -- one instruction less per pair, 96 fewer per iteration -- and measured 4.5 % SLOWER:
// profiles/r3_mfma_split_mixlo_ab.txt; the partial-register writes serialise.)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
  float l0, l1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(x1));
  lo = pk_rne(l0, l1);
}

// eight float32 -> float16 hi / lo parts (hi = RTZ(x), lo = RNE(x - hi): 22 bits, unbiased)
__device__ __forceinline__ Frag split8(const float (&x)[8]) {
  u4 ph, pl;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned h2, l2;
    split_pair(x[2 * p], x[2 * p + 1], h2, l2);
    ph[p] = h2;
    pl[p] = l2;
  }
  return Frag{__builtin_bit_cast(h8, ph), __builtin_bit_cast(h8, pl)};
}

// acc += (Ahi + Alo) (Bhi + Blo) without the lo.lo term
__device__ __forceinline__ f4 mma3(const Frag& A, const Frag& B, f4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.hi, B.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.hi, B.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.lo, B.hi, acc, 0, 0, 0);
  return acc;
}

// NB blocks of 31 samples + one sum slot per series, 31 (NB - 1) < N <= 31 NB (only the last block can hold
// padding), HRFs of up to 33 taps (NT = 2 near tiles) or 64 taps (NT = 3: one more near tile per block and
// pass, the far field starts one block further away; every variant but CERT).  No stop rule.
// WITH_J: cost trace, J[it] = 0.5 ||T_c w_{it+1} - y||^2 + lbda ||w_{it+1}||_1 (pybold/bold_signal.py:74-77)
//   from the residual of the NEXT forward pass (the loop is rotated: one forward pass in front).
// TAPS_DEV: the HRF and the step are read from device memory (a.taps_pp: K float64 shared by every
//   problem, a.step_vec[0]): the shared-HRF blind step, whose taps never pass through the host.
// Scales: taps 2^a (max |c| in [4, 8)), series 2^a sigma with max |2^a sigma y| in [2^13, 2^14), so
// that the residual fragments 2^a sigma (x - y) -- what float16 has to hold -- have a margin of
// 2-4x the largest sample.  A problem whose residual hi parts reached 2^15 or whose |sigma w|
// reached 60000 (checked every 8 iterations and at the end) is left untouched with n_done = -1
// for the exact kernels (capi.hip re-solves it); between checks the conversions saturate.
// CERT: the deconv window rule (wind = 6) as the per-iteration NO-FIRE CERTIFICATE of
//   fista_pair_ffa.h: numerator bounded from below by ONE tracked sample per lane (four per problem,
//   in four different blocks), denominator from above by ||w_k|| + 2 ||w_{k+1}|| + 4 th sqrt(N);
//   a problem that cannot be cleared is handed back (n_done = -1) for the exact rule.  Implies WITH_J.
#ifndef PB_MFMA_KATTR
#define PB_MFMA_KATTR
#endif
// LOOPS: the _loops_deconv stop rule (pybold/bold_signal.py:267-273), evaluated in full (not certified), in float64,
//   on the iterate of the 22-bit operators: ||w_{k+1} - u_k|| / (||w_{k+1}|| + 1e-10) < tol from the fourth iteration on
//   (a criterion within ~1e-6 of tol may cross one iteration apart from the float64 reference).  With w_{k+1} = u_k - (1 + beta) d, d = clamp(u_k, +-th), the numerator is
//   (1 + beta) ||d||: two float64 multiply-adds per sample next to the update.  A problem that meets the rule is
//   written out at that moment (after a range / accuracy check of its own) and its lanes keep iterating, results
//   discarded -- as in fista_fast.h.  Plain variant only (no cost trace, taps as kernel arguments, two near tiles).
template <int NB, bool WITH_J = false, bool TAPS_DEV = false, bool CERT = false, int NT = 2, bool LOOPS = false>
__global__ __launch_bounds__(256) PB_MFMA_KATTR void fista_mfma_kernel(FistaArgs a, MfmaTaps tp) {
  static_assert(NT == 2 || NT == 3, "two near tiles (K <= 33) or three (K <= 64)");
  static_assert(!LOOPS || (!WITH_J && !TAPS_DEV && !CERT && NT == 2), "the stop rule rides the plain variant");
  constexpr int LCW = NT == 2 ? 64 : 96;           // cumulative taps kept per wave: lags 0 .. 32 NT - 1
  static_assert(!CERT || (WITH_J && !TAPS_DEV), "the certificate runs in the rotated (cost trace) loop");
  const int lane = threadIdx.x & 63;
  const int v = lane & 15, g = lane >> 4;
  const int wave = (int)((blockIdx.x * 256 + threadIdx.x) >> 6);
  int s0, n_list;                                 // this launch's slots [s0, n_list) of its list (the batch itself without a partition)
  launch_slots(a, s0, n_list);
  // the whole wave lies beyond the list: leave (no workgroup barrier anywhere below; a SCALAR branch -- the wave index is
  // the same in every lane -- so that the body does not run under a saved exec mask)
  if (__builtin_amdgcn_readfirstlane(wave) * 16 + s0 >= n_list) return;
  bool live;
  const int p = slot_to_problem(a, wave * 16 + v + s0, n_list, live);
  const int tb = 8 * g;                          // this lane's first slot inside a block of 32 slots
  const bool g3 = g == 3;                        // lane group 3 holds the sum slot (slot 31 = its j = 7)

  extern __shared__ __attribute__((aligned(16))) char mf_smem[];
  u4* lrf = reinterpret_cast<u4*>(mf_smem) + ((threadIdx.x >> 6) * NB * 2 * 64 + lane);
  float* lc = reinterpret_cast<float*>(mf_smem + (size_t)4 * NB * 2 * 64 * sizeof(u4)) + (threadIdx.x >> 6) * LCW;
  // CERT: per-lane words behind the taps (slot-major: conflict-free): [0..3] ring of the tracked
  // sample's last four increments, [4,5] its u_{k-1} (float64 halves), [6] this lane's ||w_k||^2 part
  float* lt = reinterpret_cast<float*>(mf_smem + (size_t)4 * NB * 2 * 64 * sizeof(u4)) + 4 * LCW + threadIdx.x;
  if constexpr (CERT) {
#pragma unroll
    for (int q = 0; q < 7; ++q) lt[q * 256] = 0.0f;
  }
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  // ---- cumulative taps c[m] = sum_{k <= min(m, K-1)} h[k], scaled by 2^a so that max |c| is in [4, 8) ----
  double step = a.step, g_scale = tp.g_scale;
  float y_scale = tp.y_scale;
  if constexpr (TAPS_DEV) {
    double run = 0.0, run2 = 0.0;
    for (int k = 0; k <= lane && k < a.K; ++k) run += (double)(float)a.taps_pp[k];
    float cm = fabsf((float)run);
    if constexpr (NT == 3) {                       // lags 64 .. 95 (K <= 64: sum of the first lane + 65 taps)
      run2 = run;
      for (int k = lane + 1; k <= lane + 64 && k < a.K; ++k) run2 += (double)(float)a.taps_pp[k];
      cm = fmaxf(cm, fabsf((float)run2));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) cm = fmaxf(cm, __shfl_xor(cm, o, 64));
    int e = 0;
    if (cm > 0.0f) (void)frexpf(cm, &e);
    const int sa = 3 - e;
    lc[lane] = (float)ldexp(run, sa);
    if constexpr (NT == 3) { if (lane < 32) lc[64 + lane] = (float)ldexp(run2, sa); }
    g_scale = ldexp(1.0, -2 * sa);
    y_scale = ldexpf(1.0f, sa);
    step = a.step_vec[0];
  } else {
    lc[lane] = tp.c[lane];
    if constexpr (NT == 3) lc[64 + (lane & 31)] = tp.c[64 + (lane & 31)];      // (every lane: no exec-masked region here, fista_mfma4.h says why)
  }
  wave_sync();

  // ---- operator tiles (A operands): lane holds row rho = lane & 15, k = 8 (lane >> 4) + j ----
  // Output slot ko = 8 gp + 4 r + i of the block, input slot ki = 8 kg + j of the block o blocks away.  Slot 31 is the
  // sum slot: as an output row (tile 0 only) it adds up the block's samples, scaled by 2^-9, plus what the slot held
  // ([31][31] = 1); as an input column (last tile only) it carries 2^9 S to the 31 real rows.
  Frag An[2][NT], Bn[2][NT];                     // forward near [r][o], adjoint near [r][o]
  {
    const int rho = lane & 15, kg = lane >> 4, gp = rho >> 2, i = rho & 3;
    auto cval = [&](int lag) -> float { return lag < 0 ? 0.0f : lc[lag > LCW - 1 ? LCW - 1 : lag]; };
    const float sfar = lc[LCW - 1] * (float)(1 << MFMA_SSHIFT), eps = 1.0f / (float)(1 << MFMA_SSHIFT);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int o = 0; o < NT; ++o) {
        float fa[8], fb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int ko = 8 * gp + 4 * r + i, ki = 8 * kg + j;
          const float vs = ko == 31 ? (o == 0 ? (ki == 31 ? 1.0f : eps) : 0.0f) : (o == NT - 1 ? sfar : 0.0f);
          const bool special = ko == 31 || ki == 31;
          fa[j] = special ? vs : cval(MFMA_SPAN * o + ko - ki);
          fb[j] = special ? vs : cval(MFMA_SPAN * o + ki - ko);
        }
        An[r][o] = split8(fa);
        Bn[r][o] = split8(fb);
      }
  }

  // ---- load the problem, scale it into the float16 range ---------------------------------
  const double lb = a.lbda_vec ? a.lbda_vec[p] : a.lbda;
  float ysn[NB][8];                              // -2^a sigma y
  double w[NB][8];                               // sigma w
  float sigma = 1.0f, inv_sigma = 1.0f;
  bool degenerate = false;                       // an all-zero (or non-finite) series with a warm start: no scale to work at
  {
    const float* yrow = a.y + (int64_t)(p / a.y_rep) * a.ldy;
    float m = 0.0f;
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // Only the LAST block can hold padding (31 (NB-1) < N): every other load is unconditional, the last block's
        // are branch-free (clamped address + select), and so is the sum slot (j = 7 of lane group 3: no sample).
        // A per-lane `if` around each load put 80 exec-masked regions
        // into this prologue, and the compiler was seen to place a VGPR -> AGPR spill INSIDE such a region
        // (fista_mfma2.h, round 4: with the mask empty -- a cold start -- the spill never happened and its reload,
        // an LDS address, was garbage); tools/isa_spill_lint.py looks for that pattern in the listings.
        const int t = MFMA_SPAN * q + tb + j;
        float yv;
        if (q == NB - 1) {
          const bool ok = t < a.N && !(j == 7 && g3);
          float yl = yrow[t < a.N ? t : a.N - 1];
          asm volatile("" : "+v"(yl));           // (the load stays unconditional)
          yv = ok ? yl : 0.0f;
        } else if (j == 7) {
          float yl = yrow[t];                    // (lane group 3: the first sample of block q+1, a valid address)
          asm volatile("" : "+v"(yl));
          yv = g3 ? 0.0f : yl;
        } else {
          yv = yrow[t];
        }
        ysn[q][j] = yv;
        m = fmaxf(m, fabsf(yv));
      }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    {                                            // (branch-free too)
      const bool okm = m > 0.0f && m < 3.0e38f;
      int e = 0;
      (void)frexpf(okm ? m : 1.0f, &e);          // m = f 2^e, f in [0.5, 1)
      // max |2^a sigma y| in [2^(ybits-1), 2^ybits): the residual fragments 2^a sigma (x - y) are
      // what float16 has to hold (hi part below 2^15), so the tap scale 2^a belongs in sigma
      const float sg = ldexpf(1.0f, a.ybits - e) / y_scale, isg = ldexpf(1.0f, e - a.ybits) * y_scale;
      sigma = okm ? sg : 1.0f;
      inv_sigma = okm ? isg : 1.0f;
      // y = 0 gives no scale: a warm start is then iterated as it comes, and a small one falls into the float16
      // subnormals (its prefix sums first: they ride scaled by 2^-9).  Such a problem is handed back (n_done = -1) like
      // one that left the float16 range; from a cold start the iterate stays exactly 0 and nothing is lost.
      degenerate = !okm && !a.cold;
    }
    const float ys = -sigma * y_scale;
    const double* wrow = a.w + (int64_t)p * a.ldw;
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ysn[q][j] *= ys;
        w[q][j] = 0.0;
      }
    if (!a.cold) {                                // (the same for every lane: a scalar branch)
#pragma unroll
      for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int t = MFMA_SPAN * q + tb + j;
          if (q == NB - 1) {
            double wl = wrow[t < a.N ? t : a.N - 1] * (double)sigma;
            asm volatile("" : "+v"(wl));         // (the load stays unconditional: no exec-masked region, see above)
            w[q][j] = (t < a.N && !(j == 7 && g3)) ? wl : 0.0;
          } else if (j == 7) {
            double wl = wrow[t] * (double)sigma;
            asm volatile("" : "+v"(wl));
            w[q][j] = g3 ? 0.0 : wl;             // the sum slot's own iterate stays 0 (its step is 0, see `update`)
          } else {
            w[q][j] = wrow[t] * (double)sigma;
          }
        }
    }
  }
  const double th = lb * step * (double)sigma;
  const double nstep = -step * g_scale;
  const double nstep7 = g3 ? 0.0 : nstep;        // j = 7: the sum slot of lane group 3 takes no step (its "gradient" is a sum)
  float guard = 0.0f;                            // largest |operand| seen by the range checks
  float wlast = 0.0f;                            // largest |sigma w| of this lane at the last check
  // cost trace: 0.5 ||r''||^2 / (2^a sigma)^2 + lbda ||w'||_1 / sigma
  const float jq = 0.5f * (inv_sigma / y_scale) * (inv_sigma / y_scale), jl = (float)lb * inv_sigma;
  float jsq = 0.0f, jl1 = 0.0f;
  // CERT: tracked sample = sample 3 of block CQ[g] (one block per lane group, spread over the series)
  constexpr int CQ0 = NB / 8, CQ1 = (3 * NB) / 8, CQ2 = (5 * NB) / 8, CQ3 = (7 * NB) / 8;
  static_assert(!CERT || (CQ0 < CQ1 && CQ1 < CQ2 && CQ2 < CQ3), "four distinct blocks");
  double ldsq0 = 0.0, ldsq1 = 0.0, lwsq0 = 0.0, lwsq1 = 0.0;     // LOOPS: this lane's parts of ||d||^2 and ||w_{k+1}||^2
  bool lactive = true;                           // LOOPS: this problem has not met its rule yet
  double cu = 0.0, cw = 0.0;                     // u_k and w_{k+1} of the tracked sample
  float jw2 = 0.0f, cvsq = 0.0f;                 // this lane's ||w||^2 part, its v^2
  bool cflag = false;
  int cert_it = -1;
  constexpr float CP1 = 0.3133f, CP2 = 0.6467f, CP3 = 0.04f;
  const float cert_t2 = ((float)a.tol * 1.001f) * ((float)a.tol * 1.001f);
  const float cert_c0 = (float)th * (4.0f * 1.0001f) * __builtin_sqrtf(32.0f * NB) + 3.1e-10f * sigma;
  const float cert_lim = cert_t2 * cert_c0 * cert_c0 * (1.0001f / CP3);

  // Register placement: the operator tiles are read by matrix instructions only and -y'' once per
  // iteration: they live in the accumulator half of the register file (the asm constraints put
  // them there; MFMA reads A/B operands from either half), the float64 iterate and everything
  // the vector pipe touches in the other.
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int o = 0; o < NT; ++o)
      asm volatile("" : "+a"(An[r][o].hi), "+a"(An[r][o].lo), "+a"(Bn[r][o].hi), "+a"(Bn[r][o].lo));
#pragma unroll
  for (int q = 0; q < NB; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(ysn[q][j]));

  // The iteration is straight-line.  The matrix pipe takes one instruction per 16 cycles and holds
  // the vector issue port for 8 of them, so the source is software-pipelined BY HAND at that grain
  // (data dependences keep the order; PB_MFMA_SB can pin it): every matrix instruction of block q is
  // followed by a slice of the vector work of its neighbours -- the float16 fragments of block
  // q+1, the residual (or the update) of the block before.
  auto mfma_part = [](const Frag& A, const Frag& B, f4 acc, int part) __attribute__((always_inline)) -> f4 {
    // the three products of a split pair, one per call: hi.hi, hi.lo, lo.hi
#ifdef PB_MFMA_LOLO
    if (part == 2) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A.lo, B.lo, acc, 0, 0, 0);
#endif
    return part == 0   ? __builtin_amdgcn_mfma_f32_16x16x32_f16(A.hi, B.hi, acc, 0, 0, 0)
           : part == 1 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(A.hi, B.lo, acc, 0, 0, 0)
                       : __builtin_amdgcn_mfma_f32_16x16x32_f16(A.lo, B.hi, acc, 0, 0, 0);
  };
  Frag rlast;                                    // residual fragment of the last block, forward pass -> adjoint pass
  // two float32 -> float16 hi / lo as split_pair, and what the split drops of x1: rem = x1 - hi - lo (exact)
  auto split_pair_rem = [](float x0, float x1, unsigned& hi, unsigned& lo, float& rem) __attribute__((always_inline)) {
    hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(x1));
    lo = pk_rne(l0, l1);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rem) : "v"(lo), "v"(l1));
  };
  // slot order of the 6 NT products of a block: row half r = 1 - (slot & 1) (the half that holds the sum row finishes
  // one slot before the end), tile o = NT-1 ... 0 (the block's own fragment last), the three split products in turn
  // ---- forward: r = T_c w - y, block by block (ascending); fragments of r go to LDS -------------
  auto forward = [&]() __attribute__((always_inline)) {
    Frag wf[NB + 1];
    f4 acc[NB + 1][2];
    unsigned ph[NB + 1][4], pl[NB + 1][4];
    unsigned rh[NB][4], rl[NB][4];
    float remf = 0.0f;                            // what the 22-bit split dropped of the last prefix sum
    if constexpr (WITH_J) { jsq = 0.0f; jl1 = 0.0f; }
    if constexpr (CERT) jw2 = 0.0f;
    auto prep_pair = [&](auto qc, auto pc) {      // slots 2p, 2p+1 of block q -> float16 hi / lo
      constexpr int q = decltype(qc)::value, pp = decltype(pc)::value;
      float x0, x1;                               // (asm: the conversion stays HERE, not behind the update)
      asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x0) : "v"(w[q][2 * pp]));
      asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(x1) : "v"(w[q][2 * pp + 1]));
      if constexpr (WITH_J) jl1 += fabsf(x0) + fabsf(x1);       // (the sum slot's own iterate is 0)
      if constexpr (CERT) jw2 = fmaf(x1, x1, fmaf(x0, x0, jw2));
      if constexpr (pp == 3 && q >= 1) {          // the sum slot: D_q = 2^-9 (w_0 + ... + w_{q-1}), finished by block q-1
        const float d = acc[q - 1][1][3] + remf;
        x1 = g3 ? d : x1;
      }
      if constexpr (PB_MFMA_REM && pp == 3 && q + 1 < NB) split_pair_rem(x0, x1, ph[q][pp], pl[q][pp], remf);
      else split_pair(x0, x1, ph[q][pp], pl[q][pp]);
      if constexpr (pp == 3) {
        wf[q].hi = __builtin_bit_cast(h8, u4{ph[q][0], ph[q][1], ph[q][2], ph[q][3]});
        wf[q].lo = __builtin_bit_cast(h8, u4{pl[q][0], pl[q][1], pl[q][2], pl[q][3]});
      }
    };
    auto cinit = [&](auto qc, auto rc) {          // accumulators of block q start from -y (the sum row: from 0)
      constexpr int q = decltype(qc)::value, r = decltype(rc)::value;
      acc[q][r] = f4{ysn[q][4 * r + 0], ysn[q][4 * r + 1], ysn[q][4 * r + 2], ysn[q][4 * r + 3]};
    };
    auto finish_pair = [&](auto qc, auto pc) {    // residual slots 2p, 2p+1 of block q -> fragment
      constexpr int q = decltype(qc)::value, pp = decltype(pc)::value;
      float x0 = acc[q][pp >> 1][(2 * pp) & 3], x1 = acc[q][pp >> 1][(2 * pp + 1) & 3];
      if constexpr (q == NB - 1) {                // padding behind sample N-1 and the sum slot (last block only)
        x0 = (MFMA_SPAN * q + tb + 2 * pp < a.N) ? x0 : 0.0f;
        x1 = (MFMA_SPAN * q + tb + 2 * pp + 1 < a.N && !(pp == 3 && g3)) ? x1 : 0.0f;
      }
      // (blocks before the last: the sum slot of the stored fragment holds the prefix sum; the adjoint pass patches it)
      if constexpr (WITH_J) {
        if constexpr (pp == 3 && q < NB - 1) jsq = fmaf(g3 ? 0.0f : x1, x1, fmaf(x0, x0, jsq));
        else jsq = fmaf(x1, x1, fmaf(x0, x0, jsq));
      }
      split_pair(x0, x1, rh[q][pp], rl[q][pp]);
      if constexpr (pp == 3) {
        // the residual fragments wait in LDS for the adjoint pass (each lane reads back only what
        // it wrote: no barrier); in registers they would cost 80 accumulator-file copies per iteration
        lrf[(2 * q) * 64] = u4{rh[q][0], rh[q][1], rh[q][2], rh[q][3]};
        lrf[(2 * q + 1) * 64] = u4{rl[q][0], rl[q][1], rl[q][2], rl[q][3]};
        if constexpr (q == NB - 1) {                // the adjoint pass starts with this block: it takes the fragment as it is
          rlast.hi = __builtin_bit_cast(h8, u4{rh[q][0], rh[q][1], rh[q][2], rh[q][3]});      // (no LDS round trip at the turn of the passes)
          rlast.lo = __builtin_bit_cast(h8, u4{rl[q][0], rl[q][1], rl[q][2], rl[q][3]});
        }
      }
    };
    static_for<0, 4>([&](auto pc) { prep_pair(std::integral_constant<int, 0>{}, pc); });
    cinit(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    cinit(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    static_for<0, NB>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      static_for<0, 6 * NT>([&](auto sc) {
        constexpr int sl = decltype(sc)::value;
        // -- the matrix instruction of this slot
        constexpr int r = 1 - (sl & 1), k = sl >> 1, o = NT - 1 - k / 3;     // near tile o: block q-o
        if constexpr (q >= o) acc[q][r] = mfma_part(An[r][o], wf[q >= o ? q - o : 0], acc[q][r], k % 3);
        // -- a slice of the neighbours' vector work.  The first two slots get work that depends on nothing just finished
        // (the next block's fragment); what reads the finished block q-1 comes two slots in -- the sum slot first: the
        // block's own fragment is needed from slot 6 on
        if constexpr (sl < 2) {
          if constexpr (q + 1 < NB) prep_pair(std::integral_constant<int, q + 1>{}, sc);
        } else if constexpr (sl == 2) {
          if constexpr (q >= 1) prep_pair(qc, std::integral_constant<int, 3>{});
        } else if constexpr (sl >= 3 && sl < 7) {
          if constexpr (q >= 1) finish_pair(std::integral_constant<int, q - 1>{}, std::integral_constant<int, sl - 3>{});
        } else if constexpr (sl == 7) {
          if constexpr (q + 1 < NB) prep_pair(std::integral_constant<int, q + 1>{}, std::integral_constant<int, 2>{});
        } else if constexpr (sl == 8 || sl == 9) {
          if constexpr (q + 1 < NB) cinit(std::integral_constant<int, q + 1>{}, std::integral_constant<int, sl - 8>{});
        }
        if constexpr ((sl % PB_MFMA_SBK) == PB_MFMA_SBK - 1) PB_MFMA_SB;
      });
    });
    static_for<0, 4>([&](auto pc) { finish_pair(std::integral_constant<int, NB - 1>{}, pc); });
  };
  // ---- adjoint and update: g = T_c^T r, block by block (descending) ------------------------------
  auto backward = [&](const double beta) __attribute__((always_inline)) {
    const double nb1 = -(1.0 + beta);
    f4 acc[NB + 1][2];
    Frag rf[NB + 2];
    u3 fh[NB], fl[NB];                           // fetched fragments of the blocks before the last: words 0-2 ...
    unsigned fh3[NB], fl3[NB];                    // ... and word 3
    float remb = 0.0f;                            // what the 22-bit split dropped of the last suffix sum
    // v_perm_b32 selector: lane group 3 takes the upper half-word (slot 7) from the new value, every other lane keeps its own
    const unsigned psel = g3 ? 0x07060100u : 0x03020100u;
    auto fetch = [&](auto qc) {                   // residual fragment of block q: LDS -> registers
      constexpr int q = decltype(qc)::value;
      if constexpr (q == NB - 1) {                // (the last block's sum slot was zeroed by the forward pass)
        rf[q].hi = __builtin_bit_cast(h8, lrf[(2 * q) * 64]);
        rf[q].lo = __builtin_bit_cast(h8, lrf[(2 * q + 1) * 64]);
      } else {
        // words 0-2 and word 3 apart: word 3 (slots 6, 7) is rebuilt by `patch`, and a fragment loaded in one piece
        // was seen to be copied register by register to make room for the new word
        const unsigned* ph3 = reinterpret_cast<const unsigned*>(&lrf[(2 * q) * 64]);
        const unsigned* pl3 = reinterpret_cast<const unsigned*>(&lrf[(2 * q + 1) * 64]);
        fh[q] = *reinterpret_cast<const u3*>(ph3);
        fl[q] = *reinterpret_cast<const u3*>(pl3);
        fh3[q] = ph3[3];
        fl3[q] = pl3[3];
      }
    };
    auto patch = [&](auto qc) {                   // the sum slot of block q: 2^-9 (r_{q+1} + ... + r_{NB-1}), finished by block q+1
      constexpr int q = decltype(qc)::value;
      const float d = acc[q + 1][1][3] + remb;
      const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(d, d));
      float l;
      asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h2), "v"(d));
      const unsigned l2 = pk_rne(l, l);
      if constexpr (PB_MFMA_REM && q >= 1) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(remb) : "v"(l2), "v"(l));
      rf[q].hi = __builtin_bit_cast(h8, u4{fh[q][0], fh[q][1], fh[q][2], __builtin_amdgcn_perm(h2, fh3[q], psel)});
      rf[q].lo = __builtin_bit_cast(h8, u4{fl[q][0], fl[q][1], fl[q][2], __builtin_amdgcn_perm(l2, fl3[q], psel)});
    };
    auto update = [&](auto qc, auto jc) {
      constexpr int q = decltype(qc)::value, j = decltype(jc)::value;
      const double gj = (double)acc[q][j >> 2][j & 3];
      const double u = fma(j == 7 ? nstep7 : nstep, gj, w[q][j]);
      const double d = fmin(fmax(u, -th), th);
      w[q][j] = fma(nb1, d, u);
      if constexpr (LOOPS) {
        if constexpr ((j & 1) == 0) { ldsq0 = fma(d, d, ldsq0); lwsq0 = fma(w[q][j], w[q][j], lwsq0); }
        else { ldsq1 = fma(d, d, ldsq1); lwsq1 = fma(w[q][j], w[q][j], lwsq1); }
      }
      if constexpr (CERT && j == 3 && (q == CQ0 || q == CQ1 || q == CQ2 || q == CQ3)) {
        constexpr int gq = q == CQ0 ? 0 : (q == CQ1 ? 1 : (q == CQ2 ? 2 : 3));
        cu = (g == gq) ? u : cu;
        cw = (g == gq) ? w[q][j] : cw;
      }
    };
    rf[NB - 1] = rlast;                             // (the last block's fragment never left the registers)
    static_for<0, NB>([&](auto qq) {
      constexpr int q = NB - 1 - decltype(qq)::value;
      constexpr int omax = (NT - 1 < NB - 1 - q) ? NT - 1 : NB - 1 - q;     // the oldest tile this block has
      static_for<0, 6 * NT>([&](auto sc) {
        constexpr int sl = decltype(sc)::value;
        constexpr int r = 1 - (sl & 1), k = sl >> 1, o = NT - 1 - k / 3;     // near tile o: block q+o
        if constexpr (o <= omax) {
          if constexpr (o == omax && (k % 3) == 0) acc[q][r] = mfma_part(Bn[r][o], rf[q + o], f4{0.f, 0.f, 0.f, 0.f}, 0);
          else acc[q][r] = mfma_part(Bn[r][o], rf[q + o], acc[q][r], k % 3);
        }
        // vector slices: the update of block q+1 needs its finished accumulators, so its last three samples wait for the
        // first slots of the NEXT block -- where nothing else is ready yet (the iterate is not needed before the next pass)
#if PB_MFMA_FETCH_SLOT >= 0
        if constexpr (sl == PB_MFMA_FETCH_SLOT) {   // the NEXT block's fragment: its sum slot is rebuilt at that block's slot 3
          if constexpr (q >= 1) fetch(std::integral_constant<int, q - 1>{});
        }
#else
        if constexpr (sl == 0) {
          if constexpr (q + 1 < NB) fetch(std::integral_constant<int, q>{});   // (needed from slot 6 on: the older tile runs first)
        }
#endif
        if constexpr (sl < 3) {
          if constexpr (q + 2 < NB) update(std::integral_constant<int, q + 2>{}, std::integral_constant<int, sl + 5>{});
        } else if constexpr (sl == 3) {
          if constexpr (q + 1 < NB) patch(std::integral_constant<int, q>{});
        } else if constexpr (sl >= 4 && sl < 9) {
          if constexpr (q + 1 < NB) update(std::integral_constant<int, q + 1>{}, std::integral_constant<int, sl - 4>{});
        } else if constexpr (sl >= 9 && sl < 12) {   // (the last block of the pass has no block behind it to wait for)
          if constexpr (q == 0 && NB >= 2) update(std::integral_constant<int, 1>{}, std::integral_constant<int, sl - 4>{});
        }
        if constexpr ((sl % PB_MFMA_SBK) == PB_MFMA_SBK - 1) PB_MFMA_SB;
      });
    });
    static_for<0, 8>([&](auto jc) { update(std::integral_constant<int, 0>{}, jc); });
    if constexpr (CERT) {
      // the window combination on this lane's tracked sample (see fista_pair_ffa.h): float32 from
      // float64 differences; its rounding, and that of the stored increments, is below 2^-21 M
      const float d1 = lt[((cert_it + 3) & 3) * 256], d2 = lt[((cert_it + 2) & 3) * 256], d3 = lt[((cert_it + 1) & 3) * 256];
      const unsigned ulo = __builtin_bit_cast(unsigned, lt[4 * 256]), uhi = __builtin_bit_cast(unsigned, lt[5 * 256]);
      const double up = __builtin_bit_cast(double, ((unsigned long long)uhi << 32) | ulo);
      const float dk = (float)(cu - up), e = (float)(cw - cu);
      const float v = fmaf(2.0f, d2, fmaf(3.0f, d1, fmaf(2.0f, dk, e))) + d3;
      const float m = fmaf(2.0f, fabsf(d2), fmaf(3.0f, fabsf(d1), fmaf(2.0f, fabsf(dk), fabsf(e)))) + fabsf(d3);
      const float vs = fmaxf(fmaf(-0x1p-21f, m, fabsf(v)), 0.0f);
      cvsq = vs * vs;
      lt[(cert_it & 3) * 256] = dk;
      const unsigned long long ub = __builtin_bit_cast(unsigned long long, cu);
      lt[4 * 256] = __builtin_bit_cast(float, (unsigned)ub);
      lt[5 * 256] = __builtin_bit_cast(float, (unsigned)(ub >> 32));
    }
  };
  // Range guard, every 8th iteration and after the last one: the largest |sigma w| (registers) and
  // the largest exponent among the hi halves of the residual fragments (LDS copy) -- a small block
  // of its own, so that the iteration body exists once.
  auto range_check = [&]() {
    unsigned mb = 0;                              // largest |sigma w| by its float32 bits: NaN and inf rank highest
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) mb = max(mb, __builtin_bit_cast(unsigned, (float)w[q][j]) & 0x7fffffffu);
    const float m = mb >= 0x7f800000u ? 65504.0f : __builtin_bit_cast(float, mb);
    wlast = m;
    unsigned e = 0;
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const u4 h = lrf[(2 * q) * 64];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        e = max(e, h[c] & 0x7fffu);
        e = max(e, (h[c] >> 16) & 0x7fffu);
      }
    }
    // float16 bits of |hi|: 0x7800 = 32768
    guard = __builtin_fmaxf(guard, __builtin_fmaxf(m, e >= 0x7800u ? 65504.0f : 0.0f));
  };

  if constexpr (!WITH_J) {
    for (int it = 0; it < a.n_iter; ++it) {
      const double beta = a.betas[it];
      forward();
      if constexpr (LOOPS) { ldsq0 = ldsq1 = lwsq0 = lwsq1 = 0.0; }
      backward(beta);
      if constexpr (LOOPS) {
        double num = ldsq0 + ldsq1, den = lwsq0 + lwsq1;     // this lane's 8 NB slots -> the problem's 32 NB
        num += __shfl_xor(num, 16, 64);
        den += __shfl_xor(den, 16, 64);
        num += __shfl_xor(num, 32, 64);
        den += __shfl_xor(den, 32, 64);
        // ||w' - u|| = (1 + beta) ||d||; everything lives at the scale sigma, and so does the reference's 1e-10 floor
        const bool fire = lactive && it >= 3 &&
                          (1.0 + beta) * sqrt(num) / (sqrt(den) + 1.0e-10 * (double)sigma) < a.tol;
        if (__builtin_amdgcn_ballot_w64(fire) != 0) {         // (rare: at most once per problem)
          range_check();                                      // this moment's operands, for the problems that finish now
          float gq = guard, wq = wlast;
          gq = fmaxf(gq, __shfl_xor(gq, 16, 64));
          gq = fmaxf(gq, __shfl_xor(gq, 32, 64));
          wq = fmaxf(wq, __shfl_xor(wq, 16, 64));
          wq = fmaxf(wq, __shfl_xor(wq, 32, 64));
          const bool badq = !(gq < 60000.0f) || (a.rho_guard && wq > 0.0f && (float)th > MFMA_RHO_MAX * wq) || degenerate;
          if (fire) {
            lactive = false;
            if (live && !badq) {
              // (addresses and bounds made HERE, from laundered values: hoisted out of the loop as invariants they
              // cost ~95 registers through the whole solve -- the NB = 10 variant spilled 156 B per lane, round 4)
              double* wrow = a.w + (int64_t)p * a.ldw;
              int tbv = tb;
              asm volatile("" : "+v"(wrow), "+v"(tbv));
#pragma unroll
              for (int q = 0; q < NB; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const int t = MFMA_SPAN * q + tbv + j;
                  if (t < a.N && !(j == 7 && g3)) wrow[t] = w[q][j] * (double)inv_sigma;
                }
            }
            if (live && a.n_done && g == 0) a.n_done[p] = badq ? -1 : it + 1;
          }
          if (__builtin_amdgcn_ballot_w64(lactive && live) == 0) return;       // every problem of the wave has finished
        }
      }
      // (a warm start may overshoot in its first iterations -- the blind loop changes the HRF gain 2-4x between
      // z-steps -- and come back into range before iteration 7: it is checked right after its first pass too)
      if (PB_MFMA_CHECKS && (((it & 7) == 7) || (it == a.n_iter - 1) || (it == 0 && !a.cold))) range_check();
    }
  } else {
    forward();
    if constexpr (CERT) lt[6 * 256] = jw2;       // ||w_0||^2 part
    for (int it = 0; it < a.n_iter; ++it) {
      const double beta = a.betas[it];
      cert_it = it;
      backward(beta);
      forward();
      float sq = jsq, l1 = jl1;                   // this lane's 8 NB slots -> the problem's 32 NB
      sq += __shfl_xor(sq, 16, 64);
      l1 += __shfl_xor(l1, 16, 64);
      sq += __shfl_xor(sq, 32, 64);
      l1 += __shfl_xor(l1, 32, 64);
      if (live && g == 0 && (!CERT || (a.J != nullptr && !cflag))) a.J[(int64_t)p * a.ldj + it] = fmaf(jq, sq, jl * l1);
      if constexpr (CERT) {
        // close the certificate of iteration `it` (the rule is first tested at wind + 1 = 7):
        // sum over the problem's four lanes of v^2 - tol^2 (||w_k||^2 / p1 + 4 ||w_{k+1}||^2 / p2)
        float t = cvsq - cert_t2 * ((1.0001f / CP1) * lt[6 * 256] + (4.0001f / CP2) * jw2);
        t += __shfl_xor(t, 16, 64);
        t += __shfl_xor(t, 32, 64);
        cflag = cflag | ((it >= 7) & !(t >= cert_lim));      // NaN-safe: anything unclear is flagged
        lt[6 * 256] = jw2;
      }
      if (PB_MFMA_CHECKS && (((it & 7) == 7) || (it == a.n_iter - 1) || (it == 0 && !a.cold))) range_check();
    }
  }

  // ---- store (unscaled); a problem that came near the float16 range is handed back ----------
  guard = fmaxf(guard, __shfl_xor(guard, 16, 64));
  guard = fmaxf(guard, __shfl_xor(guard, 32, 64));
  wlast = fmaxf(wlast, __shfl_xor(wlast, 16, 64));
  wlast = fmaxf(wlast, __shfl_xor(wlast, 32, 64));
  // Accuracy guard.  An error eps in the gradient moves an entry of the solution by ~eps th, so the
  // relative error of a solution grows like th / max|w| -- for every arithmetic; the 22-bit operands
  // here start 8x above float32 operators.  Measured along regularisation paths
  // (tools/r3_mfma_precision.py): <= 2e-6 on diff_z for th / max|w| < 0.03, up to 1.5e-5 beyond 0.1.
  // Problems above MFMA_RHO_MAX (sparse solutions, lambda near lambda_max) go back to the float32
  // operators like those that left the float16 range.
  const bool bad = !(guard < 60000.0f) || degenerate ||        // NaN-safe (float16: 65504)
                   (a.rho_guard && wlast > 0.0f && (float)th > MFMA_RHO_MAX * wlast) || (CERT && cflag);
  if (live && !bad && (!LOOPS || lactive)) {
    double* wrow = a.w + (int64_t)p * a.ldw;
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = MFMA_SPAN * q + tb + j;
        if (t < a.N && !(j == 7 && g3)) wrow[t] = w[q][j] * (double)inv_sigma;
      }
  }
  if (live && a.n_done && g == 0 && (!LOOPS || lactive)) a.n_done[p] = bad ? -1 : a.n_iter;
}

// 16 problems per wave, 4 waves per workgroup, one wave per SIMD.  NT near tiles: 2 for K <= 33, 3 for
// K <= 64 (PB_MFMA_NT3; plain solves with or without cost trace and the shared-HRF z-step; not the
// window-rule certificate).
#ifndef PB_MFMA_NT3
#define PB_MFMA_NT3 1
#endif
template <int NB, int NT>
int launch_mfma_nt(const FistaArgs& a, const double* taps, int K, bool with_j, hipStream_t st) {
  const int64_t waves = (launch_count(a) + 15) / 16;
  const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
  const bool cert = a.stop_mode == PB_STOP_WINDOW;
  if (a.stop_mode == PB_STOP_LOOPS) {            // the exact _loops_deconv rule: plain variant with two near tiles only
    if constexpr (NT == 2) {
      if (!a.n_done || a.taps_pp || with_j) return 1;
      const size_t lds0 = (size_t)4 * NB * 2 * 64 * sizeof(u4) + 4 * 64 * sizeof(float);
      const MfmaTaps tp0 = make_mfma_taps(taps, K);
      hipLaunchKernelGGL((fista_mfma_kernel<NB, false, false, false, 2, true>), grid, block, lds0, st, a, tp0);
      return 0;
    } else {
      return 1;
    }
  }
  if (cert && (!a.n_done || a.taps_pp)) return 1;
  if (NT == 3 && cert) return 1;               // (the certificate's state spills beside three near tiles: 260 B per lane)
  const size_t lds = (size_t)4 * NB * 2 * 64 * sizeof(u4) + 4 * (NT == 2 ? 64 : 96) * sizeof(float) +    // residual fragments (8 KB per block of 32 slots), taps
                     (cert ? 7 * 256 * sizeof(float) : 0);                                // certificate state
  if (a.taps_pp) {                              // shared HRF and step in device memory; no cost trace
    const MfmaTaps none{};
    hipLaunchKernelGGL((fista_mfma_kernel<NB, false, true, false, NT>), grid, block, lds, st, a, none);
    return 0;
  }
  const MfmaTaps tp = make_mfma_taps(taps, K);
  if constexpr (NT == 2) {
    if (cert) hipLaunchKernelGGL((fista_mfma_kernel<NB, true, false, true>), grid, block, lds, st, a, tp);
    else if (with_j) hipLaunchKernelGGL((fista_mfma_kernel<NB, true, false>), grid, block, lds, st, a, tp);
    else hipLaunchKernelGGL((fista_mfma_kernel<NB, false, false>), grid, block, lds, st, a, tp);
  } else {
    if (with_j) hipLaunchKernelGGL((fista_mfma_kernel<NB, true, false, false, NT>), grid, block, lds, st, a, tp);
    else hipLaunchKernelGGL((fista_mfma_kernel<NB, false, false, false, NT>), grid, block, lds, st, a, tp);
  }
  return 0;
}

template <int NB>
int launch_mfma(const FistaArgs& a, const double* taps, int K, bool with_j, hipStream_t st) {
  if (a.N > MFMA_SPAN * NB || a.N <= MFMA_SPAN * (NB - 1) || K < 1) return 1;
  if (K <= 33) return launch_mfma_nt<NB, 2>(a, taps, K, with_j, st);
#if PB_MFMA_NT3
  if (K <= 64) return launch_mfma_nt<NB, 3>(a, taps, K, with_j, st);
#endif
  return 1;
}

}  // namespace pb
