// Partition of a regularisation path for pb_fista_solve_path: (voxel, lambda) problems whose lambda lies below
// `ratio` x lambda_max of their series are "dense" (their solutions have many entries far above the threshold: the
// matrix-pipe form's 22-bit operators hold eps there), the others "sparse" (near lambda_max the solution is a few small
// entries: float32 operators).  The reference has no such routine -- its lambda lists are hard-coded "already
// grid-search" values per SNR (examples/icassp_2019/simulation.py:113-114, validation.py:60-62); a batch of such lists
// is BASELINE config 5.
//
// Three small launches, no host synchronisation, deterministic (order-preserving) result in `work`:
//   work[0 .. P)            perm: dense problems ascending from the front, sparse problems ascending from the back
//                           (perm[P-1] = first sparse problem, perm[P-2] = second, ...)
//   work[P]                 number of dense problems
//   work[P+1 .. P+1+nblk)   per-block dense counts, then their exclusive prefix (scratch)
//
// Round 5: the same three launches serve EVERY call of pb_fista_solve(_ex) that the matrix-pipe form can carry -- one
// lambda for the batch or one per problem, with or without cost trace and stop rule (`lbda` may be nullptr: `lbda_s`
// then holds for every problem) -- and, with the predicate "n_done[p] < 0", build the list of the problems a guard or
// a certificate handed back, which the exact vector forms then re-solve at full occupancy instead of one flagged
// problem per wave.  `plan_kernel` (one thread) turns the list lengths into launch plans (plan.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "plan.h"

namespace pb {

constexpr int PATH_THREADS = 256, PATH_PER_THREAD = 16, PATH_PER_BLOCK = PATH_THREADS * PATH_PER_THREAD;

// what puts a problem on the FRONT list
struct ClassPred {
  const double* lbda;      // per-problem lambdas or nullptr
  double lbda_s;           // the one lambda of the batch (lbda == nullptr)
  const double* lmax;      // lambda_max of every series
  int y_rep;
  double ratio;
  const int32_t* n_done;   // != nullptr: the predicate is n_done[p] < 0 instead (handed-back problems)
};

// class of a problem: 0 = front list (dense; or handed back, with the n_done predicate), 1 = the list behind it (sparse),
// 2 = ILL-CONDITIONED (round 5).  A series most of whose energy the operator H = K_h . cumsum does not see (alternating
// signs, fast sinusoids, high-pass noise, an ordinary signal under a strong fast carrier) loses digits in every
// arithmetic narrower than float64: with the coherence gamma_2 = lambda_max / (||y||_2 sum|c| / sqrt(N)), c = cumsum(h)
// (ordinary block signals at SNR 1 dB: 0.25 .. 0.5, white noise: 0.02 .. 0.2), the matrix-pipe form reaches 1e-5 on
// diff_z / z / x below gamma_2 ~ 3e-2 and 4e-4 below 3e-3, the float32 vector forms 7e-6 below 5e-3 and 5e-3 below 1e-3
// (tools/r5_gamma_calibration.py, profiles/r5_gamma_calibration_*.txt).  The lambda_max pass therefore marks series with
// gamma_2 < 1e-2 by a NEGATIVE lambda_max (class 2: float64 LDS kernel) and stores 0 for gamma_2 < 7e-2 (class 1
// whatever lambda: float32 vector forms, never the matrix pipe).
__device__ __forceinline__ int path_class(const ClassPred& c, int p) {
  if (c.n_done) return c.n_done[p] < 0 ? 0 : 1;
  const double lm = c.lmax[p / c.y_rep];
  if (lm < 0.0) return 2;
  const double lb = c.lbda ? c.lbda[p] : c.lbda_s;
  return lb < c.ratio * lm ? 0 : 1;                 // (lambda_max = 0, an all-zero series: sparse -- its solution is 0)
}

// layout of the counters behind the list array (int32 units from work + P): [0] n_front, [1, 1+nblk) front counts per block,
// [1+nblk, 1+2 nblk) ill counts per block, [1+2 nblk] n_ill
__host__ __device__ inline int path_nill_offset(int nblk) { return 1 + 2 * nblk; }

// front and ill problems of every block of 4 096
__global__ __launch_bounds__(PATH_THREADS) void path_count_kernel(ClassPred cp, int P, int32_t* work) {
  __shared__ int part[2][PATH_THREADS / 64];
  const int p0 = blockIdx.x * PATH_PER_BLOCK + threadIdx.x * PATH_PER_THREAD;
  int c = 0, ci = 0;
  for (int i = 0; i < PATH_PER_THREAD; ++i) {
    if (p0 + i >= P) break;
    const int k = path_class(cp, p0 + i);
    c += k == 0;
    ci += k == 2;
  }
  for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); ci += __shfl_xor(ci, o, 64); }
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = c; part[1][threadIdx.x >> 6] = ci; }
  __syncthreads();
  if (threadIdx.x == 0) {
    work[P + 1 + blockIdx.x] = part[0][0] + part[0][1] + part[0][2] + part[0][3];
    work[P + 1 + gridDim.x + blockIdx.x] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
  }
}

// exclusive prefix of the block counts (in place) and the total -- one workgroup, blocks dealt in chunks of 256
struct PlanSpec {          // how a list is to be laid over the forms (the arguments of plan.h's planners)
  int kind;                // 0: nothing, 1: plan_pieces_mfma (matrix-pipe form + vector remainder), 2: plan_pieces (vector forms),
                           // 3: series of 311..640 scans, dense class: whole passes of the split form (and a remainder above
                           //    5/16 of a pass), the rest on the backup form; 4: the same shapes, sparse class or handed-back
                           //    problems: the pair form over two slots from `min_pair` problems on, else the backup form
  int has_pair, has_wide, one_launch, one_stream, has_mfma2, beside_chunks;
  double slots;
  int backup_form = FORM_WIDE;   // kinds 3, 4: FORM_FAST1 (311..320 scans with a single-row entry) or FORM_WIDE
  int min_pair = 1024;
  int pass_mult = 4;             // kind 3: a pass of the split form = pass_mult * slots problems (two waves: 4; four waves,
  int rem_num16 = 5;             // fista_mfma4.h: 2); a remainder above rem_num16 / 16 of a pass runs on it too
  int merged = 0;                // front spec only: 1 = ONE plan for the whole call (plan.h: plan_partitioned / its long-series
                                 // form): the front list's remainder joins the back list, ranges are positions in the list array
};
__device__ __forceinline__ void plan_list(const PlanSpec& sp, int n, int32_t* ranges);
__device__ __forceinline__ void plan_call(const PlanSpec& front, const PlanSpec& back, int n_d, int P, int32_t* ranges);

// ... and, once the total is known, the launch plans of the two lists (thread 0: plan.h's planners on the list lengths)
__global__ __launch_bounds__(PATH_THREADS) void path_scan_kernel(int P, int nblk, int32_t* work, PlanSpec front, PlanSpec back,
                                                                  int32_t* ranges_front, int32_t* ranges_back, int32_t* range_ill) {
  __shared__ int buf[PATH_THREADS];
  __shared__ int carry;
  int total[2] = {0, 0};
  for (int which = 0; which < 2; ++which) {         // 0: front counts, 1: ill counts
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    int32_t* cnt = work + P + 1 + which * nblk;
    for (int b0 = 0; b0 < nblk; b0 += PATH_THREADS) {
      const int b = b0 + (int)threadIdx.x;
      const int v = b < nblk ? cnt[b] : 0;
      buf[threadIdx.x] = v;
      __syncthreads();
      for (int o = 1; o < PATH_THREADS; o <<= 1) {    // inclusive Hillis-Steele scan of 256 counts
        const int add = threadIdx.x >= (unsigned)o ? buf[threadIdx.x - o] : 0;
        __syncthreads();
        buf[threadIdx.x] += add;
        __syncthreads();
      }
      if (b < nblk) cnt[b] = carry + buf[threadIdx.x] - v;
      __syncthreads();
      if (threadIdx.x == PATH_THREADS - 1) carry += buf[PATH_THREADS - 1];
      __syncthreads();
    }
    total[which] = carry;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int n_front = total[0], n_ill = total[1], P_eff = P - n_ill;
    work[P] = n_front;
    work[P + path_nill_offset(nblk)] = n_ill;
    if (range_ill) { range_ill[0] = P_eff; range_ill[1] = P; }       // the ill-conditioned problems: the tail of the list array
    if (ranges_front && front.merged) plan_call(front, back, n_front, P_eff, ranges_front);
    else {
      if (ranges_front) plan_list(front, n_front, ranges_front);
      if (ranges_back) plan_list(back, P_eff - n_front, ranges_back);
    }
  }
}

// the lists: front problems ascending from position 0, the others ascending behind them (from n_front), ill-conditioned
// ones descending from the end -- every list is read as list[slot]
__global__ __launch_bounds__(PATH_THREADS) void path_scatter_kernel(ClassPred cp, int P, int32_t* work) {
  __shared__ int buf[2][PATH_THREADS];
  const int p0 = blockIdx.x * PATH_PER_BLOCK + threadIdx.x * PATH_PER_THREAD;
  unsigned mask = 0, maski = 0;
  int c = 0, ci = 0;
  for (int i = 0; i < PATH_PER_THREAD; ++i) {
    if (p0 + i >= P) break;
    const int k = path_class(cp, p0 + i);
    if (k == 0) { mask |= 1u << i; ++c; }
    if (k == 2) { maski |= 1u << i; ++ci; }
  }
  buf[0][threadIdx.x] = c;
  buf[1][threadIdx.x] = ci;
  __syncthreads();
  for (int o = 1; o < PATH_THREADS; o <<= 1) {
    const int add = threadIdx.x >= (unsigned)o ? buf[0][threadIdx.x - o] : 0;
    const int addi = threadIdx.x >= (unsigned)o ? buf[1][threadIdx.x - o] : 0;
    __syncthreads();
    buf[0][threadIdx.x] += add;
    buf[1][threadIdx.x] += addi;
    __syncthreads();
  }
  const int n_front = work[P];
  int d = work[P + 1 + blockIdx.x] + buf[0][threadIdx.x] - c;                  // front problems before this thread's first one
  int e = work[P + 1 + gridDim.x + blockIdx.x] + buf[1][threadIdx.x] - ci;     // ill-conditioned problems before it
  for (int i = 0; i < PATH_PER_THREAD; ++i) {
    const int p = p0 + i;
    if (p >= P) break;
    if (mask & (1u << i)) work[d++] = p;
    else if (maski & (1u << i)) work[P - 1 - (e++)] = p;
    else work[n_front + (p - d - e)] = p;          // p - d - e = problems of the middle class before p
  }
}

// ---- launch plans of the two lists, made where their lengths are known (called by path_scan_kernel) --------------
__device__ __forceinline__ void plan_list(const PlanSpec& sp, int n, int32_t* ranges) {
  Piece pc[MAX_PIECES];
  int npc = 0;
  if (n > 0 && sp.kind == 1)
    npc = plan_pieces_mfma(n, sp.has_pair != 0, sp.has_wide != 0, sp.one_launch != 0, sp.one_stream != 0, sp.has_mfma2 != 0,
                           sp.beside_chunks, sp.slots, pc);
  else if (n > 0 && sp.kind == 2)
    npc = plan_pieces(n, sp.has_pair != 0, sp.has_wide != 0, sp.one_launch != 0, sp.one_stream != 0, sp.slots, pc);
  else if (n > 0 && sp.kind == 3) {
    const int pass = (int)sp.slots * sp.pass_mult; // 16 problems x (slots / 2 SIMDs / 2 waves): capi.hip, mfma2_long_base / mfma4_base
    int base = (n / pass) * pass;
    if (sp.one_launch || n - base > pass * sp.rem_num16 / 16) base = n;
    if (base > 0) pc[npc++] = Piece{FORM_MFMA2, 0, base, false, false};
    if (base < n) pc[npc++] = Piece{sp.backup_form, base, n, false, false};
  } else if (n > 0 && sp.kind == 4) {
    pc[npc++] = Piece{(sp.has_pair && n >= sp.min_pair) ? FORM_PAIR : sp.backup_form, 0, n, false, false};
  }
  if (plan_to_candidates(pc, npc, ranges) != 0) {
    // (unreachable for the plans of plan.h; if it ever happened: everything on the one candidate that always exists)
    for (int c = 0; c < 2 * CAND_COUNT; ++c) ranges[c] = 0;
    const int c = sp.kind == 1 ? CAND_MFMA : (sp.kind == 3 ? CAND_MFMA2 : ((sp.kind == 4 && sp.backup_form == FORM_WIDE) ? CAND_WIDE : CAND_FAST0));
    ranges[2 * c + 1] = n;
  }
}

// one plan for a partitioned call: dense problems at positions [0, n_d), sparse ones behind (see PlanSpec::merged)
__device__ __forceinline__ void plan_call(const PlanSpec& front, const PlanSpec& back, int n_d, int P, int32_t* ranges) {
  Piece pc[MAX_PIECES];
  int npc = 0;
  if (front.kind == 1) {
    npc = plan_partitioned(n_d, P, front.has_pair != 0, front.has_wide != 0, front.one_stream != 0, front.has_mfma2 != 0,
                           front.beside_chunks, front.slots, pc);
  } else {                                           // series of 311..1 280 scans (kinds 3 / 4)
    const int pass = (int)front.slots * front.pass_mult;
    int base = (n_d / pass) * pass;
    if (n_d - base > pass * front.rem_num16 / 16) base = n_d;
    if (base > 0) pc[npc++] = Piece{FORM_MFMA2, 0, base, false, false};
    if (base < P) pc[npc++] = Piece{(back.has_pair && P - base >= back.min_pair) ? FORM_PAIR : back.backup_form, base, P, false, false};
  }
  if (plan_to_candidates(pc, npc, ranges) != 0) {     // (unreachable; if it ever happened: everything on the exact single-row / wide form)
    for (int c = 0; c < 2 * CAND_COUNT; ++c) ranges[c] = 0;
    const int c = (front.kind != 1 && back.backup_form == FORM_WIDE) ? CAND_WIDE : CAND_FAST0;
    ranges[2 * c + 1] = P;
  }
}

// ---- lambda_max of every series, for the partition (float32 in, float64 out) ---------------------------------
// || H^T y ||_inf with H^T y = T_c^T y, c = cumsum(h): (H^T y)[j] = sum_m h[m] s[j + m], s = suffix sums of y
// (upper-triangular Toeplitz matrices commute).  One series per wave: its lanes hold strips of SL consecutive
// samples (loaded straight from HBM: staging the series through LDS for coalesced loads measured 18 % slower -- the
// pass is bound by the latency of its 100 000 short waves, not by bandwidth), the suffix sums go through LDS (stride-SL
// reads: conflict-free for odd SL), the taps come as kernel arguments.  Float32 arithmetic: the result only decides a class (ratio test at 13 %), and pb_lambda_max stays the
// float64 answer for callers who want the number.  The pass also sees max|y| and marks ILL-CONDITIONED series (path_class).
constexpr int LMAX_KT = 64;
struct LmaxTaps { float h[LMAX_KT]; };

template <int SL>
__global__ __launch_bounds__(256) void lmax_wave_kernel(const float* y, int64_t ldy, int V, int N, LmaxTaps tp, int K, double* out,
                                                        float f64_bound, float vec_bound) {
  extern __shared__ float lm_smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int v = blockIdx.x * 4 + wv;
  if (v >= V) return;                                      // (wave-level synchronisation only below)
  float* s = lm_smem + wv * (64 * SL + LMAX_KT);
  const float* yrow = y + (int64_t)v * ldy;
  float loc[SL];
  float run = 0.0f, ysq = 0.0f;
#pragma unroll
  for (int j = SL - 1; j >= 0; --j) {
    const int t = lane * SL + j;
    const float yv = t < N ? yrow[t] : 0.0f;
    ysq = fmaf(yv, yv, ysq);
    run += yv;
    loc[j] = run;                                          // suffix sums inside the strip
  }
  float above = run;                                       // exclusive suffix over the lanes above
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(above, o, 64);
    above += (lane + o < 64) ? t : 0.0f;
  }
  above -= run;
#pragma unroll
  for (int j = 0; j < SL; ++j) s[lane * SL + j] = loc[j] + above;
  if (lane < LMAX_KT) s[64 * SL + lane] = 0.0f;            // past the end: zeros
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float acc[SL];
#pragma unroll
  for (int j = 0; j < SL; ++j) acc[j] = 0.0f;
  for (int k0 = 0; k0 < K; k0 += 8) {                      // eight taps per scalar load (taps beyond K are zero), SL independent chains
    float hk[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) hk[q] = tp.h[k0 + q];
#pragma unroll
    for (int q = 0; q < 8; ++q)
#pragma unroll
      for (int j = 0; j < SL; ++j) acc[j] = fmaf(hk[q], s[lane * SL + j + k0 + q], acc[j]);
  }
  float m = 0.0f;
#pragma unroll
  for (int j = 0; j < SL; ++j) m = fmaxf(m, lane * SL + j < N ? fabsf(acc[j]) : 0.0f);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); ysq += __shfl_xor(ysq, o, 64); }
  // coherence gamma_2 = lambda_max / (||y||_2 sum|c| / sqrt(N)) (the bounds carry gamma sum|c| / sqrt(N)):
  //   below the float64 bound: marked by the SIGN (path_class: class 2);  below the matrix-pipe bound: stored as 0, so
  //   that `lbda < ratio * lambda_max` is false and the series takes the float32 vector forms whatever its lambda
  const float yn = __builtin_sqrtf(ysq);
  if (lane == 0) out[v] = (m > 0.0f && m < f64_bound * yn) ? -(double)m : ((m < vec_bound * yn) ? 0.0 : (double)m);
}

}  // namespace pb
