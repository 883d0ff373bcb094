// Any-size float64 kernels: one 256-thread workgroup per row, the row staged in
// LDS.  They serve (i) the operator surface (.op/.adj of DiscretInteg and
// ConvAndLinear, pybold/linear.py:9-113; toeplitz products,
// pybold/convolution.py:105-132), (ii) the outputs z, x of deconv
// (pybold/bold_signal.py:74-75), (iii) hrf_fit_err (pybold/bold_signal.py:217-222)
// and (iv) the FISTA recurrence for shapes the register-resident kernel does not
// cover, including the windowed early-stopping rule (pybold/bold_signal.py:82-95).
#pragma once
#include "common.h"
#include "fista_fast.h"  // FistaArgs

namespace pb {

constexpr int GEN_THREADS = 256;
constexpr int GEN_WAVES = GEN_THREADS / 64;

// ---- block-wide primitives (all threads of the workgroup must call) ---------

// sum of one double per thread, result in every thread; red = 2*GEN_WAVES doubles of LDS
__device__ __forceinline__ double block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wid = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wid] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < GEN_WAVES; ++i) s += red[i];
  return s;
}

// dst[i] = sum_{j<=i} src[j] (REVERSE=false) or sum_{j>=i} src[j] (REVERSE=true),
// i in [0, n).  src/dst in LDS, may alias.  Each thread owns a contiguous chunk.
template <bool REVERSE>
__device__ __forceinline__ void block_cumsum(const double* src, double* dst, int n, double* red) {
  const int chunk = (n + GEN_THREADS - 1) / GEN_THREADS;
  const int t = threadIdx.x;
  const int lo = t * chunk;
  const int hi = min(lo + chunk, n);
  auto idx = [&](int i) { return REVERSE ? n - 1 - i : i; };
  double local = 0.0;
  for (int i = lo; i < hi; ++i) local += src[idx(i)];
  // exclusive scan of the per-thread totals
  double incl = local;
  const int lane = t & 63, wid = t >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  __syncthreads();
  if (lane == 63) red[wid] = incl;
  __syncthreads();
  double woff = 0.0;
  for (int i = 0; i < wid; ++i) woff += red[i];
  double run = woff + incl - local;
  __syncthreads();
  // dst may alias src: a thread reads an element of its own chunk right before
  // it overwrites it and never touches another thread's chunk
  for (int i = lo; i < hi; ++i) {
    run += src[idx(i)];
    dst[idx(i)] = run;
  }
  __syncthreads();
}

// out[i] = sum_m k[m] in[i-m], 0 <= i-m < n_in, i in [0, n_out)   (toeplitz @ in)
__device__ __forceinline__ void block_conv(const double* in, int n_in, double* out, int n_out,
                                           const double* k, int K) {
  for (int i = threadIdx.x; i < n_out; i += GEN_THREADS) {
    const int m0 = max(0, i - n_in + 1);
    const int m1 = min(K - 1, i);
    double acc = 0.0;
    for (int m = m0; m <= m1; ++m) acc = fma(k[m], in[i - m], acc);
    out[i] = acc;
  }
  __syncthreads();
}

// out[j] = sum_m k[m] r[j+m], j+m < n_r, j in [0, n_out)           (toeplitz^T @ r)
__device__ __forceinline__ void block_corr(const double* r, int n_r, double* out, int n_out,
                                           const double* k, int K) {
  for (int j = threadIdx.x; j < n_out; j += GEN_THREADS) {
    const int m1 = min(K - 1, n_r - 1 - j);
    double acc = 0.0;
    for (int m = 0; m <= m1; ++m) acc = fma(k[m], r[j + m], acc);
    out[j] = acc;
  }
  __syncthreads();
}

// ---- operator kernels -------------------------------------------------------
enum OpKind { OP_INTEG = 0, OP_INTEG_ADJ = 1, OP_CONV = 2, OP_CORR = 3, OP_FWD = 4, OP_ADJ = 5 };

// LDS: a[nmax] b[nmax] k[K] red[8]
template <int KIND>
__global__ __launch_bounds__(GEN_THREADS) void op_kernel(const double* x, int64_t ldx, double* out,
                                                         int64_t ldo, int n_src, int n_dst,
                                                         const double* taps, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nmax = max(n_src, n_dst);
  double* a = reinterpret_cast<double*>(smem);
  double* b = a + nmax;
  double* k = b + nmax;
  double* red = k + K;
  const double* xr = x + (int64_t)blockIdx.x * ldx;
  double* orow = out + (int64_t)blockIdx.x * ldo;
  for (int i = threadIdx.x; i < n_src; i += GEN_THREADS) a[i] = xr[i];
  for (int i = threadIdx.x; i < K; i += GEN_THREADS) k[i] = taps[i];
  __syncthreads();
  double* res = b;
  if constexpr (KIND == OP_INTEG) {
    block_cumsum<false>(a, b, n_src, red);
  } else if constexpr (KIND == OP_INTEG_ADJ) {
    block_cumsum<true>(a, b, n_src, red);
  } else if constexpr (KIND == OP_CONV) {
    block_conv(a, n_src, b, n_dst, k, K);
  } else if constexpr (KIND == OP_CORR) {
    block_corr(a, n_src, b, n_dst, k, K);
  } else if constexpr (KIND == OP_FWD) {       // K . cumsum
    block_cumsum<false>(a, a, n_src, red);
    block_conv(a, n_src, b, n_dst, k, K);
  } else {                                     // revcumsum . K^T
    block_corr(a, n_src, b, n_dst, k, K);
    block_cumsum<true>(b, b, n_dst, red);
  }
  for (int i = threadIdx.x; i < n_dst; i += GEN_THREADS) orow[i] = res[i];
}

// z = cumsum(w), x = k * z
__global__ __launch_bounds__(GEN_THREADS) void outputs_kernel(const double* w, int64_t ldw, int N,
                                                              const double* taps, int64_t ldt, int K,
                                                              double* z, int64_t ldz, double* x,
                                                              int64_t ldx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* a = reinterpret_cast<double*>(smem);
  double* b = a + N;
  double* k = b + N;
  double* red = k + K;
  const double* wr = w + (int64_t)blockIdx.x * ldw;
  const double* tp = taps + (int64_t)blockIdx.x * ldt;     // ldt = 0: taps shared by all rows
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) a[i] = wr[i];
  for (int i = threadIdx.x; i < K; i += GEN_THREADS) k[i] = tp[i];
  __syncthreads();
  block_cumsum<false>(a, a, N, red);
  if (z) {
    double* zr = z + (int64_t)blockIdx.x * ldz;
    for (int i = threadIdx.x; i < N; i += GEN_THREADS) zr[i] = a[i];
  }
  if (x) {
    block_conv(a, N, b, N, k, K);
    double* xr = x + (int64_t)blockIdx.x * ldx;
    for (int i = threadIdx.x; i < N; i += GEN_THREADS) xr[i] = b[i];
  }
}

// Two-gamma SPM HRF sampled at K given times for M dilations (pybold/hrf_model.py:25-31,
// un-normalised): out[m][k] = pdf(d_m t_k; a1, loc1) - ratio * pdf(d_m t_k; a2, loc2) with the
// gamma density pdf(x; a, loc) = exp((a-1) log(x-loc) - (x-loc) - lgamma(a)) for x > loc, else 0.
__global__ __launch_bounds__(GEN_THREADS) void spm_hrf_kernel(const double* deltas, int M,
                                                              const double* t, int K, double a1,
                                                              double loc1, double lg1, double a2,
                                                              double loc2, double lg2, double ratio,
                                                              double* out) {
  const int64_t idx = (int64_t)blockIdx.x * GEN_THREADS + threadIdx.x;
  if (idx >= (int64_t)M * K) return;
  const int m = (int)(idx / K), k = (int)(idx % K);
  const double x = deltas[m] * t[k];
  auto pdf = [](double v, double a, double lg) {
    return v > 0.0 ? exp((a - 1.0) * log(v) - v - lg) : 0.0;
  };
  out[idx] = pdf(x - loc1, a1, lg1) - ratio * pdf(x - loc2, a2, lg2);
}

// L[p] = || A^T A ||_F, A = toeplitz(h_p) tril(1): the _loops_deconv Lipschitz constant
// (pybold/bold_signal.py:249-253) for P different HRFs.  A is lower-triangular
// Toeplitz with kernel c = cumsum(h); (A^T A)[j, j+d] = R_d(N-1-j-d) with the
// partial autocorrelations R_d(T) = sum_{t<=T} c[t] c[t+d]; thread d walks diagonal d.
__global__ __launch_bounds__(GEN_THREADS) void gram_frobenius_kernel(const double* taps, int64_t ldt,
                                                                     int K, int N, double* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* c = reinterpret_cast<double*>(smem);
  double* red = c + N;
  const double* h = taps + (int64_t)blockIdx.x * ldt;
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) c[i] = (i < K) ? h[i] : 0.0;
  __syncthreads();
  block_cumsum<false>(c, c, N, red);
  double part = 0.0;
  for (int d = threadIdx.x; d < N; d += GEN_THREADS) {
    double r = 0.0, acc = 0.0;
    for (int t = 0; t < N - d; ++t) {
      r = fma(c[t], c[t + d], r);
      acc = fma(r, r, acc);
    }
    part += (d == 0 ? 1.0 : 2.0) * acc;
  }
  const double tot = block_sum(part, red);
  if (threadIdx.x == 0) out[blockIdx.x] = sqrt(tot);
}

// The same constant in O(K^2 + N) instead of O(N^2 / 2) per HRF, for K <= N.  The kernel of A,
// c = cumsum(h), is constant from t = K-1 on (c[t] = S = sum h), so for T >= K-2
//   R_d(T) = alpha_d + (T - (K-2)) S^2,   alpha_d = R_d(K-2) = sum_{t<=K-2} c[t] c[min(t+d, K-1)]
// and sum_T R_d(T)^2 over that range is a quadratic-polynomial sum in closed form; only the
// first K-2 partial sums of the K-1 diagonals with d <= K-2 are accumulated term by term
// (for d >= K-1 they are S * prefix(c), shared by all those diagonals).  One wave per HRF.
// LDS: c[K] C[K] E[K].
// (one wave; h may live in LDS or global memory; lds3k = 3 K doubles of scratch; the result is
// valid in lane 0 -- all 64 lanes must call)
__device__ __forceinline__ double gram_frobenius_fir_wave(const double* h, int K, int N, double* lds3k,
                                                          int lane) {
  double* c = lds3k;                             // c[t], t < K (c[K-1] = S)
  double* C = c + K;                             // C[T] = sum_{t<=T} c[t]
  double* E = C + K;                             // E[T] = sum_{T'<=T} (S C[T'])^2
  if (lane == 0) {
    double run = 0.0;
    for (int t = 0; t < K; ++t) { run += h[t]; c[t] = run; }
    const double S0 = run;
    double cs = 0.0, es = 0.0;
    for (int t = 0; t < K; ++t) {
      cs += c[t];
      C[t] = cs;
      es = fma(S0 * cs, S0 * cs, es);
      E[t] = es;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  const double S = c[K - 1], S2 = S * S;
  double part = 0.0;
  for (int d = lane; d < N; d += 64) {
    const int Tmax = N - 1 - d;                  // valid T: 0 .. Tmax
    double direct = 0.0, alpha = 0.0;
    if (d <= K - 2) {
      double r = 0.0;
      for (int t = 0; t <= K - 2; ++t) {
        const int td = t + d < K - 1 ? t + d : K - 1;
        r = fma(c[t], c[td], r);
        if (t <= K - 3 && t <= Tmax) direct = fma(r, r, direct);
      }
      alpha = r;
    } else {
      const int last = (K - 3 < Tmax ? K - 3 : Tmax);
      direct = last >= 0 ? E[last] : 0.0;
      alpha = K >= 2 ? S * C[K - 2] : 0.0;
    }
    double tot = direct;
    if (Tmax >= K - 2) {
      const double M = (double)(Tmax - (K - 2));
      const double s1 = 0.5 * M * (M + 1.0), s2 = M * (M + 1.0) * (2.0 * M + 1.0) / 6.0;
      tot += (M + 1.0) * alpha * alpha + 2.0 * alpha * S2 * s1 + S2 * S2 * s2;
    }
    part += (d == 0 ? 1.0 : 2.0) * tot;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) part += __shfl_xor(part, o, 64);
  return sqrt(part);
}

__global__ __launch_bounds__(64) void gram_frobenius_fir_kernel(const double* taps, int64_t ldt, int K,
                                                               int N, double* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const double r = gram_frobenius_fir_wave(taps + (int64_t)blockIdx.x * ldt, K, N,
                                           reinterpret_cast<double*>(smem), threadIdx.x);
  if (threadIdx.x == 0) out[blockIdx.x] = r;
}

// cost[c][v] = 0.5 || y_v - taps * z_v ||^2 ; grid = (V, n_hrf).  taps index:
// shared candidates taps[c][K] (per_voxel = 0) or one HRF per (candidate, voxel)
// taps[c][v][K] (per_voxel = 1).
template <typename TY>
__global__ __launch_bounds__(GEN_THREADS) void hrf_cost_kernel(const double* z, int64_t ldz,
                                                               const TY* y, int64_t ldy, int V,
                                                               int N, const double* taps, int K,
                                                               double* cost, int per_voxel) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* a = reinterpret_cast<double*>(smem);
  double* k = a + N;
  double* red = k + K;
  const int v = blockIdx.x, c = blockIdx.y;
  const double* zr = z + (int64_t)v * ldz;
  const TY* yr = y + (int64_t)v * ldy;
  const double* tc = taps + (per_voxel ? ((int64_t)c * V + v) * K : (int64_t)c * K);
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) a[i] = zr[i];
  for (int i = threadIdx.x; i < K; i += GEN_THREADS) k[i] = tc[i];
  __syncthreads();
  double sq = 0.0;
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) {
    const int m1 = min(K - 1, i);
    double acc = 0.0;
    for (int m = 0; m <= m1; ++m) acc = fma(k[m], a[i - m], acc);
    const double d = (double)yr[i] - acc;
    sq = fma(d, d, sq);
  }
  const double tot = block_sum(sq, red);
  if (threadIdx.x == 0) cost[(int64_t)c * V + v] = 0.5 * tot;
}

// Power iteration of pybold/utils.py:94-109 on H^T H, H = toeplitz(taps) . cumsum, entirely in
// one workgroup: x <- H^T H x / ||x||, stop when | ||x_new|| - ||x_old|| | < tol or after nb_iter
// steps; out[0] = ||x_new||, out[1] = steps done.  LDS: x[N] a[N] b[N] k[K] red[8].
__global__ __launch_bounds__(GEN_THREADS) void power_iter_kernel(const double* x0, int N,
                                                                 const double* taps, int K,
                                                                 int nb_iter, double tol,
                                                                 double* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* x = reinterpret_cast<double*>(smem);
  double* a = x + N;
  double* b = a + N;
  double* k = b + N;
  double* red = k + K;
  double part = 0.0;
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) {
    x[i] = x0[i];
    part = fma(x[i], x[i], part);
  }
  for (int i = threadIdx.x; i < K; i += GEN_THREADS) k[i] = taps[i];
  double n_old = sqrt(block_sum(part, red));
  double n_new = n_old;
  int it = 0;
  for (; it < nb_iter; ++it) {
    block_cumsum<false>(x, a, N, red);
    block_conv(a, N, b, N, k, K);
    block_corr(b, N, a, N, k, K);
    block_cumsum<true>(a, a, N, red);
    part = 0.0;
    for (int i = threadIdx.x; i < N; i += GEN_THREADS) {
      x[i] = a[i] / n_old;
      part = fma(x[i], x[i], part);
    }
    n_new = sqrt(block_sum(part, red));
    if (fabs(n_new - n_old) < tol) { ++it; break; }
    n_old = n_new;
  }
  if (threadIdx.x == 0) {
    out[0] = n_new;
    out[1] = (double)it;
  }
}

// r2[p] = || taps * cumsum(w_p) - y_p ||^2 , l1[p] = || w_p ||_1
// (the quantities R, G the lambda search tracks, pybold/bold_signal.py:141-157)
template <typename TY>
__global__ __launch_bounds__(GEN_THREADS) void stats_kernel(const double* w, int64_t ldw,
                                                            const TY* y, int64_t ldy, int y_rep,
                                                            int N, const double* taps, int K,
                                                            double* r2, double* l1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* a = reinterpret_cast<double*>(smem);
  double* b = a + N;
  double* k = b + N;
  double* red = k + K;
  const int p = blockIdx.x;
  const double* wr = w + (int64_t)p * ldw;
  const TY* yr = y + (int64_t)(p / y_rep) * ldy;
  double part_l1 = 0.0;
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) {
    a[i] = wr[i];
    part_l1 += fabs(a[i]);
  }
  for (int i = threadIdx.x; i < K; i += GEN_THREADS) k[i] = taps[i];
  __syncthreads();
  block_cumsum<false>(a, a, N, red);
  block_conv(a, N, b, N, k, K);
  double part_r2 = 0.0;
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) {
    const double d = b[i] - (double)yr[i];
    part_r2 = fma(d, d, part_r2);
  }
  const double tot_r2 = block_sum(part_r2, red);
  const double tot_l1 = block_sum(part_l1, red);
  if (threadIdx.x == 0) {
    r2[p] = tot_r2;
    l1[p] = tot_l1;
  }
}

// max of one double per thread, result in every thread; red = 2*GEN_WAVES doubles of LDS
__device__ __forceinline__ double block_max(double v, double* red) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  const int wid = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wid] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int i = 1; i < GEN_WAVES; ++i) s = fmax(s, red[i]);
  return s;
}

// out[v] = || H^T y_v ||_inf, H = toeplitz(taps) . cumsum: the smallest lambda for which
// the solution of min 0.5||H w - y||^2 + lambda ||w||_1 is w = 0, i.e. the top of a
// regularisation path (the reference hard-codes its lambda lists,
// examples/icassp_2019/simulation.py:113-114).  LDS: a[N] b[N] k[K] red[8].
template <typename TY>
__global__ __launch_bounds__(GEN_THREADS) void lambda_max_kernel(const TY* y, int64_t ldy, int N,
                                                                 const double* taps, int K,
                                                                 double* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* a = reinterpret_cast<double*>(smem);
  double* b = a + N;
  double* k = b + N;
  double* red = k + K;
  const TY* yr = y + (int64_t)blockIdx.x * ldy;
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) a[i] = (double)yr[i];
  for (int i = threadIdx.x; i < K; i += GEN_THREADS) k[i] = taps[i];
  __syncthreads();
  block_corr(a, N, b, N, k, K);
  block_cumsum<true>(b, b, N, red);
  double m = 0.0;
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) m = fmax(m, fabs(b[i]));
  m = block_max(m, red);
  if (threadIdx.x == 0) out[blockIdx.x] = m;
}

// out = x / (max|x| + 1e-12) per row (inf_norm, pybold/utils.py:112-138); rows of any length
// (two passes over global memory, nothing staged)
__global__ __launch_bounds__(GEN_THREADS) void inf_norm_kernel(const double* x, int64_t ldx,
                                                               double* out, int64_t ldo, int64_t n) {
  __shared__ double red[2 * GEN_WAVES];
  const double* xr = x + (int64_t)blockIdx.x * ldx;
  double* orow = out + (int64_t)blockIdx.x * ldo;
  double m = 0.0, has_nan = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += GEN_THREADS) {
    const double v = fabs(xr[i]);
    m = fmax(m, v);
    if (v != v) has_nan = 1.0;                 // np.max propagates NaN, fmax drops it
  }
  m = block_max(m, red);
  has_nan = block_max(has_nan, red);
  const double d = (has_nan > 0.0 ? __builtin_nan("") : m) + 1.0e-12;
  for (int64_t i = threadIdx.x; i < n; i += GEN_THREADS) orow[i] = xr[i] / d;
}

// ---- generic FISTA: one workgroup per problem, float64 state in LDS ---------
// LDS: w[N] a[N] b[N] k[K] red[8] hist[wind*N] (window rule only)
// F64IO: y and the cost trace are float64 (a_.y64 / a_.J64): the reference's arithmetic
// end to end, used for single-voxel / small-batch calls (pb_fista_solve_d).
template <bool WITH_J, bool F64IO = false>
__global__ __launch_bounds__(GEN_THREADS) void fista_generic_kernel(FistaArgs a_, const double* taps,
                                                                    int K, int wind) {
  using TY = std::conditional_t<F64IO, double, float>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a_.N;
  double* w = reinterpret_cast<double*>(smem);
  double* a = w + N;
  double* b = a + N;
  double* k = b + N;
  double* red = k + K;
  double* hist = red + 2 * GEN_WAVES;
  // a launch over a device-side list (round 5: the ill-conditioned problems of a partitioned call): the workgroups stride
  // over the slots [range[0], range[1]) of the list array, which may hold none or many
  int slot = a_.range ? a_.range[0] + (int)blockIdx.x : 0;
  const int slot_end = a_.range ? a_.range[1] : 1;
  for (; slot < slot_end; slot += (a_.range ? (int)gridDim.x : 1)) {
  const int p = a_.range ? a_.perm[slot] : (int)blockIdx.x + a_.p0;
  const TY* yr = [&] {
    if constexpr (F64IO) return a_.y64 + (int64_t)(p / a_.y_rep) * a_.ldy;
    else return a_.y + (int64_t)(p / a_.y_rep) * a_.ldy;
  }();
  double* wrow = a_.w + (int64_t)p * a_.ldw;
  const double* tp = a_.taps_pp ? a_.taps_pp + (int64_t)p * a_.ldt : taps;
  const double stp = a_.step_vec ? a_.step_vec[a_.step_shared ? 0 : p] : a_.step;
  __syncthreads();                                  // (the previous problem of this workgroup is out of LDS)
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) w[i] = a_.cold ? 0.0 : wrow[i];
  for (int i = threadIdx.x; i < K; i += GEN_THREADS) k[i] = tp[i];
  __syncthreads();
  const double lb = a_.lbda_vec ? a_.lbda_vec[p] : a_.lbda;
  const double th = lb * stp;
  const int stop = a_.stop_mode;
  using TJ = std::conditional_t<F64IO, double, float>;
  TJ* Jrow = nullptr;
  if constexpr (WITH_J) {
    if constexpr (F64IO) Jrow = a_.J64 + (int64_t)p * a_.ldj;
    else Jrow = a_.J + (int64_t)p * a_.ldj;
  }

  int n_stop = a_.n_iter;   // iterations to execute (shrinks when the stop rule fires)
  int it = 0;
  for (;; ++it) {
    if (!WITH_J && it >= n_stop) break;
    // forward: a = cumsum(w); b = k*a - y
    block_cumsum<false>(w, a, N, red);
    block_conv(a, N, b, N, k, K);
    for (int i = threadIdx.x; i < N; i += GEN_THREADS) b[i] -= (double)yr[i];
    __syncthreads();
    if constexpr (WITH_J) {
      if (it > 0) {
        double part = 0.0;
        for (int i = threadIdx.x; i < N; i += GEN_THREADS)
          part += 0.5 * b[i] * b[i] + lb * fabs(w[i]);
        const double cost = block_sum(part, red);
        if (threadIdx.x == 0) Jrow[it - 1] = (TJ)cost;
      }
      if (it >= n_stop) break;
    }
    // adjoint: a = K^T b ; a = revcumsum(a)
    block_corr(b, N, a, N, k, K);
    block_cumsum<true>(a, a, N, red);
    // update
    const double beta = a_.betas[it];
    const double nb1 = -(1.0 + beta);
    double num = 0.0, den = 0.0;
    for (int i = threadIdx.x; i < N; i += GEN_THREADS) {
      const double u = fma(-stp, a[i], w[i]);
      const double d = prox_excess_ref(u, th);
      const double wn = fma(nb1, d, u);
      w[i] = wn;
      if (stop == 1) {
        num = fma(wn - u, wn - u, num);
        den = fma(wn, wn, den);
      } else if (stop == 2) {
        // window of the last `wind` stored iterates; the slot that held w_k was
        // overwritten in place by u_k (pybold/bold_signal.py:65,72,82)
        if (it > 0) hist[((it - 1) % wind) * N + i] = u;
        hist[(it % wind) * N + i] = wn;
      }
    }
    __syncthreads();
    if (stop == 1) {
      num = block_sum(num, red);
      den = block_sum(den, red);
      if (it > 2 && sqrt(num) / (sqrt(den) + 1.0e-10) < a_.tol) n_stop = it + 1;
    } else if (stop == 2 && it > wind) {
      const int half = wind / 2;
      for (int i = threadIdx.x; i < N; i += GEN_THREADS) {
        double so = 0.0, sn = 0.0;
        for (int q = it - wind + 1; q <= it - half; ++q) so += hist[(q % wind) * N + i];
        for (int q = it - half + 1; q <= it; ++q) sn += hist[(q % wind) * N + i];
        so /= (double)(wind - half);
        sn /= (double)half;
        num = fma(sn - so, sn - so, num);
        den = fma(sn, sn, den);
      }
      num = block_sum(num, red);
      den = block_sum(den, red);
      if (sqrt(num) / (sqrt(den) + 1.0e-10) < a_.tol) n_stop = it + 1;
    }
  }
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) wrow[i] = w[i];
  if (a_.n_done && threadIdx.x == 0) a_.n_done[p] = min(n_stop, a_.n_iter);
  }
}

// ---- opt-in extra: the same recurrence with a BACKTRACKED step (north_star's "Lipschitz-backtracked step") ----------
// The reference has a constant step only (pybold/bold_signal.py:52-53, :253-254), so this mode is never part
// of a parity run; it is checked against its own float64 NumPy statement (tests/: `fista_backtrack_batch` of the checker).
// Beck & Teboulle's rule on the reference's recurrence: per iteration, with g = H^T(H w - y) at the extrapolated point w,
//   repeat:  u = w - s g;  p = soft(u, lbda s);  accept if  F(p) <= F(w) + <p - w, g> + ||p - w||^2 / (2 s)
//            (F = 0.5 ||H . - y||^2), else s <- eta s          (at most max_bt times per iteration; s never grows)
//   w <- p + beta_k (p - u)                                     (the reference's momentum on the gradient point)
// With s <= 1/L the test passes at once and the iterates are those of the constant-step kernel.
// One workgroup per problem, float64 end to end.  LDS: w[N] g[N] pt[N] a[N] b[N] k[K] red[8].
__global__ __launch_bounds__(GEN_THREADS) void fista_backtrack_kernel(FistaArgs a_, const double* taps, int K, double eta,
                                                                      int max_bt, double* step_out, int32_t* halvings_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a_.N;
  double* w = reinterpret_cast<double*>(smem);
  double* g = w + N;
  double* pt = g + N;
  double* a = pt + N;
  double* b = a + N;
  double* k = b + N;
  double* red = k + K;
  const int p = blockIdx.x + a_.p0;
  const double* yr = a_.y64 + (int64_t)(p / a_.y_rep) * a_.ldy;
  double* wrow = a_.w + (int64_t)p * a_.ldw;
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) w[i] = a_.cold ? 0.0 : wrow[i];
  for (int i = threadIdx.x; i < K; i += GEN_THREADS) k[i] = taps[i];
  __syncthreads();
  const double lb = a_.lbda_vec ? a_.lbda_vec[p] : a_.lbda;
  double s = a_.step;
  int halvings = 0;
  for (int it = 0; it < a_.n_iter; ++it) {
    // F(w) and g = H^T (H w - y)
    block_cumsum<false>(w, a, N, red);
    block_conv(a, N, b, N, k, K);
    double part = 0.0;
    for (int i = threadIdx.x; i < N; i += GEN_THREADS) {
      b[i] -= yr[i];
      part = fma(b[i], b[i], part);
    }
    const double f_w = 0.5 * block_sum(part, red);
    __syncthreads();
    block_corr(b, N, a, N, k, K);
    block_cumsum<true>(a, g, N, red);
    // trials
    for (int bt = 0;; ++bt) {
      const double th = lb * s;
      double dot = 0.0, nd = 0.0;
      for (int i = threadIdx.x; i < N; i += GEN_THREADS) {
        const double u = fma(-s, g[i], w[i]);
        const double pi = u - prox_excess_ref(u, th);
        pt[i] = pi;
        const double d = pi - w[i];
        dot = fma(d, g[i], dot);
        nd = fma(d, d, nd);
      }
      __syncthreads();
      dot = block_sum(dot, red);
      nd = block_sum(nd, red);
      block_cumsum<false>(pt, a, N, red);
      block_conv(a, N, b, N, k, K);
      double fp = 0.0;
      for (int i = threadIdx.x; i < N; i += GEN_THREADS) {
        const double r = b[i] - yr[i];
        fp = fma(r, r, fp);
      }
      fp = 0.5 * block_sum(fp, red);
      __syncthreads();
      if (fp <= f_w + dot + nd / (2.0 * s) || bt >= max_bt) break;      // (uniform: every thread holds the same sums)
      s *= eta;
      ++halvings;
    }
    const double beta = a_.betas[it];
    for (int i = threadIdx.x; i < N; i += GEN_THREADS) {
      const double u = fma(-s, g[i], w[i]);
      w[i] = fma(beta, pt[i] - u, pt[i]);
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < N; i += GEN_THREADS) wrow[i] = w[i];
  if (threadIdx.x == 0) {
    if (a_.n_done) a_.n_done[p] = a_.n_iter;
    if (step_out) step_out[p] = s;
    if (halvings_out) halvings_out[p] = halvings;
  }
}

}  // namespace pb
