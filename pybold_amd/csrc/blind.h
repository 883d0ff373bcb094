// HRF-fit (theta-step) kernels of the blind solver, float64.
//
// The reference prices one candidate HRF at a time: hrf_fit_err(theta) =
// 0.5 || y - h(theta) * z ||^2 (pybold/bold_signal.py:217-222), called ~10-20 times per
// outer iteration by L-BFGS-B with a finite-difference gradient (:329-333), each call a
// full pass over the voxel's data.  The cost is a quadratic form in the K taps,
//
//   F(theta) = 0.5 yy - h^T b + 0.5 h^T G h,      h = h(theta)
//   G[m][m'] = sum_i z[i-m] z[i-m']   b[m] = sum_i z[i-m] y[i]   yy = sum_i y[i]^2
//
// so ONE pass over the data (normal_eq_kernel) gives (G, b, yy) -- summed over the voxels
// of the rank for the shared-HRF variant, then over ranks by one all-reduce of K^2+K+1
// float64 -- and the whole 1-D search over theta runs on those few numbers in a single
// wave (theta_fit_kernel), with no further pass over the data and no host round trip.
//
// G is Toeplitz up to the truncation at the end of the series:
//   G[m][m'] = R_d(N-1-max(m,m')),  d = |m-m'|,  R_d(T) = sum_{j<=T} z[j] z[j+d]
// so a lane owns one lag d, walks j once and records the running sum at the K-d
// truncation points: K lanes x N steps per voxel instead of K^2 x N.
#pragma once
#include "common.h"
#include "generic.h"

namespace pb {

constexpr int NE_THREADS = 256;

__host__ __device__ inline int ne_len(int K) { return K * K + K + 1; }

// Layout of one normal-equation set: G row-major [K][K], then b [K], then yy.
// One voxel at a time per workgroup; SUB adjacent lanes share one role (SUB = 4 for K <= 31):
//   role r <  K   autocorrelation lag r        (G diagonals)
//        r < 2K   cross-correlation lag r-K    (b)
//        r = 2K   yy
// The sums over j < jb = N-K have no truncation point inside them ("bulk"), so the SUB lanes
// of a role split them (stride SUB); only the last K samples ("tail") need the running sum
// at every truncation point.  LDS: z[N] y[N] (float64).

// One set per voxel: out[v][ne_len].  Lane 0 of a role walks the tail alone and stores the
// running sum at each truncation point.
template <typename TY>
__global__ __launch_bounds__(NE_THREADS) void normal_eq_kernel(const double* z, int64_t ldz,
                                                               const TY* y, int64_t ldy, int V,
                                                               int N, int K, int sub_log2,
                                                               double* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int SUB = 1 << sub_log2;
  const int ne = ne_len(K);
  const int role = threadIdx.x >> sub_log2;
  const int sub = threadIdx.x & (SUB - 1);
  const bool worker = role < 2 * K + 1;
  double* lz = reinterpret_cast<double*>(smem);
  double* ly = lz + N;
  const int jb = N > K ? N - K : 0;
  for (int v = blockIdx.x; v < V; v += gridDim.x) {
    __syncthreads();                           // previous voxel fully consumed
    {
      const double* zr = z + (int64_t)v * ldz;
      const TY* yr = y + (int64_t)v * ldy;
      for (int i = threadIdx.x; i < N; i += NE_THREADS) {
        lz[i] = zr[i];
        ly[i] = (double)yr[i];
      }
    }
    __syncthreads();
    double* dst = out + (int64_t)v * ne;
    double r = 0.0;
    if (worker) {
      if (role < K) {                          // autocorrelation, bulk part
        const int d = role;
        for (int j = sub; j < jb; j += SUB) r = fma(lz[j], lz[j + d], r);
      } else if (role < 2 * K) {               // cross-correlation lag m, all of it
        const int m = role - K;
        for (int j = sub; j < N - m; j += SUB) r = fma(lz[j], ly[j + m], r);
      } else {
        for (int i = sub; i < N; i += SUB) r = fma(ly[i], ly[i], r);
      }
    }
    for (int o = SUB >> 1; o >= 1; o >>= 1) r += __shfl_xor(r, o, 64);   // fixed order
    if (worker && sub == 0) {
      if (role < K) {
        const int d = role;
        // T = j; record at T = N-1-m' for m' = K-1 .. d  (j = N-K .. N-1-d)
        for (int j = jb; j < N - d; ++j) {
          r = fma(lz[j], lz[j + d], r);
          const int mp = N - 1 - j;            // m' = max(m, m') < K here, m = m' - d
          const int m = mp - d;
          dst[m * K + mp] = r;
          dst[mp * K + m] = r;
        }
        // series shorter than the HRF: truncation point before j = 0 -> empty sums
        for (int mp = (N > d ? N : d); mp < K; ++mp) {
          const int m = mp - d;
          dst[m * K + mp] = 0.0;
          dst[mp * K + m] = 0.0;
        }
      } else if (role < 2 * K) {
        dst[K * K + (role - K)] = r;
      } else {
        dst[K * K + K] = r;
      }
    }
  }
}

// Sum over the voxels of the block: part[blockIdx.x][ne_len] (a second kernel adds the blocks
// in a fixed order: deterministic).  Summed over voxels, the running sums become
//   G[m][m'] = BULK_d + sum_{j' <= j} P_d[j'],   P_d[j'] = sum_v z_v[jb+j'] z_v[jb+j'+d]
// so nothing in the voxel loop depends on the previous step.
//
// The bulk sums are register-blocked: a thread owns NE_LB consecutive lags of one kind
// (autocorrelation of z, or cross-correlation of z with y) and one slice of the sample axis;
// per block of NE_JB samples it reads NE_JB + (NE_JB + NE_LB - 1) values from LDS for
// NE_JB x NE_LB multiply-adds (0.36 reads per FMA; one lag per lane costs 2 and made the
// kernel LDS-bound: 0.26 ms for 50 k voxels, 4x the time of its HBM traffic).  Accumulators
// live in registers across the whole voxel loop; slices are folded once, in a fixed order.
// The tail products P (last K samples: one running sum per truncation point) and yy are dealt
// over the threads entry by entry.
// LDS: z[N + pad] y[N + K + pad] P[K*K] red[256 * NE_LB].
constexpr int NE_LB = 8, NE_JB = 8;

// LDS index of sample i: one pad every NE_JB samples, so that lanes whose blocks start NE_JB
// samples apart read 9 doubles apart (2-way bank conflicts instead of 16-way)
__host__ __device__ inline int ne_sk(int i) { return i + (i >> 3); }

__host__ __device__ inline int ne_sum_lds_doubles(int N, int K) {
  return ne_sk(N + NE_LB + NE_JB) + 1 + ne_sk(N + K + NE_LB + NE_JB) + 1 + K * K + 2 * NE_LB * 16 + 2 * K + 16;
}

// FROM_W: `z` holds the innovation w = diff_z; the block integrates it after staging (the
// summation tree of block_cumsum in generic.h: same z, bit for bit, as pb_integ_op gives) and
// also sums |w| -- part[ne] = ||w||_1 of the block's voxels, one entry more per block -- so
// that an outer iteration of the shared-HRF loop needs no pass of its own for either.
template <typename TY, bool FROM_W = false>
__global__ __launch_bounds__(NE_THREADS) void normal_eq_sum_kernel(const double* z, int64_t ldz,
                                                                   const TY* y, int64_t ldy, int V,
                                                                   int N, int K, double* part_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ne = ne_len(K);
  const int ne_out = ne + (FROM_W ? 1 : 0);
  double l1 = 0.0;
  static_assert(NE_JB == 8 && NE_LB == 8, "ne_sk() pads every 8 samples");
  const int nz = N + NE_LB + NE_JB, ny = N + K + NE_LB + NE_JB;      // logical lengths
  double* lz = reinterpret_cast<double*>(smem);
  double* ly = lz + ne_sk(nz) + 1;
  double* P = ly + ne_sk(ny) + 1;
  double* red = P + K * K;                     // [roles <= 32][NE_LB]
  double* bulk = red + 2 * NE_LB * 16;         // [2K]: autocorrelation lags, then cross lags
  double* red8 = bulk + 2 * K;                 // block_sum scratch
  const int t = threadIdx.x;
  for (int e = t; e < K * K; e += NE_THREADS) P[e] = 0.0;
  for (int i = t; i <= ne_sk(nz); i += NE_THREADS) lz[i] = 0.0;      // padding stays zero
  for (int i = t; i <= ne_sk(ny); i += NE_THREADS) ly[i] = 0.0;
  const int jb = N > K ? N - K : 0;
  // roles: lag blocks of the two kinds; slices: threads that share a role
  const int nlb = (K + NE_LB - 1) / NE_LB;
  int slices = 1;                              // threads per role: a power of two (shuffle fold)
  while (2 * slices * 2 * nlb <= NE_THREADS) slices *= 2;
  const int role = t / slices, slice = t % slices;
  const bool worker = role < 2 * nlb;
  const bool cross = role >= nlb;
  const int d0 = (cross ? role - nlb : role) * NE_LB;
  const int jrange = cross ? N : jb;           // autocorrelation: bulk part only
  const double* src = cross ? ly : lz;
  double acc[NE_LB];
#pragma unroll
  for (int l = 0; l < NE_LB; ++l) acc[l] = 0.0;
  double yy = 0.0;
  // this thread's tail entries (d, jj), fixed for the whole launch: LDS offsets, -1 = none
  constexpr int NE_TE = 4;                     // up to 4 * 256 entries: K <= 32; larger K loop below
  int te_a[NE_TE], te_b[NE_TE];
#pragma unroll
  for (int q = 0; q < NE_TE; ++q) {
    const int e = t + q * NE_THREADS;
    const int d = e / K, jj = e - d * K;
    const bool ok = e < K * K && jb + jj + d < N;
    te_a[q] = ok ? ne_sk(jb + jj) : -1;
    te_b[q] = ok ? ne_sk(jb + jj + d) : 0;
  }
  // samples of this thread for the voxel being staged (N <= 4 * 256: registers, prefetched
  // while the previous voxel is being processed; longer series are loaded in place)
  constexpr int NE_PF = 4;
  const bool prefetch = N <= NE_PF * NE_THREADS;
  double pz[NE_PF];
  TY py[NE_PF];
  auto fetch = [&](int v) {
    const double* zr = z + (int64_t)v * ldz;
    const TY* yr = y + (int64_t)v * ldy;
#pragma unroll
    for (int q = 0; q < NE_PF; ++q) {
      const int i = t + q * NE_THREADS;
      if (i < N) { pz[q] = zr[i]; py[q] = yr[i]; }
    }
  };
  if (prefetch && (int)blockIdx.x < V) fetch(blockIdx.x);
  for (int v = blockIdx.x; v < V; v += gridDim.x) {
    __syncthreads();                           // previous voxel fully consumed
    if (prefetch) {
#pragma unroll
      for (int q = 0; q < NE_PF; ++q) {
        const int i = t + q * NE_THREADS;
        if (i < N) { lz[ne_sk(i)] = pz[q]; ly[ne_sk(i)] = (double)py[q]; }
      }
      if (v + (int)gridDim.x < V) fetch(v + gridDim.x);            // in flight during the sums below
    } else {
      const double* zr = z + (int64_t)v * ldz;
      const TY* yr = y + (int64_t)v * ldy;
      for (int i = t; i < N; i += NE_THREADS) {
        lz[ne_sk(i)] = zr[i];
        ly[ne_sk(i)] = (double)yr[i];
      }
    }
    __syncthreads();
    if constexpr (FROM_W) {
      // z = cumsum(w) in place: per-thread chunk, wave scan, 4-wave combine (block_cumsum's tree)
      const int chunk = (N + NE_THREADS - 1) / NE_THREADS;
      const int lo = t * chunk, hi2 = min(lo + chunk, N);
      double local = 0.0;
      for (int i = lo; i < hi2; ++i) {
        const double wv = lz[ne_sk(i)];
        local += wv;
        l1 += fabs(wv);
      }
      double incl = local;
      const int lane = t & 63, wid = t >> 6;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      if (lane == 63) red8[wid] = incl;
      __syncthreads();
      double woff = 0.0;
      for (int i = 0; i < wid; ++i) woff += red8[i];
      double run = woff + incl - local;
      for (int i = lo; i < hi2; ++i) {
        run += lz[ne_sk(i)];
        lz[ne_sk(i)] = run;
      }
      __syncthreads();
    }
    if (worker) {
      for (int j0 = slice * NE_JB; j0 < jrange; j0 += slices * NE_JB) {
        double a[NE_JB], b[NE_JB + NE_LB - 1];
        const double* pa = lz + ne_sk(j0);
        const double* pb = src + ne_sk(j0 + d0);                   // j0, d0: multiples of 8
#pragma unroll
        for (int q = 0; q < NE_JB; ++q) a[q] = (j0 + q < jrange) ? pa[q] : 0.0;
#pragma unroll
        for (int q = 0; q < NE_JB + NE_LB - 1; ++q) b[q] = pb[q + (q >> 3)];
#pragma unroll
        for (int q = 0; q < NE_JB; ++q)
#pragma unroll
          for (int l = 0; l < NE_LB; ++l) acc[l] = fma(a[q], b[q + l], acc[l]);
      }
    }
    // tail products: entry (d, jj), jj < number of truncation points of lag d
#pragma unroll
    for (int q = 0; q < NE_TE; ++q)
      if (te_a[q] >= 0) P[t + q * NE_THREADS] = fma(lz[te_a[q]], lz[te_b[q]], P[t + q * NE_THREADS]);
    for (int e = t + NE_TE * NE_THREADS; e < K * K; e += NE_THREADS) {
      const int d = e / K, jj = e - d * K;
      if (jb + jj + d < N) P[e] = fma(lz[ne_sk(jb + jj)], lz[ne_sk(jb + jj + d)], P[e]);
    }
    for (int i = t; i < N; i += NE_THREADS) yy = fma(ly[ne_sk(i)], ly[ne_sk(i)], yy);
  }
  // fold the slices of every role (adjacent lanes; fixed order), then one value per lag
  for (int o = slices >> 1; o >= 1; o >>= 1) {
    if (o < 64) {
#pragma unroll
      for (int l = 0; l < NE_LB; ++l) acc[l] += __shfl_xor(acc[l], o, 64);
    }
  }
  __syncthreads();
  // (slices > 64: a role spans several waves -- only for K <= 8; their leaders are folded below)
  const int per_wave = slices < 64 ? slices : 64;
  if (worker && (t % per_wave) == 0) {
    const int part_i = (slices > 64) ? (t % slices) / 64 : 0;
    if (part_i < 2) {
#pragma unroll
      for (int l = 0; l < NE_LB; ++l) red[((part_i * 16 + role) * NE_LB) + l] = acc[l];
    }
  }
  __syncthreads();
  for (int e = t; e < 2 * K; e += NE_THREADS) {
    const bool cr = e >= K;
    const int lag = cr ? e - K : e;
    const int r = (cr ? nlb : 0) + lag / NE_LB, l = lag % NE_LB;
    double sum = red[(r * NE_LB) + l];
    if (slices > 64) sum += red[((16 + r) * NE_LB) + l];
    bulk[e] = sum;
  }
  yy = block_sum(yy, red8);                    // (syncs inside: bulk[] is visible afterwards)
  if constexpr (FROM_W) l1 = block_sum(l1, red8);
  double* part = part_out + (int64_t)blockIdx.x * ne_out;
  for (int e = t; e < ne; e += NE_THREADS) part[e] = 0.0;            // entries with no sample
  if (FROM_W && t == 0) part[ne] = l1;
  __syncthreads();
  if (t < K) {
    const int d = t;
    const int tail = (N - d > jb) ? N - d - jb : 0;
    double r = bulk[d];
    for (int jj = 0; jj < tail; ++jj) {
      r += P[d * K + jj];
      const int mp = N - 1 - (jb + jj);
      const int m = mp - d;
      part[m * K + mp] = r;
      part[mp * K + m] = r;
    }
  } else if (t < 2 * K) {
    part[K * K + (t - K)] = bulk[t];
  } else if (t == 2 * K) {
    part[K * K + K] = yy;
  }
}

// The same sums with ONE WAVE PER VOXEL (K <= 32): normal_eq_sum_kernel stages one voxel per
// workgroup behind three workgroup barriers for ~64 multiply-adds per thread -- barrier latency,
// not arithmetic, LDS or HBM, set its 0.39 ms per 100 k voxels (5x the time of its HBM traffic).
// Here every wave owns a voxel from staging to the last product: no workgroup barrier inside the
// voxel loop, next voxel's loads in flight during the sums, two waves per SIMD (190-240 registers)
// hiding each other's LDS latency: 0.25 ms per 100 k voxels (profiles/r3_config4_kernel_stats.csv).  Roles inside the wave: lane = (lag block of 8 lags, of z.z or z.y) x (slice of
// the sample axis), the register blocking of normal_eq_sum_kernel; tail products K^2 / 64 per lane
// in registers.  The four waves of a workgroup fold their sums once, at the end, in wave order;
// part[blockIdx.x] as before (normal_eq_reduce_kernel adds the workgroups in a fixed order).
// FROM_W: z = cumsum(w) inside the wave (chunk per lane, wave scan) + ||w||_1.
// LDS per wave: z[N + pad] y[N + K + pad]; per workgroup: fold[K^2 + 2K + 2].
// NEW_PE: tail entries per lane (K^2 <= 64 NEW_PE); NEW_PF: samples per lane prefetched in
// registers (N <= 64 NEW_PF; longer series are loaded in place).  The (12, 5) instance -- K <= 27,
// N <= 320: the shared-HRF loop of BASELINE config 4 -- is the leanest (forcing three waves per SIMD
// on it spills and measured slower: 0.40 ms).

__host__ __device__ inline int ne_wave_doubles(int N, int K) {      // one wave's staging area
  return ne_sk(N + NE_LB + NE_JB) + 1 + ne_sk(N + K + NE_LB + NE_JB) + 1;
}
__host__ __device__ inline int ne_wave_lds_doubles(int N, int K) {
  return GEN_WAVES * ne_wave_doubles(N, K) + K * K + 2 * K + 2;
}

template <typename TY, bool FROM_W, int NEW_PE = 16, int NEW_PF = 8>
__global__ __launch_bounds__(NE_THREADS) void normal_eq_wave_kernel(const double* z, int64_t ldz,
                                                                    const TY* y, int64_t ldy, int V,
                                                                    int N, int K, double* part_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ne = ne_len(K);
  const int ne_out = ne + (FROM_W ? 1 : 0);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int nz = N + NE_LB + NE_JB, ny = N + K + NE_LB + NE_JB;
  double* lz = reinterpret_cast<double*>(smem) + wv * ne_wave_doubles(N, K);
  double* ly = lz + ne_sk(nz) + 1;
  double* fold = reinterpret_cast<double*>(smem) + GEN_WAVES * ne_wave_doubles(N, K);   // [2K] bulk, [K^2] P, yy, l1
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  for (int i = lane; i <= ne_sk(nz); i += 64) lz[i] = 0.0;           // padding stays zero
  for (int i = lane; i <= ne_sk(ny); i += 64) ly[i] = 0.0;
  const int jb = N > K ? N - K : 0;
  const int nlb = (K + NE_LB - 1) / NE_LB;                           // <= 4
  int slices = 1;
  while (2 * slices * 2 * nlb <= 64) slices *= 2;
  const int role = lane / slices, slice = lane % slices;
  const bool worker = role < 2 * nlb;
  const bool cross = role >= nlb;
  const int d0 = (cross ? role - nlb : role) * NE_LB;
  const int jrange = cross ? N : jb;
  const double* src = cross ? ly : lz;
  double acc[NE_LB];
#pragma unroll
  for (int l = 0; l < NE_LB; ++l) acc[l] = 0.0;
  double yy = 0.0, l1 = 0.0;
  // this lane's tail entries (d, jj): LDS offsets of the two factors, 16 bits each; entries that do
  // not exist multiply the last padding slot (zero) by itself
  const unsigned zero_slot = (unsigned)ne_sk(nz) | ((unsigned)ne_sk(nz) << 16);
  unsigned te[NEW_PE];
  double P[NEW_PE];
#pragma unroll
  for (int q = 0; q < NEW_PE; ++q) {
    const int e = lane + q * 64;
    const int d = e / K, jj = e - d * K;
    const bool ok = e < K * K && jb + jj + d < N;
    te[q] = ok ? ((unsigned)ne_sk(jb + jj) | ((unsigned)ne_sk(jb + jj + d) << 16)) : zero_slot;
    P[q] = 0.0;
  }
  const bool prefetch = N <= NEW_PF * 64;
  double pz[NEW_PF];
  TY py[NEW_PF];
  auto fetch = [&](int v) {
    const double* zr = z + (int64_t)v * ldz;
    const TY* yr = y + (int64_t)v * ldy;
#pragma unroll
    for (int q = 0; q < NEW_PF; ++q) {
      const int i = lane + q * 64;
      if (i < N) { pz[q] = zr[i]; py[q] = yr[i]; }
    }
  };
  const int v0 = blockIdx.x * GEN_WAVES + wv, vstride = gridDim.x * GEN_WAVES;
  if (prefetch && v0 < V) fetch(v0);
  const int chunk = (N + NE_THREADS - 1) / NE_THREADS;
  for (int v = v0; v < V; v += vstride) {
    wave_sync();                                 // previous voxel fully consumed
    if (prefetch) {
#pragma unroll
      for (int q = 0; q < NEW_PF; ++q) {
        const int i = lane + q * 64;
        if (i < N) {
          const double yv = (double)py[q];
          lz[ne_sk(i)] = pz[q];
          ly[ne_sk(i)] = yv;
          yy = fma(yv, yv, yy);
        }
      }
      if (v + vstride < V) fetch(v + vstride);   // in flight during the sums below
    } else {
      const double* zr = z + (int64_t)v * ldz;
      const TY* yr = y + (int64_t)v * ldy;
      for (int i = lane; i < N; i += 64) {
        const double yv = (double)yr[i];
        lz[ne_sk(i)] = zr[i];
        ly[ne_sk(i)] = yv;
        yy = fma(yv, yv, yy);
      }
    }
    wave_sync();
    if constexpr (FROM_W) {
      // z = cumsum(w) with the summation tree of block_cumsum (generic.h) -- chunks of
      // ceil(N / 256), a 64-wide scan per group of 64 chunks, group offsets added in order -- so
      // that z equals pb_integ_op's bit for bit; this wave plays the four waves one after another
      // (the four scans are independent until the offsets are added: written side by side so
      // that their shuffle latencies overlap)
      constexpr int CH = (NEW_PF * 64 + NE_THREADS - 1) / NE_THREADS;   // chunk <= CH while the prefetch form applies
      double local[GEN_WAVES], incl[GEN_WAVES], x[GEN_WAVES][CH];
      const bool inreg = chunk <= CH;              // chunk elements in registers: ONE LDS round trip
#pragma unroll
      for (int g4 = 0; g4 < GEN_WAVES; ++g4) {
        const int lo = (g4 * 64 + lane) * chunk, hi2 = min(lo + chunk, N);
        local[g4] = 0.0;
        if (inreg) {
#pragma unroll
          for (int c = 0; c < CH; ++c) x[g4][c] = (c < chunk && lo + c < N) ? lz[ne_sk(lo + c)] : 0.0;
        } else {
          for (int i = lo; i < hi2; ++i) {
            const double wvv = lz[ne_sk(i)];
            local[g4] += wvv;
            l1 += fabs(wvv);
          }
        }
      }
      if (inreg) {
#pragma unroll
        for (int g4 = 0; g4 < GEN_WAVES; ++g4)
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            local[g4] += x[g4][c];                 // (an absent element adds +0.0)
            l1 += fabs(x[g4][c]);
          }
      }
#pragma unroll
      for (int g4 = 0; g4 < GEN_WAVES; ++g4) incl[g4] = local[g4];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int g4 = 0; g4 < GEN_WAVES; ++g4) {
          const double up = __shfl_up(incl[g4], o, 64);
          if (lane >= o) incl[g4] += up;
        }
      }
      double woff = 0.0;
#pragma unroll
      for (int g4 = 0; g4 < GEN_WAVES; ++g4) {
        const int lo = (g4 * 64 + lane) * chunk, hi2 = min(lo + chunk, N);
        double run = woff + incl[g4] - local[g4];
        if (inreg) {
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            run += x[g4][c];
            if (c < chunk && lo + c < N) lz[ne_sk(lo + c)] = run;
          }
        } else {
          for (int i = lo; i < hi2; ++i) {
            run += lz[ne_sk(i)];
            lz[ne_sk(i)] = run;
          }
        }
        woff += __shfl(incl[g4], 63, 64);
      }
      wave_sync();
    }
    if (worker) {
      for (int j0 = slice * NE_JB; j0 < jrange; j0 += slices * NE_JB) {
        double a[NE_JB], b[NE_JB + NE_LB - 1];
        const double* pa = lz + ne_sk(j0);
        const double* pb = src + ne_sk(j0 + d0);
#pragma unroll
        for (int q = 0; q < NE_JB; ++q) a[q] = (j0 + q < jrange) ? pa[q] : 0.0;
#pragma unroll
        for (int q = 0; q < NE_JB + NE_LB - 1; ++q) b[q] = pb[q + (q >> 3)];
#pragma unroll
        for (int q = 0; q < NE_JB; ++q)
#pragma unroll
          for (int l = 0; l < NE_LB; ++l) acc[l] = fma(a[q], b[q + l], acc[l]);
      }
    }
#pragma unroll
    for (int q = 0; q < NEW_PE; ++q)
      P[q] = fma(lz[te[q] & 0xffffu], lz[te[q] >> 16], P[q]);
  }
  // ---- fold: slices of a role (adjacent lanes, fixed order), then the four waves in wave order ----
  for (int o = slices >> 1; o >= 1; o >>= 1) {
#pragma unroll
    for (int l = 0; l < NE_LB; ++l) acc[l] += __shfl_xor(acc[l], o, 64);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    yy += __shfl_xor(yy, o, 64);
    l1 += __shfl_xor(l1, o, 64);
  }
  for (int w = 0; w < GEN_WAVES; ++w) {
    __syncthreads();
    if (wv == w) {
      if (worker && slice == 0) {
#pragma unroll
        for (int l = 0; l < NE_LB; ++l)
          if (d0 + l < K) {
            double* dst = fold + (cross ? K : 0) + d0 + l;
            *dst = (w == 0 ? 0.0 : *dst) + acc[l];
          }
      }
#pragma unroll
      for (int q = 0; q < NEW_PE; ++q) {
        const int e = lane + q * 64;
        if (e < K * K) fold[2 * K + e] = (w == 0 ? 0.0 : fold[2 * K + e]) + P[q];
      }
      if (lane == 0) {
        fold[2 * K + K * K] = (w == 0 ? 0.0 : fold[2 * K + K * K]) + yy;
        fold[2 * K + K * K + 1] = (w == 0 ? 0.0 : fold[2 * K + K * K + 1]) + l1;
      }
    }
  }
  __syncthreads();
  const double* bulk = fold;
  const double* Pf = fold + 2 * K;
  double* part = part_out + (int64_t)blockIdx.x * ne_out;
  for (int e = t; e < ne; e += NE_THREADS) part[e] = 0.0;            // entries with no sample
  if (FROM_W && t == 0) part[ne] = fold[2 * K + K * K + 1];
  __syncthreads();
  if (t < K) {
    const int d = t;
    const int tail = (N - d > jb) ? N - d - jb : 0;
    double r = bulk[d];
    for (int jj = 0; jj < tail; ++jj) {
      r += Pf[d * K + jj];
      const int mp = N - 1 - (jb + jj);
      const int m = mp - d;
      part[m * K + mp] = r;
      part[mp * K + m] = r;
    }
  } else if (t < 2 * K) {
    part[K * K + (t - K)] = bulk[t];
  } else if (t == 2 * K) {
    part[K * K + K] = fold[2 * K + K * K];
  }
}

// out[e] = sum_b part[b][e]: one workgroup per entry e, the blocks dealt over its threads
// (stride NE_THREADS) and folded by the fixed tree of block_sum -- the summation order
// depends on nblocks only, never on timing (nblocks may be 0: an empty shard gives zeros).
__global__ __launch_bounds__(NE_THREADS) void normal_eq_reduce_kernel(const double* part, int nblocks,
                                                                      int ne, double* out) {
  __shared__ double red[2 * GEN_WAVES];
  const int e = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += NE_THREADS) s += part[(int64_t)b * ne + e];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[e] = s;
}

struct HrfModel {        // two-gamma SPM HRF, un-normalised (pybold/hrf_model.py:25-31)
  double a1, loc1, lg1, a2, loc2, lg2, ratio;
  int n1, n2;            // a - 1 when that is a small non-negative integer (the default model:
                         // 5 and 15), else -1
  double c1, c2;         // 1 / Gamma(a)
};

inline int hrf_int_power(double a) {
  const double e = a - 1.0;
  return (e >= 0.0 && e <= 64.0 && e == (double)(int)e) ? (int)e : -1;
}

// gamma density at v = x - loc: v^(a-1) exp(-v) / Gamma(a).  Integer a - 1 (uniform, a scalar
// branch): the power by repeated squaring instead of exp(. log v) -- half the cost of the
// density, which is nine tenths of the theta-fit's instructions -- and more accurate.
__device__ __forceinline__ double gamma_pdf(double v, double a, double lg, int n) {
  if (!(v > 0.0)) return 0.0;
  if (n >= 0) {
    double p = 1.0, b = v;
    for (int e = n; e > 0; e >>= 1) {
      if (e & 1) p *= b;
      b *= b;
    }
    return p * exp(-v - lg);
  }
  return exp((a - 1.0) * log(v) - v - lg);
}

__device__ __forceinline__ double ipow(double b, int n) {
  double p = 1.0;
  for (int e = n; e > 0; e >>= 1) {
    if (e & 1) p *= b;
    b *= b;
  }
  return p;
}

__device__ __forceinline__ double spm_hrf_value(const HrfModel& hm, double x) {
  if (hm.n1 >= 0 && hm.n2 >= 0 && hm.loc1 == hm.loc2) {
    // the default model: both densities at the same v with integer powers -- one exp
    const double v = x - hm.loc1;
    if (!(v > 0.0)) return 0.0;
    return exp(-v) * (ipow(v, hm.n1) * hm.c1 - hm.ratio * hm.c2 * ipow(v, hm.n2));
  }
  return gamma_pdf(x - hm.loc1, hm.a1, hm.lg1, hm.n1) - hm.ratio * gamma_pdf(x - hm.loc2, hm.a2, hm.lg2, hm.n2);
}

// argmin_theta F(theta) over [lo, hi] for M independent normal-equation sets, one
// 256-thread workgroup per set: a first scan of 64 candidate dilations (lane = candidate, the K
// taps of each candidate and the K rows of the quadratic form dealt over the 4 waves), then
// scans of 16 candidates around the best one (thread = candidate x one of 16 shares).
// Section search, the bracket shrinks to the two grid cells around the best candidate (x31.5
// per refinement); the last refinement ends with the vertex of the parabola through the best
// candidate and its neighbours.  Every wave reduces the same 64 values in the same order, so
// the four agree on the bracket without exchanging it.
// LDS: set [ne], h [K][64] (lane-major columns, conflict-free), part [4][64].
// theta[s], cost[s] = F(theta*), taps[s][K] = h(theta*).
__global__ __launch_bounds__(256) void theta_fit_kernel(const double* ne_sets, int64_t ldne, int M,
                                                        int K, const double* t, HrfModel hm,
                                                        double lo, double hi, int n_refine,
                                                        double* theta, double* cost, double* taps,
                                                        int64_t ldt, int n_scans, double* step_out,
                                                        double lbda, double* jcost_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int s = blockIdx.x;
  if (s >= M) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ne = ne_len(K);
  double* G = reinterpret_cast<double*>(smem);
  double* b = G + K * K;
  double* hl = G + ne + lane;                 // h[k] of candidate `lane` at hl[k * 64]
  double* part = G + ne + 64 * K;             // [4][64]
  const double* src = ne_sets + (int64_t)s * ldne;
  for (int e = threadIdx.x; e < ne; e += 256) G[e] = src[e];
  __syncthreads();
  const double yy = b[K];

  auto price = [&](double th) -> double {
    for (int k = wv; k < K; k += 4) hl[k * 64] = spm_hrf_value(hm, th * t[k]);
    __syncthreads();
    double quad = 0.0, lin = 0.0;
    for (int m = wv; m < K; m += 4) {
      double row = 0.0;
      for (int mp = 0; mp < K; ++mp) row = fma(G[m * K + mp], hl[mp * 64], row);
      quad = fma(hl[m * 64], row, quad);
      lin = fma(hl[m * 64], b[m], lin);
    }
    part[wv * 64 + lane] = 0.5 * quad - lin;
    __syncthreads();
    const double f = 0.5 * yy + ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]));
    __syncthreads();                           // hl / part are rewritten by the next call
    return f;
  };

  double a = lo, c = hi, best_t = lo;
  {
    // first scan: 64 candidates over the whole interval (lane = candidate)
    const double th = a + (c - a) * ((double)lane / 63.0);
    const double f = price(th);
    // argmin over the wave (first minimum wins: deterministic)
    double fm = f;
    int im = lane;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const double fo = __shfl_xor(fm, o, 64);
      const int io = __shfl_xor(im, o, 64);
      if (fo < fm || (fo == fm && io < im)) { fm = fo; im = io; }
    }
    const int il = im > 0 ? im - 1 : 0, ir = im < 63 ? im + 1 : 63;
    const double tl = __shfl(th, il, 64), tm = __shfl(th, im, 64), tr = __shfl(th, ir, 64);
    const double fl = __shfl(f, il, 64), fr = __shfl(f, ir, 64);
    best_t = tm;
    if (n_refine == 1 && im > 0 && im < 63) {
      const double den = fl - 2.0 * fm + fr;
      if (den > 0.0) best_t = fmin(fmax(tm + 0.5 * (tm - tl) * (fl - fr) / den, tl), tr);
    }
    a = tl;
    c = tr;
  }
  // every further level: two scans of 16 candidates over the two cells around the best one
  // (bracket / 7.5 each, / 56 per level).  Thread = (candidate, 1 of 16 shares of the taps and
  // rows): a sixteenth of the gamma densities per thread instead of a quarter.
  const int cand = threadIdx.x >> 4, sh = threadIdx.x & 15;
  double* h16 = G + ne;                        // [K][16]
  for (int r = 0; r < 2 * (n_refine - 1); ++r) {
    const double th = a + (c - a) * ((double)cand / 15.0);
    for (int k = sh; k < K; k += 16) h16[k * 16 + cand] = spm_hrf_value(hm, th * t[k]);
    __syncthreads();
    double v = 0.0;
    for (int m = sh; m < K; m += 16) {
      double row = 0.0;
      for (int mp = 0; mp < K; ++mp) row = fma(G[m * K + mp], h16[mp * 16 + cand], row);
      v = fma(h16[m * 16 + cand], 0.5 * row - b[m], v);
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);     // the 16 shares of one candidate
    if (sh == 0) part[cand] = 0.5 * yy + v;
    __syncthreads();
    int im = 0;
    double fm = part[0];
    for (int q = 1; q < 16; ++q) {
      const double fq = part[q];
      if (fq < fm) { fm = fq; im = q; }
    }
    const int il = im > 0 ? im - 1 : 0, ir = im < 15 ? im + 1 : 15;
    const double step16 = (c - a) / 15.0;
    const double tl = a + step16 * il, tm = a + step16 * im, tr = a + step16 * ir;
    const double fl = part[il], fr = part[ir];
    __syncthreads();                           // h16 / part are rewritten by the next scan
    best_t = tm;
    if (r == 2 * (n_refine - 1) - 1 && im > 0 && im < 15) {
      // vertex of the parabola through (tl, fl), (tm, fm), (tr, fr); equal spacing
      const double den = fl - 2.0 * fm + fr;
      if (den > 0.0) best_t = fmin(fmax(tm + 0.5 * step16 * (fl - fr) / den, tl), tr);
    }
    a = tl;
    c = tr;
  }
  // cost and taps at the returned dilation: ONE candidate, thread k = tap / row k
  double* h1 = G + ne;                         // [K]
  double* v1 = h1 + K;                         // [K]
  for (int k = threadIdx.x; k < K; k += 256) h1[k] = spm_hrf_value(hm, best_t * t[k]);
  __syncthreads();
  for (int m = threadIdx.x; m < K; m += 256) {
    double row = 0.0;
    for (int mp = 0; mp < K; ++mp) row = fma(G[m * K + mp], h1[mp], row);
    v1[m] = h1[m] * (0.5 * row - b[m]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double f = 0.0;
    for (int m = 0; m < K; ++m) f += v1[m];    // fixed order
    theta[s] = best_t;
    cost[s] = 0.5 * yy + f;
  }
  if (taps) {
    double* trow = taps + (int64_t)s * ldt;
    for (int k = threadIdx.x; k < K; k += 256) trow[k] = h1[k];
  }
  // what the next z-step needs beside the taps: its step 1 / ||A^T A||_F for the new HRF
  // (pybold/bold_signal.py:249-254; closed form of gram_frobenius_fir_wave, first wave), and the
  // normalised cost of this outer iteration (2 F + lbda ||w||_1) / ||y||^2 (:337-342), ||w||_1
  // being the entry behind the normal equations (pb_hrf_normal_eq_w)
  if (step_out && n_scans > 0 && wv == 0) {
    const double fro = gram_frobenius_fir_wave(h1, K, n_scans, v1 + K, lane);
    if (lane == 0) step_out[s] = 1.0 / fro;
  }
  if (jcost_out && threadIdx.x == 0) {
    double f = 0.0;
    for (int m = 0; m < K; ++m) f += v1[m];
    jcost_out[s] = (2.0 * (0.5 * yy + f) + lbda * src[ne]) / yy;
  }
}

}  // namespace pb
