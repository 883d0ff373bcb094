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
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pb {

constexpr int PATH_THREADS = 256, PATH_PER_THREAD = 16, PATH_PER_BLOCK = PATH_THREADS * PATH_PER_THREAD;

__device__ __forceinline__ bool path_is_dense(const double* lbda, const double* lmax, int y_rep, double ratio, int p) {
  return lbda[p] < ratio * lmax[p / y_rep];        // (lambda_max = 0, an all-zero series: sparse -- its solution is 0)
}

// dense problems of every block of 4 096
__global__ __launch_bounds__(PATH_THREADS) void path_count_kernel(const double* lbda, const double* lmax, int y_rep,
                                                                   double ratio, int P, int32_t* work) {
  __shared__ int part[PATH_THREADS / 64];
  const int p0 = blockIdx.x * PATH_PER_BLOCK + threadIdx.x * PATH_PER_THREAD;
  int c = 0;
  for (int i = 0; i < PATH_PER_THREAD; ++i) c += (p0 + i < P && path_is_dense(lbda, lmax, y_rep, ratio, p0 + i)) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) work[P + 1 + blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive prefix of the block counts (in place) and the total -- one workgroup, blocks dealt in chunks of 256
__global__ __launch_bounds__(PATH_THREADS) void path_scan_kernel(int P, int nblk, int32_t* work) {
  __shared__ int buf[PATH_THREADS];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  int32_t* cnt = work + P + 1;
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
  if (threadIdx.x == 0) work[P] = carry;
}

// the two lists
__global__ __launch_bounds__(PATH_THREADS) void path_scatter_kernel(const double* lbda, const double* lmax, int y_rep,
                                                                     double ratio, int P, int32_t* work) {
  __shared__ int buf[PATH_THREADS];
  const int p0 = blockIdx.x * PATH_PER_BLOCK + threadIdx.x * PATH_PER_THREAD;
  unsigned mask = 0;
  int c = 0;
  for (int i = 0; i < PATH_PER_THREAD; ++i)
    if (p0 + i < P && path_is_dense(lbda, lmax, y_rep, ratio, p0 + i)) { mask |= 1u << i; ++c; }
  buf[threadIdx.x] = c;
  __syncthreads();
  for (int o = 1; o < PATH_THREADS; o <<= 1) {
    const int add = threadIdx.x >= (unsigned)o ? buf[threadIdx.x - o] : 0;
    __syncthreads();
    buf[threadIdx.x] += add;
    __syncthreads();
  }
  int d = work[P + 1 + blockIdx.x] + buf[threadIdx.x] - c;       // dense problems before this thread's first one
  for (int i = 0; i < PATH_PER_THREAD; ++i) {
    const int p = p0 + i;
    if (p >= P) break;
    if (mask & (1u << i)) work[d++] = p;
    else work[P - 1 - (p - d)] = p;                // p - d = sparse problems before p
  }
}

}  // namespace pb
