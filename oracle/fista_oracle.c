/*
 * CPU oracle, C form: float64 restatement of pyBOLD's fixed-step FISTA-like
 * deconvolution loop for a batch of voxels.  TEST INFRASTRUCTURE ONLY (checker
 * for tests/ and the cpu_baseline leg of bench.py; never linked into the
 * product).  Parity: pinned -- tests/test_oracle_golden.py compares it with the
 * golden vectors captured from the reference.
 *
 * Reference (file:line in hcherkaoui/pybold):
 *   recurrence           pybold/bold_signal.py:62-72   (deconv, fixed lambda)
 *                        pybold/bold_signal.py:259-276 (_loops_deconv)
 *   H.op  = K cumsum     pybold/linear.py:73-93, :15-28
 *   H.adj = revcumsum K' pybold/linear.py:95-113, :30-43
 *   K[i,j] = k[i-j]      pybold/convolution.py:105-132
 *
 *   u = w - s H'(H w - y);  p = soft(u, lbda s);  w = p + beta_k (p - prev)
 *   prev = 0 for k = 0 and prev = u for k >= 1  (in-place update at :65 plus
 *   the alias at :72), t_0 = 1, t_{k+1} = (1 + sqrt(1 + 4 t_k^2)) / 2,
 *   beta_k = (t_k - 1) / t_{k+1}.
 *
 * The convolution is matrix-free (the reference multiplies by the dense
 * Toeplitz matrix): same numbers up to summation order, ~N/K times fewer flops,
 * i.e. a CPU baseline that is *faster* than the reference's own formulation.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void forward(const double* w, const double* h, int N, int K, double* z, double* x) {
  double s = 0.0;
  for (int i = 0; i < N; ++i) { s += w[i]; z[i] = s; }
  for (int i = 0; i < N; ++i) {
    const int m1 = i < K - 1 ? i : K - 1;
    double acc = 0.0;
    for (int m = 0; m <= m1; ++m) acc += h[m] * z[i - m];
    x[i] = acc;
  }
}

static void adjoint(const double* r, const double* h, int N, int K, double* g) {
  for (int j = 0; j < N; ++j) {
    const int m1 = (N - 1 - j) < K - 1 ? (N - 1 - j) : K - 1;
    double acc = 0.0;
    for (int m = 0; m <= m1; ++m) acc += h[m] * r[j + m];
    g[j] = acc;
  }
  double s = 0.0;
  for (int j = N - 1; j >= 0; --j) { s += g[j]; g[j] = s; }
}

/* One voxel, n_iter iterations from warm start w (in/out).  J (n_iter) optional:
 * 0.5||H w_{k+1} - y||^2 + lbda ||w_{k+1}||_1 (bold_signal.py:74-77). */
void oracle_fista_voxel(const double* y, const double* h, int N, int K, double lbda, double step,
                        int n_iter, double* w, double* J, double* scratch) {
  double* z = scratch;
  double* x = scratch + N;
  double* g = scratch + 2 * N;
  const double th = lbda * step;
  double t_old = 1.0;
  for (int k = 0; k < n_iter; ++k) {
    forward(w, h, N, K, z, x);
    for (int i = 0; i < N; ++i) x[i] -= y[i];
    adjoint(x, h, N, K, g);
    const double t = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t_old * t_old));
    const double beta = (t_old - 1.0) / t;
    for (int i = 0; i < N; ++i) {
      const double u = w[i] - step * g[i];
      const double a = fabs(u) - th;
      const double sgn = (u > 0.0) - (u < 0.0);
      const double p = sgn * (a > 0.0 ? a : 0.0);
      const double prev = (k > 0) ? u : 0.0;
      w[i] = p + beta * (p - prev);
    }
    t_old = t;
    if (J) {
      forward(w, h, N, K, z, x);
      double sq = 0.0, l1 = 0.0;
      for (int i = 0; i < N; ++i) { const double d = x[i] - y[i]; sq += d * d; l1 += fabs(w[i]); }
      J[k] = 0.5 * sq + lbda * l1;
    }
  }
}

/* Batch: Y, W are (V, N) row-major; lbda_vec (V) may be NULL.  Voxels are spread
 * over `threads` OpenMP threads (<= 0: runtime default).  Returns threads used. */
int oracle_fista_batch(const double* Y, int V, int N, const double* h, int K, double lbda,
                       const double* lbda_vec, double step, int n_iter, double* W, double* J,
                       int threads) {
  int used = 1;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
    double* scratch = (double*)malloc(sizeof(double) * 3 * (size_t)N);
#pragma omp for schedule(dynamic, 4)
    for (int v = 0; v < V; ++v)
      oracle_fista_voxel(Y + (size_t)v * N, h, N, K, lbda_vec ? lbda_vec[v] : lbda, step, n_iter,
                         W + (size_t)v * N, J ? J + (size_t)v * n_iter : NULL, scratch);
    free(scratch);
  }
#else
  double* scratch = (double*)malloc(sizeof(double) * 3 * (size_t)N);
  for (int v = 0; v < V; ++v)
    oracle_fista_voxel(Y + (size_t)v * N, h, N, K, lbda_vec ? lbda_vec[v] : lbda, step, n_iter,
                       W + (size_t)v * N, J ? J + (size_t)v * n_iter : NULL, scratch);
  free(scratch);
#endif
  return used;
}
