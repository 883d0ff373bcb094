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

/* ---- deconv(lbda=None): noise-driven lambda search (pybold/bold_signal.py:99-214) --------
 * C form of oracle/pybold_oracle.py::deconv_auto_lbda (pinned like it since round 5: the
 * reference's own branch run with sigma injected, tests/golden/auto_lbda.npz).  Inner solve = the fixed-lambda recurrence WITHOUT cost trace,
 * t restarted, warm-started, windowed stop on the stored iterates [u_{k-wind+2} .. u_k,
 * w_{k+1}] (:125-138; the aliasing of :65/:72 is why all but the newest entry are gradient
 * points).  Outer loop: alpha += mu (||x - y||^2 - N sigma^2), lbda = 1/(2 alpha) (:141-145),
 * windowed stop on alpha (:164-178), then a last inner solve (:181-209). */
static int inner_fista(double* w, const double* y, const double* hty, const double* h, int N, int K,
                       double step, double th, int nb_sub_iter, int early_stopping, int wind,
                       double tol, double* scratch, double* hist) {
  double* z = scratch;
  double* x = scratch + N;
  double* g = scratch + 2 * N;
  double t_old = 1.0;
  const int half = wind / 2;
  int n_hist = 0, head = 0;                 /* ring: slot (head + i) % wind = i-th oldest */
  int j = 0;
  (void)y;
  for (j = 0; j < nb_sub_iter; ++j) {
    forward(w, h, N, K, z, x);
    adjoint(x, h, N, K, g);                 /* H^T H w */
    const double t = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t_old * t_old));
    const double beta = (t_old - 1.0) / t;
    double* newest = n_hist ? hist + (size_t)((head + n_hist - 1) % wind) * N : NULL;
    for (int i = 0; i < N; ++i) {
      const double u = w[i] - step * (g[i] - hty[i]);
      if (j > 0 && newest) newest[i] = u;   /* the stored alias of w_j was overwritten */
      const double a = fabs(u) - th;
      const double sgn = (u > 0.0) - (u < 0.0);
      const double p = sgn * (a > 0.0 ? a : 0.0);
      const double prev = (j > 0) ? u : 0.0;
      w[i] = p + beta * (p - prev);
    }
    t_old = t;
    if (n_hist == wind) { head = (head + 1) % wind; --n_hist; }
    memcpy(hist + (size_t)((head + n_hist) % wind) * N, w, sizeof(double) * N);
    ++n_hist;
    if (early_stopping && j > wind) {
      double num = 0.0, den = 0.0;
      for (int i = 0; i < N; ++i) {
        double so = 0.0, sn = 0.0;
        for (int q = 0; q < n_hist - half; ++q) so += hist[(size_t)((head + q) % wind) * N + i];
        for (int q = n_hist - half; q < n_hist; ++q) sn += hist[(size_t)((head + q) % wind) * N + i];
        so /= (double)(n_hist - half);
        sn /= (double)half;
        num += (sn - so) * (sn - so);
        den += sn * sn;
      }
      if (sqrt(num) / (sqrt(den) + 1.0e-10) < tol) { ++j; break; }
    }
  }
  return j;
}

/* One voxel.  J, R, G: nb_iter doubles each (raw values per outer iteration, :149-157).
 * Returns the number of outer iterations recorded; *inner_total = inner iterations run. */
int oracle_deconv_auto_lbda(const double* y, int N, const double* h, int K, double sigma,
                            double lipschitz, int early_stopping, double tol, int wind, int nb_iter,
                            int nb_sub_iter, double* w, double* J, double* R, double* G,
                            long long* inner_total) {
  double* scratch = (double*)malloc(sizeof(double) * (size_t)(4 + wind) * N);
  double* hty = scratch + 3 * N;
  double* hist = scratch + 4 * N;
  double* l_alpha = (double*)malloc(sizeof(double) * (size_t)(wind + 1));
  adjoint(y, h, N, K, hty);
  const double step = 1.0 / lipschitz, mu = 1.0e-4;
  double alpha = 1.0, lbda = 1.0 / (2.0 * alpha);
  int n_alpha = 0, n_out = 0;
  long long inner = 0;
  memset(w, 0, sizeof(double) * N);
  for (int i = 0; i < nb_iter; ++i) {
    inner += inner_fista(w, y, hty, h, N, K, step, lbda / lipschitz, nb_sub_iter, early_stopping, wind,
                         tol, scratch, hist);
    forward(w, h, N, K, scratch, scratch + N);
    double r = 0.0, g = 0.0;
    for (int q = 0; q < N; ++q) {
      const double d = scratch[N + q] - y[q];
      r += d * d;
      g += fabs(w[q]);
    }
    alpha += mu * (r - N * sigma * sigma);
    lbda = 1.0 / (2.0 * alpha);
    if (n_alpha == wind) { memmove(l_alpha, l_alpha + 1, sizeof(double) * (wind - 1)); --n_alpha; }
    l_alpha[n_alpha++] = alpha;
    R[n_out] = r; G[n_out] = g; J[n_out] = 0.5 * r + lbda * g; ++n_out;
    if (early_stopping && i > wind) {
      const int half = wind / 2;
      double so = 0.0, sn = 0.0;
      for (int q = 0; q < n_alpha - half; ++q) so += l_alpha[q];
      for (int q = n_alpha - half; q < n_alpha; ++q) sn += l_alpha[q];
      so /= (double)(n_alpha - half);
      sn /= (double)half;
      if (fabs(sn - so) / fabs(sn) < tol) break;
    }
  }
  inner += inner_fista(w, y, hty, h, N, K, step, lbda / lipschitz, nb_sub_iter, early_stopping, wind, tol,
                       scratch, hist);
  if (inner_total) *inner_total = inner;
  free(l_alpha);
  free(scratch);
  return n_out;
}

/* Batch over voxels (OpenMP); sigma (V), W (V, N), J/R/G (V, nb_iter), n_outer (V). */
void oracle_deconv_auto_lbda_batch(const double* Y, int V, int N, const double* h, int K,
                                   const double* sigma, double lipschitz, int early_stopping,
                                   double tol, int wind, int nb_iter, int nb_sub_iter, double* W,
                                   double* J, double* R, double* G, int* n_outer, int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
  for (int v = 0; v < V; ++v)
    n_outer[v] = oracle_deconv_auto_lbda(Y + (size_t)v * N, N, h, K, sigma[v], lipschitz, early_stopping,
                                         tol, wind, nb_iter, nb_sub_iter, W + (size_t)v * N,
                                         J + (size_t)v * nb_iter, R + (size_t)v * nb_iter,
                                         G + (size_t)v * nb_iter, NULL);
}
