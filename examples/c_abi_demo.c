/*
 * The C ABI from plain C, without Python or PyTorch: what a cgo / JNI / ctypes binding of
 * include/pybold_hip.h does.  Reads float32 series and float64 HRF taps from files, runs
 * pb_fista_solve on the default stream, writes the float64 iterates.
 *
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c \
 *       -o examples/c_abi_demo -Lpybold_amd -lpybold_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,'$ORIGIN/../pybold_amd' -Wl,-rpath,/opt/rocm/lib
 *   examples/c_abi_demo y.f32 V N taps.f64 K step lbda n_iter w_out.f64
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pybold_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static void* read_file(const char* path, size_t bytes) {
  FILE* f = fopen(path, "rb");
  void* p = malloc(bytes);
  if (!f || !p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
  fclose(f);
  return p;
}

int main(int argc, char** argv) {
  if (argc != 10) {
    fprintf(stderr, "usage: %s y.f32 V N taps.f64 K step lbda n_iter w_out.f64\n", argv[0]);
    return 2;
  }
  const int V = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[5]), n_iter = atoi(argv[8]);
  const double step = atof(argv[6]), lbda = atof(argv[7]);
  float* y = (float*)read_file(argv[1], (size_t)V * N * sizeof(float));
  double* taps = (double*)read_file(argv[4], (size_t)K * sizeof(double));

  /* momentum factors beta_k = (t_k - 1) / t_{k+1}, pybold/bold_signal.py:60,68-71 */
  double* betas = (double*)malloc((size_t)(n_iter > 0 ? n_iter : 1) * sizeof(double));
  double t_old = 1.0;
  for (int k = 0; k < n_iter; ++k) {
    const double t = 0.5 * (1.0 + sqrt(1.0 + 4.0 * t_old * t_old));
    betas[k] = (t_old - 1.0) / t;
    t_old = t;
  }

  float* y_dev; double *w_dev, *taps_dev, *betas_dev; int32_t* n_done_dev;
  HIP_OK(hipMalloc((void**)&y_dev, (size_t)V * N * sizeof(float)));
  HIP_OK(hipMalloc((void**)&w_dev, (size_t)V * N * sizeof(double)));
  HIP_OK(hipMalloc((void**)&taps_dev, (size_t)K * sizeof(double)));
  HIP_OK(hipMalloc((void**)&betas_dev, (size_t)(n_iter > 0 ? n_iter : 1) * sizeof(double)));
  HIP_OK(hipMalloc((void**)&n_done_dev, (size_t)V * sizeof(int32_t)));
  HIP_OK(hipMemcpy(y_dev, y, (size_t)V * N * sizeof(float), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(taps_dev, taps, (size_t)K * sizeof(double), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(betas_dev, betas, (size_t)n_iter * sizeof(double), hipMemcpyHostToDevice));

  const int rc = pb_fista_solve(y_dev, N, 1, w_dev, N, V, N, taps, taps_dev, K, step, lbda, NULL,
                                betas_dev, n_iter, NULL, 0, PB_STOP_NONE, 0.0, 0, n_done_dev,
                                PB_FLAG_COLD_START /* w_dev is output only */,
                                NULL /* default stream */);
  if (rc != PB_OK) {
    fprintf(stderr, "pb_fista_solve failed (%d): %s\n", rc, pb_last_error());
    return 1;
  }
  HIP_OK(hipDeviceSynchronize());

  double* w = (double*)malloc((size_t)V * N * sizeof(double));
  HIP_OK(hipMemcpy(w, w_dev, (size_t)V * N * sizeof(double), hipMemcpyDeviceToHost));
  FILE* f = fopen(argv[9], "wb");
  if (!f || fwrite(w, sizeof(double), (size_t)V * N, f) != (size_t)V * N) { fprintf(stderr, "cannot write %s\n", argv[9]); return 2; }
  fclose(f);
  double l1 = 0.0;
  for (size_t i = 0; i < (size_t)V * N; ++i) l1 += fabs(w[i]);
  printf("pybold_hip %d: %d voxels x %d scans, %d iterations, sum|diff_z| = %.12e\n", pb_version(), V, N, n_iter, l1);
  hipFree(y_dev); hipFree(w_dev); hipFree(taps_dev); hipFree(betas_dev); hipFree(n_done_dev);
  free(y); free(taps); free(betas); free(w);
  return 0;
}
