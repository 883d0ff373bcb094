// Round 5: what does each vector instruction of fista_mfma_kernel cost when ONE wave has the SIMD to itself (the
// kernel's situation: 256 + 167 registers), and with two?  8 independent chains per type, 2048 trips x 8 instructions.
//   hipcc -O3 --offload-arch=gfx950 valu_one_wave.hip -o valu_one_wave && ./valu_one_wave
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 16384;      // (long launches, and a warm-up before the first measurement: short ones run at idle clocks)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  double d[8];
  float f[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { d[i] = a + i; f[i] = b + i; u[i] = threadIdx.x + i; }
  const double da = a, db = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (KIND == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(da), "v"(db));
      if constexpr (KIND == 1) asm volatile("v_max_f64 %0, %1, %0" : "+v"(d[i]) : "v"(da));
      if constexpr (KIND == 2) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
      if constexpr (KIND == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
      if constexpr (KIND == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(a), "v"(b));
      if constexpr (KIND == 5) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]));
      if constexpr (KIND == 6) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(f[i]) : "v"(u[i]), "v"(f[(i + 1) & 7]));
      if constexpr (KIND == 7) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(f[i]), "v"(f[(i + 1) & 7]));
      if constexpr (KIND == 8) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(f[i]));
      if constexpr (KIND == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d[i]) : "v"(da), "v"(db));
      if constexpr (KIND == 10) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(da));
      if constexpr (KIND == 11) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += (float)d[i] + f[i] + (float)u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
float run(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 12; ++w) launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}
int main() {
  float* out;
  if (hipMalloc(&out, 256 * 4 * 2 * 256 * sizeof(float) * 4) != hipSuccess) return 1;
  for (int wps = 1; wps <= 2; ++wps) {
    printf("%d wave(s) per SIMD: ns per instruction of one wave (x %d = SIMD throughput)\n", wps, wps);
#define T(KIND, label) { float ms = run([&] { hipLaunchKernelGGL((k<KIND>), dim3(256 * wps), dim3(256), 0, 0, out, 1.0f, 0.5f, ITERS); }); \
    printf("  %-28s %7.3f ns\n", label, ms * 1e6 / (ITERS * 8.0)); }
    T(0, "v_fma_f64"); T(1, "v_max_f64"); T(10, "v_add_f64"); T(2, "v_cvt_f32_f64"); T(3, "v_cvt_f64_f32"); T(4, "v_fma_f32");
    T(9, "v_pk_fma_f32"); T(5, "v_cvt_pkrtz_f16_f32"); T(6, "v_fma_mix_f32"); T(7, "v_cvt_pk_f16_f32"); T(8, "v_accvgpr_read_b32"); T(11, "v_perm_b32");
  }
  return 0;
}
