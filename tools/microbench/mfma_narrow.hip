// Round 5: what does v_mfma_f32_16x16x16_f16 cost on gfx950 next to v_mfma_f32_16x16x32_f16 --
// alone (pipe cycles) and between the vector mix of fista_mfma_kernel (issue-port hold)?  If the narrow instruction ran
// in <= 0.6 of the wide one's time, the block products of the kernel could skip the empty quarter of the diagonal tile
// and the constant quarter of the previous-block tile (6 narrow products instead of 4 wide per split product).
// Also: v_mfma_f32_32x32x16_f16 and the 4x4x4 (64 blocks) form, for completeness.
//   hipcc -O3 --offload-arch=gfx950 mfma_narrow.hip -o mfma_narrow && ./mfma_narrow
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITERS = 16384;      // (long launches, and a warm-up before the first measurement: short ones run at idle clocks)
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));

// MK: 1 = 16x16x32 f16 (8 halves per lane), 2 = 16x16x16 f16 (4 halves per lane), 3 = 32x32x8 f16 (4 halves), 4 = 32x32x16 f16 (8 halves)
// NV vector instructions (the kernel's mix) after every matrix instruction; NCH independent accumulator chains
template <int MK, int NV, int NCH>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  h8v ha8, hb8;
  h4v ha4, hb4;
  for (int i = 0; i < 8; ++i) { ha8[i] = (_Float16)(a + i); hb8[i] = (_Float16)(b * i); }
  for (int i = 0; i < 4; ++i) { ha4[i] = (_Float16)(a + i); hb4[i] = (_Float16)(b * i); }
  f4v c4[8] = {};
  f16v c16[2] = {};
  double dac[4] = {a, b, a + b, a - b};
  float cf[4] = {a, b, a + b, a - b};
  const double da = a, db = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      // (inline asm on VGPR operands: the builtins put C/D into the accumulator file and shuffle it around the loop)
      if constexpr (MK == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c4[g % NCH]) : "v"(ha8), "v"(hb8));
      if constexpr (MK == 2) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(c4[g % NCH]) : "v"(ha4), "v"(hb4));
      if constexpr (MK == 3) asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(c16[g % 2]) : "v"(ha4), "v"(hb4));
      if constexpr (MK == 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c16[g % 2]) : "v"(ha8), "v"(hb8));
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int i = (g * NV + u) & 3, kind = (g * NV + u) % 5;
        if (kind == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dac[i]) : "v"(da), "v"(db));
        if (kind == 1) asm volatile("v_max_f64 %0, %1, %0" : "+v"(dac[i]) : "v"(da));
        if (kind == 2) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(cf[i]) : "v"(dac[i]));
        if (kind == 3) { int pk; asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(cf[i]), "v"(cf[(i + 1) & 3])); asm volatile("" :: "v"(pk)); }
        if (kind == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(cf[i]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += c4[i][0] + c4[i][3];
  for (int i = 0; i < 4; ++i) s += (float)dac[i] + cf[i];
  for (int i = 0; i < 16; ++i) s += c16[0][i] + c16[1][i];
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
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  float* out;
  if (hipMalloc(&out, 256 * 4 * 2 * 256 * sizeof(float) * 4) != hipSuccess) return 1;
  int clk = 0;
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  for (int wps = 1; wps <= 2; ++wps) {
  printf("%d wave(s) per SIMD, %d trips x 8 matrix instructions; ms per launch (best of 5) and ns per matrix instruction [+ its vector instructions] and wave\n", wps, ITERS);
#define T(MK, NV, NCH, label) { float ms = run([&] { hipLaunchKernelGGL((k<MK, NV, NCH>), dim3(256 * wps), dim3(256), 0, 0, out, 1.0f, 0.5f, ITERS); }); \
    printf("  %-72s %8.3f ms  %7.2f ns\n", label, ms, ms * 1e6 / (ITERS * 8.0 * wps)); }
  T(1, 0, 4, "16x16x32 f16, 4 chains");
  T(1, 0, 2, "16x16x32 f16, 2 chains");
  T(1, 0, 1, "16x16x32 f16, 1 chain (dependent)");
  T(2, 0, 4, "16x16x16 f16, 4 chains");
  T(2, 0, 2, "16x16x16 f16, 2 chains");
  T(2, 0, 1, "16x16x16 f16, 1 chain (dependent)");
  T(3, 0, 2, "32x32x8 f16, 2 chains");
  T(4, 0, 2, "32x32x16 f16, 2 chains");
  T(1, 2, 2, "16x16x32 f16 + 2 vector");
  T(1, 4, 2, "16x16x32 f16 + 4 vector       (~ fista_mfma_kernel: 4.4 vector per matrix instruction)");
  T(1, 4, 1, "16x16x32 f16 + 4 vector, ONE accumulator chain (each product waits for the one before)");
  T(1, 4, 4, "16x16x32 f16 + 4 vector, four chains");
  T(1, 5, 2, "16x16x32 f16 + 5 vector");
  T(1, 5, 1, "16x16x32 f16 + 5 vector, one chain");
  T(1, 6, 2, "16x16x32 f16 + 6 vector");
  T(0, 4, 2, "4 vector alone");
  T(0, 6, 2, "6 vector alone");
  T(4, 8, 2, "32x32x16 f16 + 8 vector       (the same work as 2 x (16x16x32 + 4 vector))");
  T(4, 10, 2, "32x32x16 f16 + 10 vector");
  T(2, 2, 2, "16x16x16 f16 + 2 vector");
  T(2, 3, 2, "16x16x16 f16 + 3 vector       (6 narrow for 4 wide: 2.9 vector per matrix instruction)");
  T(2, 4, 2, "16x16x16 f16 + 4 vector");
  T(2, 6, 2, "16x16x16 f16 + 6 vector");
  }
  return 0;
}
