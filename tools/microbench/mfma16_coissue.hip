// Does a 16-bit MFMA stream run BESIDE vector work on gfx950?  (round 2 measured only the f32
// MFMA, which shares the fp32 datapath: times add.)  Per loop trip: G groups of
// { NM x v_mfma (f16 or bf16), NV x vector instruction }, all independent chains.
//   vector kinds: v_pk_fma_f32, v_fma_f64, and the mix a split-precision FIR would need
//   (cvt f64->f32, cvt_pkrtz f32->f16, fma_mix, f64 fma/min/max).
// If the matrix pipe runs beside the VALU, "both" ~ max(vector, mfma) + the MFMA's issue
// slots; if not, the sum.  Build: hipcc -O3 --offload-arch=gfx950 mfma16_coissue.hip -o mfma16_coissue
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 1024;
constexpr int G = 8;                       // groups per trip
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __bf16 b8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));

// MK: 0 none, 1 f16 16x16x32, 2 f16 32x32x16, 3 bf16 16x16x32
// VK: 0 v_pk_fma_f32, 1 v_fma_f64, 2 mix
template <int NV, int NM, int MK, int VK>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  f2v acc[8];
  double dac[8];
  f2v av = {a, a}, bv = {b, b};
  double da = a, db = b;
  for (int i = 0; i < 8; ++i) { acc[i] = f2v{(float)threadIdx.x + i, 1.f}; dac[i] = threadIdx.x + i; }
  h8v ha, hb;
  b8v ba, bb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b * i); ba[i] = (__bf16)(a + i); bb[i] = (__bf16)(b * i); }
  f4v c4[4] = {};
  f16v c16[2] = {};
  float cf[4] = {a, b, a + b, a - b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if constexpr (MK == 1) c4[(g * NM + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c4[(g * NM + m) & 3], 0, 0, 0);
        if constexpr (MK == 2) c16[(g * NM + m) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c16[(g * NM + m) & 1], 0, 0, 0);
        if constexpr (MK == 3) c4[(g * NM + m) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, c4[(g * NM + m) & 3], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int i = (g * NV + u) & 7;
        if constexpr (VK == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
        if constexpr (VK == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dac[i]) : "v"(da), "v"(db));
        if constexpr (VK == 2) {
          // one of: f64 fma, f64 max, cvt f64->f32, cvt_pkrtz, fma_mix-like, f32 add
          const int kind = u % 6;
          if (kind == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dac[i]) : "v"(da), "v"(db));
          if (kind == 1) asm volatile("v_max_f64 %0, %1, %0" : "+v"(dac[i]) : "v"(da));
          if (kind == 2) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(cf[i & 3]) : "v"(dac[i]));
          if (kind == 3) { int pk; asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(cf[i & 3]), "v"(cf[(i + 1) & 3])); asm volatile("" :: "v"(pk)); }
          if (kind == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(cf[i & 3]) : "v"(a), "v"(b));
          if (kind == 5) asm volatile("v_add_f32 %0, %1, %0" : "+v"(cf[i & 3]) : "v"(a));
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + (float)dac[i];
  for (int i = 0; i < 4; ++i) s += c4[i][0] + c4[i][1] + c4[i][2] + c4[i][3] + cf[i];
  for (int i = 0; i < 16; ++i) s += c16[0][i] + c16[1][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float run(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
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
  CHECK(hipMalloc(&out, 256 * 4 * 2 * 256 * sizeof(float) * 4));
  for (int wps : {1, 2}) {
    const int blocks = 256 * wps;     // 4 waves per block: wps waves per SIMD
    printf("waves per SIMD = %d   (%d trips x %d groups; ms per launch, best of 3)\n", wps, ITERS, G);
#define T(NV, NM, MK, VK, label) printf("  %-64s %8.3f ms\n", label, run([&] { hipLaunchKernelGGL((k<NV, NM, MK, VK>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.5f, ITERS); }));
    T(8, 0, 0, 0, "8 pk_fma_f32 / group                      (32 issue cycles)");
    T(0, 1, 1, 0, "1 mfma f16 16x16x32 / group               (16 pipe cycles)");
    T(0, 2, 1, 0, "2 mfma f16 16x16x32 / group               (32 pipe cycles)");
    T(0, 1, 2, 0, "1 mfma f16 32x32x16 / group               (32 pipe cycles)");
    T(0, 2, 3, 0, "2 mfma bf16 16x16x32 / group              (32 pipe cycles)");
    T(8, 1, 1, 0, "8 pk_fma_f32 + 1 mfma f16 16x16x32");
    T(8, 2, 1, 0, "8 pk_fma_f32 + 2 mfma f16 16x16x32");
    T(8, 1, 2, 0, "8 pk_fma_f32 + 1 mfma f16 32x32x16");
    T(8, 2, 3, 0, "8 pk_fma_f32 + 2 mfma bf16 16x16x32");
    T(4, 2, 1, 0, "4 pk_fma_f32 + 2 mfma f16 16x16x32");
    T(4, 1, 2, 0, "4 pk_fma_f32 + 1 mfma f16 32x32x16");
    T(16, 2, 1, 0, "16 pk_fma_f32 + 2 mfma f16 16x16x32");
    T(16, 1, 2, 0, "16 pk_fma_f32 + 1 mfma f16 32x32x16");
    T(8, 0, 0, 1, "8 fma_f64 / group");
    T(8, 2, 1, 1, "8 fma_f64 + 2 mfma f16 16x16x32");
    T(8, 1, 2, 1, "8 fma_f64 + 1 mfma f16 32x32x16");
    T(12, 0, 0, 2, "12 mixed (f64 fma/max, cvt, pkrtz, f32 fma/add) / group");
    T(12, 2, 1, 2, "12 mixed + 2 mfma f16 16x16x32");
    T(12, 1, 2, 2, "12 mixed + 1 mfma f16 32x32x16");
    T(6, 0, 0, 2, "6 mixed / group");
    T(6, 1, 2, 2, "6 mixed + 1 mfma f16 32x32x16      (the ratio of a 32x32x16 form of fista_mfma_kernel)");
    T(3, 0, 0, 2, "3 mixed / group");
    T(3, 1, 1, 2, "3 mixed + 1 mfma f16 16x16x32      (~ the ratio of fista_mfma_kernel: 3.6 vector per matrix instruction)");
    T(4, 1, 1, 2, "4 mixed + 1 mfma f16 16x16x32");
    T(18, 2, 1, 2, "18 mixed + 2 mfma f16 16x16x32");
    T(18, 1, 2, 2, "18 mixed + 1 mfma f16 32x32x16");
  }
  return 0;
}
