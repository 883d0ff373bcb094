// Do MFMA instructions overlap with VALU work on gfx950?  Times a loop of 64 independent
// v_pk_fma_f32 per trip alone, with NM interleaved v_mfma_f32_32x32x2_f32 (64 cycles each
// in the matrix pipe), and the MFMAs alone.  If the matrix pipe runs beside the VALU, "both"
// costs max(valu, mfma), not the sum.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_coissue.hip -o mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int NV, int NM, int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, float b) {
  float2v acc[8];
  float2v av = {a, a}, bv = {b, b};
  for (int i = 0; i < 8; ++i) acc[i] = float2v{(float)threadIdx.x + i, 1.f};
  float16v c0 = {0}, c1 = {0};
  float4v d0 = {0}, d1 = {0};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if constexpr (NM > 0) {
        if (g < NM || NM >= 4) {
          if constexpr (KIND == 0) {
            if (g % 2 == 0) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            else c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
          } else {
            if (g % 2 == 0) d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
            else d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d1, 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NV / 32; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  for (int i = 0; i < 4; ++i) s += d0[i] + d1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float run(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  CHECK(hipMalloc(&out, 256 * 4 * 2 * 256 * sizeof(float) * 4));
  for (int wps : {1, 2}) {
    const int blocks = 256 * wps;     // 4 waves per block: wps waves per SIMD
    printf("waves per SIMD = %d  (per trip: 64 v_pk_fma_f32 = 256 issue cycles per wave)\n", wps);
#define T(NV, NM, KIND, label) printf("  %-46s %8.3f ms\n", label, run([&] { hipLaunchKernelGGL((k<NV, NM, KIND>), dim3(blocks), dim3(256), 0, 0, out, 1.0f, 0.5f); }));
    T(64, 0, 0, "VALU only (64 pk_fma)");
    T(0, 4, 0, "4 x mfma 32x32x2 f32 only (4 x 64 cycles)");
    T(64, 1, 0, "64 pk_fma + 1 mfma 32x32x2");
    T(64, 2, 0, "64 pk_fma + 2 mfma 32x32x2");
    T(64, 4, 0, "64 pk_fma + 4 mfma 32x32x2");
    T(0, 4, 1, "4 x mfma 16x16x4 f32 only (4 x 32 cycles)");
    T(64, 4, 1, "64 pk_fma + 4 mfma 16x16x4");
  }
  return 0;
}
