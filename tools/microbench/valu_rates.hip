// Instruction-rate microbenchmark for gfx950: decides the arithmetic layout of
// the fused FISTA kernel (plain vs packed fp32 FMA, fp64 FMA, DPP moves, cvt).
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;   // loop trips
constexpr int UNROLL = 32;    // instructions per trip

typedef float float2v __attribute__((ext_vector_type(2)));

__global__ void k_fma32(float* out, float a, float b) {
  float acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(b));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_pkfma32(float* out, float a, float b) {
  float2v acc[8];
  float2v av = {a, a}, bv = {b, b};
  for (int i = 0; i < 8; ++i) acc[i] = float2v{(float)threadIdx.x + i, 1.f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av), "v"(bv));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// packed FMA with a scalar-pair tap operand and op_sel broadcast of its low half
__global__ void k_pkfma32_sgpr(float* out, float a, float b) {
  float2v acc[8];
  float2v av = {a, a};
  float2v bv = {b, b + 1.f};
  for (int i = 0; i < 8; ++i) acc[i] = float2v{(float)threadIdx.x + i, 1.f};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(av), "s"(bv));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fma64(float* out, double a, double b) {
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

__global__ void k_add64(float* out, double a) {
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_add_f64 %0, %1, %0" : "+v"(acc[i]) : "v"(a));
    }
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

__global__ void k_dppmov(float* out, float a) {
  float acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x + i + a;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(acc[i]));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// DPP mov from 8 distinct sources into 8 distinct destinations (no RAW on itself)
__global__ void k_dppmov_indep(float* out, float a) {
  float src[8], dst[8];
  for (int i = 0; i < 8; ++i) { src[i] = threadIdx.x + i + a; dst[i] = 0; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(dst[i]) : "v"(src[i]));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += dst[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// fmac with DPP source: acc += dpp(src) * tap(vgpr)
__global__ void k_fmac_dpp(float* out, float a, float b) {
  float acc[8], src[8];
  for (int i = 0; i < 8; ++i) { acc[i] = threadIdx.x + i; src[i] = a + i; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(acc[i]) : "v"(src[i]), "v"(b));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_cvt_f32_f64(float* out, double a) {
  float acc[8]; double src[8];
  for (int i = 0; i < 8; ++i) { src[i] = a + threadIdx.x + i; acc[i] = 0; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(acc[i]) : "v"(src[i]));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_cvt_f64_f32(float* out, float a) {
  double acc[8]; float src[8];
  for (int i = 0; i < 8; ++i) { src[i] = a + threadIdx.x + i; acc[i] = 0; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL / 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(acc[i]) : "v"(src[i]));
    }
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

template <typename F>
int run(const char* name, F launch, float* dout, int waves_per_simd, double flop_per_instr_lane) {
  // one 256-thread block = 4 waves = 1 wave per SIMD of a CU; blocks = 256 CUs * waves_per_simd
  int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(blocks); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0)); launch(blocks); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  double instr_per_wave = (double)ITERS * UNROLL;
  double wave_instr = instr_per_wave * blocks * 4;           // total wave-instructions
  double per_simd_per_s = wave_instr / 1024.0 / (best * 1e-3);
  double cyc_per_instr = 2.4e9 / per_simd_per_s;            // at nominal 2.4 GHz
  double tflops = wave_instr * 64 * flop_per_instr_lane / (best * 1e-3) / 1e12;
  printf("%-18s waves/SIMD=%d  %.3f ms  %.2f nominal-cycles/wave-instr/SIMD  %.1f TFLOP/s\n",
         name, waves_per_simd, best, cyc_per_instr, tflops);
  return 0;
}

int main() {
  float* dout; CHECK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float)));
  for (int w : {1, 2, 4, 8}) {
    run("v_fma_f32", [&](int b) { hipLaunchKernelGGL(k_fma32, dim3(b), dim3(256), 0, 0, dout, 1.0001f, 0.5f); }, dout, w, 2);
    run("v_pk_fma_f32", [&](int b) { hipLaunchKernelGGL(k_pkfma32, dim3(b), dim3(256), 0, 0, dout, 1.0001f, 0.5f); }, dout, w, 4);
    run("v_pk_fma_f32 sgpr", [&](int b) { hipLaunchKernelGGL(k_pkfma32_sgpr, dim3(b), dim3(256), 0, 0, dout, 1.0001f, 0.5f); }, dout, w, 4);
    run("v_fma_f64", [&](int b) { hipLaunchKernelGGL(k_fma64, dim3(b), dim3(256), 0, 0, dout, 1.0001, 0.5); }, dout, w, 2);
    run("v_add_f64", [&](int b) { hipLaunchKernelGGL(k_add64, dim3(b), dim3(256), 0, 0, dout, 1.0001); }, dout, w, 1);
    run("v_mov_dpp(+nop)", [&](int b) { hipLaunchKernelGGL(k_dppmov, dim3(b), dim3(256), 0, 0, dout, 1.0001f); }, dout, w, 0);
    run("v_mov_dpp indep", [&](int b) { hipLaunchKernelGGL(k_dppmov_indep, dim3(b), dim3(256), 0, 0, dout, 1.0001f); }, dout, w, 0);
    run("v_fmac_f32_dpp", [&](int b) { hipLaunchKernelGGL(k_fmac_dpp, dim3(b), dim3(256), 0, 0, dout, 1.0001f, 0.5f); }, dout, w, 2);
    run("v_cvt_f32_f64", [&](int b) { hipLaunchKernelGGL(k_cvt_f32_f64, dim3(b), dim3(256), 0, 0, dout, 1.0001); }, dout, w, 0);
    run("v_cvt_f64_f32", [&](int b) { hipLaunchKernelGGL(k_cvt_f64_f32, dim3(b), dim3(256), 0, 0, dout, 1.0001f); }, dout, w, 0);
  }
  hipFree(dout);
  return 0;
}
