// Shared device/host helpers for the pyBOLD gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace pb {

typedef float f2 __attribute__((ext_vector_type(2)));

// Compile-time loop: f(std::integral_constant<int, I>) for I in [B, E).
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// DPP controls (wave64, rows of 16 lanes).  Out-of-row source lanes read 0
// (bound_ctrl), which is exactly the causal zero padding of the Toeplitz
// operator and the identity of the scans.
constexpr int DPP_ROW_SHL = 0x100;  // lane i reads lane i+n of its row
constexpr int DPP_ROW_SHR = 0x110;  // lane i reads lane i-n of its row
constexpr int DPP_ROW_ROR = 0x120;  // rotate right within the row

template <int CTRL>
__device__ __forceinline__ float dpp_zero(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// value of lane (i - n) of the same 16-lane row, 0 if that lane is outside it
template <int n>
__device__ __forceinline__ float row_from_below(float v) {
  static_assert(n >= 1 && n <= 15, "row shift out of range");
  return dpp_zero<DPP_ROW_SHR + n>(v);
}
// value of lane (i + n) of the same row, 0 outside
template <int n>
__device__ __forceinline__ float row_from_above(float v) {
  static_assert(n >= 1 && n <= 15, "row shift out of range");
  return dpp_zero<DPP_ROW_SHL + n>(v);
}

// sum over the 16 lanes of a row, result in every lane
__device__ __forceinline__ float row_allsum(float v) {
  v += dpp_zero<DPP_ROW_ROR + 8>(v);
  v += dpp_zero<DPP_ROW_ROR + 4>(v);
  v += dpp_zero<DPP_ROW_ROR + 2>(v);
  v += dpp_zero<DPP_ROW_ROR + 1>(v);
  return v;
}

// inclusive prefix sum over the lanes of a row (Hillis-Steele, zero fill)
__device__ __forceinline__ float row_prefix_incl(float t) {
  t += row_from_below<1>(t);
  t += row_from_below<2>(t);
  t += row_from_below<4>(t);
  t += row_from_below<8>(t);
  return t;
}
// inclusive suffix sum over the lanes of a row
__device__ __forceinline__ float row_suffix_incl(float t) {
  t += row_from_above<1>(t);
  t += row_from_above<2>(t);
  t += row_from_above<4>(t);
  t += row_from_above<8>(t);
  return t;
}

}  // namespace pb
