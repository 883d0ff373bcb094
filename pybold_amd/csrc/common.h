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

// d = u - prox(u), the prox written as pybold/bold_signal.py:66 writes it: sign(u) * max(|u| - th, 0).
// th >= 0: the clamp every kernel uses.  th < 0: the noise-driven lambda search (:141-145) lets alpha -- hence
// lambda = 1 / (2 alpha) -- go NEGATIVE (it does in the reference's default call on the golden series), and the
// reference's expression then GROWS every entry by |th| (sign(0) = 0 keeps zeros).  Only the float64 kernels
// (that search runs on them) reproduce this; th is uniform over the problem's lanes: a scalar branch.
__device__ __forceinline__ double prox_excess_ref(double u, double th) {
  if (th >= 0.0) return fmin(fmax(u, -th), th);
  return u > 0.0 ? th : (u < 0.0 ? -th : 0.0);
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

// the same move, but lanes whose source lies outside the row keep `old` (bound_ctrl off)
template <int CTRL>
__device__ __forceinline__ float dpp_keep(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                               __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
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

// ---- segments of LPV lanes (16 = one DPP row, 64 = the whole wave) ---------------
constexpr int DPP_WAVE_SHL1 = 0x130;   // lane i reads lane i+1 of the wave (0 past the end)
constexpr int DPP_WAVE_SHR1 = 0x138;   // lane i reads lane i-1 of the wave (0 before lane 0)
constexpr int DPP_ROW_BCAST15 = 0x142; // lane 15 of each row -> every lane of the next row
constexpr int DPP_ROW_BCAST31 = 0x143; // lane 31 -> every lane of rows 2 and 3

// value of lane (i - d) of the same segment, 0 outside it
template <int LPV, int d>
__device__ __forceinline__ float seg_from_below(float v) {
  if constexpr (LPV == 16) {
    return row_from_below<d>(v);
  } else {
    static_assert(LPV == 64, "segments are one DPP row or one wave");
#pragma unroll
    for (int k = 0; k < d; ++k) v = dpp_zero<DPP_WAVE_SHR1>(v);
    return v;
  }
}
template <int LPV, int d>
__device__ __forceinline__ float seg_from_above(float v) {
  if constexpr (LPV == 16) {
    return row_from_above<d>(v);
  } else {
#pragma unroll
    for (int k = 0; k < d; ++k) v = dpp_zero<DPP_WAVE_SHL1>(v);
    return v;
  }
}

// inclusive prefix sum over the lanes of a segment
template <int LPV>
__device__ __forceinline__ float seg_prefix_incl(float t) {
  t = row_prefix_incl(t);
  if constexpr (LPV == 64) {
    // rows 1 and 3 add the total of the row before them, then rows 2 and 3 the total of
    // the first half (disabled rows receive `old` = 0)
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
             0, __builtin_bit_cast(int, t), DPP_ROW_BCAST15, 0xa, 0xf, false));
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
             0, __builtin_bit_cast(int, t), DPP_ROW_BCAST31, 0xc, 0xf, false));
  }
  return t;
}
// sum of the lanes strictly below / strictly above this one in its segment
template <int LPV>
__device__ __forceinline__ float seg_sum_below(float tot) {
  return seg_from_below<LPV, 1>(seg_prefix_incl<LPV>(tot));
}
template <int LPV>
__device__ __forceinline__ float seg_sum_above(float tot) {
  if constexpr (LPV == 16) {
    return row_from_above<1>(row_suffix_incl(tot));
  } else {
    const float incl = seg_prefix_incl<64>(tot);
    const float total = __builtin_bit_cast(
        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, incl), 63));
    return total - incl;
  }
}
// sum over the segment, result in every lane
template <int LPV>
__device__ __forceinline__ float seg_allsum(float v) {
  v = row_allsum(v);
  if constexpr (LPV == 64) {
    const int b = __builtin_bit_cast(int, v);
    v = (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
        (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
  }
  return v;
}

// float64 sum over the segment, result in every lane (two 32-bit DPP rotations per step)
template <int LPV>
__device__ __forceinline__ double seg_allsum_f64(double v) {
  static_for<0, 4>([&](auto sc) {
    constexpr int sh = 8 >> decltype(sc)::value;
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, DPP_ROW_ROR + sh, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), DPP_ROW_ROR + sh, 0xf, 0xf, true);
    v += __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
  });
  if constexpr (LPV == 64) {
    const long long b = __builtin_bit_cast(long long, v);
    auto lane = [&](int l) {
      const int lo = __builtin_amdgcn_readlane((int)b, l);
      const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
      return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
    };
    v = (lane(0) + lane(16)) + (lane(32) + lane(48));
  }
  return v;
}

}  // namespace pb
