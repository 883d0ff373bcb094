// Host-side launcher of one (S, KT) specialisation of fista_fast_kernel.  Each
// specialisation is instantiated in its own translation unit (fast_inst.hip with
// -DPB_S/-DPB_KT) so the table builds in parallel; capi.hip only sees the
// extern template declarations.
#pragma once
#include "../../include/pybold_hip.h"
#include "fista_fast.h"

namespace pb {

template <int S, int KT>
int launch_fast(const FistaArgs& a, const double* taps, int K, bool with_j, int stop,
                hipStream_t st) {
  const auto tp = make_tap_pairs<KT>(taps, K);
  const dim3 grid((unsigned)((launch_count(a) * 16 + 255) / 256)), block(256);
  if (stop == PB_STOP_NONE) {
    if (with_j) hipLaunchKernelGGL((fista_fast_kernel<S, KT, true, 0>), grid, block, 0, st, a, tp);
    else hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 0>), grid, block, 0, st, a, tp);
  } else if (stop == PB_STOP_LOOPS) {
    if (with_j) hipLaunchKernelGGL((fista_fast_kernel<S, KT, true, 1>), grid, block, 0, st, a, tp);
    else hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 1>), grid, block, 0, st, a, tp);
  } else {   // PB_STOP_WINDOW with wind in {4, 6, 8} (checked by the caller): LDS ring of wind - 2 increments
    if constexpr (S <= 20) {
      const size_t lds = (size_t)16 * (a.wind - 2) * S * 16 * sizeof(float);
      if (a.wind == 6) {
        if (with_j) hipLaunchKernelGGL((fista_fast_kernel<S, KT, true, 2>), grid, block, lds, st, a, tp);
        else hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 2>), grid, block, lds, st, a, tp);
      } else if (a.wind == 4) {
        if (with_j) hipLaunchKernelGGL((fista_fast_kernel<S, KT, true, 2, false, 16, 4>), grid, block, lds, st, a, tp);
        else hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 2, false, 16, 4>), grid, block, lds, st, a, tp);
      } else if (a.wind == 8) {
        if (with_j) hipLaunchKernelGGL((fista_fast_kernel<S, KT, true, 2, false, 16, 8>), grid, block, lds, st, a, tp);
        else hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 2, false, 16, 8>), grid, block, lds, st, a, tp);
      } else {
        return 1;
      }
    } else {
      return 1;   // no increment ring for S > 20: the caller must not come here
    }
  }
  return 0;
}

// one problem per wave (long series)
template <int S, int KT>
int launch_wide(const FistaArgs& a, const double* taps, int K, bool with_j, int stop, hipStream_t st) {
  const auto tp = make_tap_pairs<KT>(taps, K);
  const dim3 grid((unsigned)((launch_count(a) * 64 + 255) / 256)), block(256);
  if (stop == PB_STOP_NONE) {
    if (with_j) hipLaunchKernelGGL((fista_fast_kernel<S, KT, true, 0, false, 64>), grid, block, 0, st, a, tp);
    else hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 0, false, 64>), grid, block, 0, st, a, tp);
  } else if (stop == PB_STOP_LOOPS) {
    if (with_j) hipLaunchKernelGGL((fista_fast_kernel<S, KT, true, 1, false, 64>), grid, block, 0, st, a, tp);
    else hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 1, false, 64>), grid, block, 0, st, a, tp);
  } else {   // PB_STOP_WINDOW, wind in {4, 6, 8} and S <= 20 (checked by the caller)
    if constexpr (S <= 20) {
      const size_t lds = (size_t)4 * (a.wind - 2) * S * 64 * sizeof(float);
      if (a.wind == 6) {
        if (with_j) hipLaunchKernelGGL((fista_fast_kernel<S, KT, true, 2, false, 64>), grid, block, lds, st, a, tp);
        else hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 2, false, 64>), grid, block, lds, st, a, tp);
      } else if (a.wind == 4) {
        if (with_j) hipLaunchKernelGGL((fista_fast_kernel<S, KT, true, 2, false, 64, 4>), grid, block, lds, st, a, tp);
        else hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 2, false, 64, 4>), grid, block, lds, st, a, tp);
      } else if (a.wind == 8) {
        if (with_j) hipLaunchKernelGGL((fista_fast_kernel<S, KT, true, 2, false, 64, 8>), grid, block, lds, st, a, tp);
        else hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 2, false, 64, 8>), grid, block, lds, st, a, tp);
      } else {
        return 1;
      }
    } else {
      return 1;   // no increment ring for S > 20: the caller must not come here
    }
  }
  return 0;
}

// per-problem taps/step variant (a.taps_pp, a.step_vec set); no cost trace
template <int S, int KT>
int launch_fast_pp(const FistaArgs& a, int stop, hipStream_t st) {
  const TapPairs<KT> tp{};
  const dim3 grid((unsigned)((launch_count(a) * 16 + 255) / 256)), block(256);
  if (stop == PB_STOP_NONE)
    hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 0, true>), grid, block, 0, st, a, tp);
  else
    hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 1, true>), grid, block, 0, st, a, tp);
  return 0;
}

// per-problem taps/step, one problem per wave
template <int S, int KT>
int launch_wide_pp(const FistaArgs& a, int stop, hipStream_t st) {
  const TapPairs<KT> tp{};
  const dim3 grid((unsigned)((launch_count(a) * 64 + 255) / 256)), block(256);
  if (stop == PB_STOP_NONE)
    hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 0, true, 64>), grid, block, 0, st, a, tp);
  else
    hipLaunchKernelGGL((fista_fast_kernel<S, KT, false, 1, true, 64>), grid, block, 0, st, a, tp);
  return 0;
}

}  // namespace pb
