// torch.ops.pybold_hip: a thin TORCH_LIBRARY shim over the C ABI of libpybold_hip.so
// (include/pybold_hip.h stays the authoritative boundary).  Tensors in, launches on
// PyTorch's current HIP stream of the tensors' device, TORCH_CHECK on the return code -- the
// dispatcher-level surface SURVEY.md 8(b) sketches for the reference's solver entry points:
//   deconv / _loops_deconv loops   pybold/bold_signal.py:62-72, :259-276   -> fista_solve
//   outputs z, x                   pybold/bold_signal.py:74-75, :97        -> fista_outputs
//   H.op / H.adj                   pybold/linear.py:73-113                 -> op_forward / op_adjoint
//   hrf_fit_err as normal equations + its 1-D fit   pybold/bold_signal.py:217-222, :329-333
//                                                                          -> hrf_normal_eq, theta_fit
// Host-only translation unit (no device code): built by `make torch_ops` with the C++ compiler.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <tuple>

#include "../../include/pybold_hip.h"

namespace {

using at::Tensor;

void check(int rc, const char* what) { TORCH_CHECK(rc == PB_OK, what, ": ", pb_last_error()); }

void* stream_of(const Tensor& t) {
  return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}

const Tensor& rows(const Tensor& t, at::ScalarType dt, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.dim() == 2 && t.scalar_type() == dt && t.stride(1) == 1, name,
              " must be a 2-D CUDA tensor of the right dtype with unit stride along time");
  return t;
}
// a 1-D float64 vector living on `like`'s device (HRF taps, sample times): a CPU tensor or another dtype here would
// make the kernel read a foreign pointer
const Tensor& dev_vec(const Tensor& t, const Tensor& like, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kDouble && t.is_contiguous() && t.numel() >= 1, name,
              " must be a contiguous float64 CUDA tensor");
  TORCH_CHECK(t.device() == like.device(), name, " must live on the same device as the data (", like.device(), ")");
  return t;
}
void same_device(const Tensor& t, const Tensor& like, const char* name) {
  TORCH_CHECK(t.device() == like.device(), name, " must live on the same device as Y (", like.device(), ")");
}
int64_t ld(const Tensor& t) { return t.size(0) > 1 ? t.stride(0) : std::max<int64_t>(t.stride(0), t.size(1)); }

// W (float64, in: warm start, out: iterate), J (float32 or undefined), n_done (int32) are written in place.
void fista_solve(const Tensor& Y, Tensor W, const Tensor& taps_host, const c10::optional<Tensor>& taps_dev,
                 double step, double lbda, const c10::optional<Tensor>& lbda_vec, const Tensor& betas,
                 int64_t n_iter, c10::optional<Tensor> J, int64_t stop_mode, double tol, int64_t wind,
                 Tensor n_done, int64_t y_rep, int64_t flags) {
  rows(Y, at::kFloat, "Y");
  rows(W, at::kDouble, "W");
  TORCH_CHECK(!taps_host.is_cuda() && taps_host.scalar_type() == at::kDouble && taps_host.is_contiguous(),
              "taps_host must be a contiguous float64 CPU tensor");
  TORCH_CHECK(betas.is_cuda() && betas.scalar_type() == at::kDouble && betas.numel() >= n_iter, "betas: float64 CUDA, n_iter entries");
  TORCH_CHECK(n_done.is_cuda() && n_done.scalar_type() == at::kInt && n_done.numel() == W.size(0), "n_done: int32 CUDA (P,)");
  TORCH_CHECK(W.size(0) == Y.size(0) * y_rep && W.size(1) == Y.size(1), "W must be (V * y_rep, N)");
  if (lbda_vec) TORCH_CHECK(lbda_vec->is_cuda() && lbda_vec->scalar_type() == at::kDouble && lbda_vec->numel() == W.size(0), "lbda_vec: float64 CUDA (P,)");
  if (J) {
    rows(*J, at::kFloat, "J");
    TORCH_CHECK(J->size(0) == W.size(0) && J->size(1) >= n_iter, "J must be (P, >= n_iter)");
    same_device(*J, Y, "J");
  }
  if (taps_dev) dev_vec(*taps_dev, Y, "taps_dev");
  TORCH_CHECK(!taps_dev || taps_dev->numel() == taps_host.numel(), "taps_dev and taps_host must hold the same taps");
  TORCH_CHECK(n_iter >= 0 && y_rep >= 1, "n_iter >= 0 and y_rep >= 1");
  same_device(W, Y, "W");
  same_device(n_done, Y, "n_done");
  same_device(betas, Y, "betas");
  if (lbda_vec) same_device(*lbda_vec, Y, "lbda_vec");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(Y.device());
  check(pb_fista_solve(Y.data_ptr<float>(), ld(Y), (int)y_rep, W.data_ptr<double>(), ld(W), (int)W.size(0), (int)Y.size(1),
                       taps_host.data_ptr<double>(), taps_dev ? taps_dev->data_ptr<double>() : nullptr,
                       (int)taps_host.numel(), step, lbda, lbda_vec ? lbda_vec->data_ptr<double>() : nullptr,
                       betas.data_ptr<double>(), (int)n_iter, J ? J->data_ptr<float>() : nullptr, J ? ld(*J) : 0,
                       (int)stop_mode, tol, (int)wind, n_done.data_ptr<int32_t>(), (unsigned)flags, stream_of(Y)),
        "pb_fista_solve");
}

std::tuple<Tensor, Tensor> fista_outputs(const Tensor& W, const Tensor& taps_dev) {
  rows(W, at::kDouble, "W");
  dev_vec(taps_dev, W, "taps_dev");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(W.device());
  Tensor Z = at::empty_like(W), X = at::empty_like(W);
  check(pb_fista_outputs(W.data_ptr<double>(), ld(W), (int)W.size(0), (int)W.size(1), taps_dev.data_ptr<double>(),
                         (int)taps_dev.numel(), Z.data_ptr<double>(), ld(Z), X.data_ptr<double>(), ld(X), stream_of(W)),
        "pb_fista_outputs");
  return {X, Z};
}

Tensor op_forward(const Tensor& X, const Tensor& taps_dev, int64_t dim_out) {
  rows(X, at::kDouble, "x");
  dev_vec(taps_dev, X, "taps_dev");
  TORCH_CHECK(dim_out >= 1, "dim_out >= 1");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(X.device());
  Tensor out = at::empty({X.size(0), dim_out}, X.options());
  check(pb_op_forward(X.data_ptr<double>(), ld(X), out.data_ptr<double>(), ld(out), (int)X.size(0), (int)X.size(1), (int)dim_out,
                      taps_dev.data_ptr<double>(), (int)taps_dev.numel(), stream_of(X)), "pb_op_forward");
  return out;
}

Tensor op_adjoint(const Tensor& R, const Tensor& taps_dev, int64_t dim_in) {
  rows(R, at::kDouble, "r");
  dev_vec(taps_dev, R, "taps_dev");
  TORCH_CHECK(dim_in >= 1, "dim_in >= 1");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(R.device());
  Tensor out = at::empty({R.size(0), dim_in}, R.options());
  check(pb_op_adjoint(R.data_ptr<double>(), ld(R), out.data_ptr<double>(), ld(out), (int)R.size(0), (int)dim_in, (int)R.size(1),
                      taps_dev.data_ptr<double>(), (int)taps_dev.numel(), stream_of(R)), "pb_op_adjoint");
  return out;
}

// the normal equations of hrf_fit_err summed over the voxels: (K*K + K + 1,) float64
Tensor hrf_normal_eq(const Tensor& Z, const Tensor& Y, int64_t K) {
  rows(Z, at::kDouble, "Z");
  rows(Y, at::kFloat, "Y");
  same_device(Z, Y, "Z");
  TORCH_CHECK(Z.size(0) == Y.size(0) && Z.size(1) == Y.size(1), "Z and Y must have the same shape");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(Z.device());
  const int64_t ne = pb_hrf_normal_eq_len((int)K);
  Tensor out = at::empty({ne}, Z.options());
  Tensor work = at::empty({2048 * ne}, Z.options());
  check(pb_hrf_normal_eq(Z.data_ptr<double>(), ld(Z), Y.data_ptr<float>(), ld(Y), (int)Z.size(0), (int)Z.size(1), (int)K, 0,
                         work.data_ptr<double>(), work.numel(), out.data_ptr<double>(), stream_of(Z)), "pb_hrf_normal_eq");
  return out;
}

// argmin over theta of the quadratic form: (theta (M,), cost (M,), taps (M, K))
std::tuple<Tensor, Tensor, Tensor> theta_fit(const Tensor& ne, const Tensor& t, double a_peak, double loc_peak,
                                             double a_under, double loc_under, double ratio, double lo, double hi,
                                             int64_t n_refine) {
  TORCH_CHECK(ne.is_cuda() && ne.scalar_type() == at::kDouble && ne.dim() == 2 && ne.stride(1) == 1, "ne: float64 CUDA (M, len)");
  dev_vec(t, ne, "t (sample times)");
  c10::hip::HIPGuardMasqueradingAsCUDA guard(ne.device());
  const int64_t M = ne.size(0), K = t.numel();
  TORCH_CHECK(ne.size(1) == pb_hrf_normal_eq_len((int)K), "ne must hold K*K + K + 1 values per set for K = t.numel() taps");
  Tensor theta = at::empty({M}, ne.options()), cost = at::empty({M}, ne.options()), taps = at::empty({M, K}, ne.options());
  check(pb_theta_fit(ne.data_ptr<double>(), ld(ne), (int)M, (int)K, t.data_ptr<double>(), a_peak, loc_peak, a_under, loc_under,
                     ratio, lo, hi, (int)n_refine, theta.data_ptr<double>(), cost.data_ptr<double>(), taps.data_ptr<double>(),
                     K, stream_of(ne)), "pb_theta_fit");
  return {theta, cost, taps};
}

}  // namespace

TORCH_LIBRARY(pybold_hip, m) {
  m.def("fista_solve(Tensor Y, Tensor(a!) W, Tensor taps_host, Tensor? taps_dev, float step, float lbda, Tensor? lbda_vec, "
        "Tensor betas, int n_iter, Tensor(b!)? J, int stop_mode, float tol, int wind, Tensor(c!) n_done, int y_rep, int flags) -> ()");
  m.def("fista_outputs(Tensor W, Tensor taps_dev) -> (Tensor, Tensor)");
  m.def("op_forward(Tensor X, Tensor taps_dev, int dim_out) -> Tensor");
  m.def("op_adjoint(Tensor R, Tensor taps_dev, int dim_in) -> Tensor");
  m.def("hrf_normal_eq(Tensor Z, Tensor Y, int K) -> Tensor");
  m.def("theta_fit(Tensor ne, Tensor t, float a_peak, float loc_peak, float a_under, float loc_under, float ratio, "
        "float lo, float hi, int n_refine) -> (Tensor, Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(pybold_hip, CUDA, m) {
  m.impl("fista_solve", &fista_solve);
  m.impl("fista_outputs", &fista_outputs);
  m.impl("op_forward", &op_forward);
  m.impl("op_adjoint", &op_adjoint);
  m.impl("hrf_normal_eq", &hrf_normal_eq);
  m.impl("theta_fit", &theta_fit);
}
