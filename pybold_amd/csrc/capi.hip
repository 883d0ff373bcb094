// C ABI of libpybold_hip.so (see include/pybold_hip.h for the contract and the
// reference interfaces each entry point replaces).
#include "../../include/pybold_hip.h"

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

#include "fista_fast.h"
#include "generic.h"
#include "launch_fast.h"
#include "fista_pair.h"
#include "fista_pair_ffa.h"
#include "fista_exact.h"
#include "blind.h"
#include "fista_mfma.h"
#include "fista_mfma2.h"
#include "fista_mfma4.h"
#include "path.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(PB_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  g_err[0] = 0;
  return PB_OK;
}

constexpr int LDS_DOUBLES_MAX = 20000;  // 160 KB of LDS per workgroup

// ---- register-resident specialisations --------------------------------------
typedef int (*fast_launch_fn)(const pb::FistaArgs&, const double* taps, int K, bool with_j,
                              int stop, hipStream_t);
typedef int (*fast_launch_pp_fn)(const pb::FistaArgs&, int stop, hipStream_t);
typedef int (*pair_launch_fn)(const pb::FistaArgs&, const double* taps, int K, bool with_j, hipStream_t);
typedef int (*pair_cert_fn)(const pb::FistaArgs&, const double* taps, int K, hipStream_t);
typedef int (*pair_split_fn)(const pb::FistaArgs&, const double* taps, int K, bool with_j, bool cert, hipStream_t);

struct FastEntry {
  int S, KT;
  fast_launch_fn fn;
  fast_launch_pp_fn fn_pp;
  pair_launch_fn fn_pair;     // two-problems-per-row kernel (S <= 20, KT <= 32 only), else nullptr
  pair_launch_fn fn_pair_ffa; // the same with 2-parallel fast FIRs (fista_pair_ffa.h)
  int (*fn_pair_dev)(const pb::FistaArgs&, hipStream_t);   // ... reading ONE shared HRF from device memory
  pair_cert_fn fn_pair_cert;  // ... carrying the window rule (wind = 6) as a no-fire certificate
  pair_split_fn fn_pair_split; // ... ONE series of 16 S < N <= 32 S scans per row (its halves in the two slots)
};

}  // namespace

// instantiated in fast_inst.hip, one translation unit per table entry
namespace pb {
#define PB_FAST(S, KT)                                                                              \
  extern template int launch_fast<S, KT>(const FistaArgs&, const double*, int, bool, int, hipStream_t); \
  extern template int launch_fast_pp<S, KT>(const FistaArgs&, int, hipStream_t);            \
  extern template int launch_pair<S, KT>(const FistaArgs&, const double*, int, bool, hipStream_t); \
  extern template int launch_pair_ffa<S, KT>(const FistaArgs&, const double*, int, bool, hipStream_t); \
  extern template int launch_pair_ffa_dev<S, KT>(const FistaArgs&, hipStream_t);                  \
  extern template int launch_pair_ffa_cert<S, KT>(const FistaArgs&, const double*, int, hipStream_t); \
  extern template int launch_pair_ffa_split<S, KT>(const FistaArgs&, const double*, int, bool, bool, hipStream_t);
#include "fast_table.inc"
#undef PB_FAST
}  // namespace pb

namespace pb {
#define PB_MFMA(NB) extern template int launch_mfma<NB>(const FistaArgs&, const double*, int, bool, hipStream_t);
PB_MFMA(5) PB_MFMA(6) PB_MFMA(7) PB_MFMA(8) PB_MFMA(9) PB_MFMA(10)
#undef PB_MFMA
}
namespace pb {
#define PB_MFMA2(A, B) extern template int launch_mfma2<A, B>(const FistaArgs&, const double*, int, bool, hipStream_t);
PB_MFMA2(2, 3) PB_MFMA2(3, 3) PB_MFMA2(3, 4) PB_MFMA2(4, 4) PB_MFMA2(4, 5) PB_MFMA2(5, 5) PB_MFMA2(5, 6) PB_MFMA2(6, 6)
PB_MFMA2(6, 7) PB_MFMA2(7, 7) PB_MFMA2(7, 8) PB_MFMA2(8, 8) PB_MFMA2(8, 9) PB_MFMA2(9, 9) PB_MFMA2(9, 10) PB_MFMA2(10, 10)
#undef PB_MFMA2
#define PB_MFMA4(A) extern template int launch_mfma4<A>(const FistaArgs&, const double*, int, bool, hipStream_t);
PB_MFMA4(6) PB_MFMA4(7) PB_MFMA4(8) PB_MFMA4(9) PB_MFMA4(10)
#undef PB_MFMA4
}
namespace {
// the matrix-pipe form with one series split over the two waves of a workgroup (fista_mfma2.h): nb = ceil(N / 32)
// blocks of 32 samples (it keeps round 3's carry tile: its waves are bound by the vector work of the exchange, not by
// their matrix instructions -- the sum-slot form of fista_mfma.h measured 4 % slower there), 5 <= nb <= 20
// (129 .. 640 scans), floor(nb / 2) of them in the left wave; K <= 33; plain solves, the cost
// trace and the window rule (wind = 6) as a no-fire certificate; the shared-HRF z-step plain only
typedef int (*mfma2_launch_fn)(const pb::FistaArgs&, const double*, int, bool, hipStream_t);
// (34 <= K <= 65: three near tiles -- series of 225+ scans (four blocks per wave; the one-wave form carries shorter ones, and
// everything up to 310 scans but the certificate), plain solves, the cost
// trace, the certificate and the _loops_deconv rule: `extras` = taps from device memory wanted)
mfma2_launch_fn pick_mfma2(int N, int K, bool extras = true) {
  static const mfma2_launch_fn tab[] = {
      &pb::launch_mfma2<2, 3>, &pb::launch_mfma2<3, 3>, &pb::launch_mfma2<3, 4>, &pb::launch_mfma2<4, 4>,
      &pb::launch_mfma2<4, 5>, &pb::launch_mfma2<5, 5>, &pb::launch_mfma2<5, 6>, &pb::launch_mfma2<6, 6>,
      &pb::launch_mfma2<6, 7>, &pb::launch_mfma2<7, 7>, &pb::launch_mfma2<7, 8>, &pb::launch_mfma2<8, 8>,
      &pb::launch_mfma2<8, 9>, &pb::launch_mfma2<9, 9>, &pb::launch_mfma2<9, 10>, &pb::launch_mfma2<10, 10>};
  const int nb = (N + 31) / 32;
  if (K < 1 || K > 65 || nb < 5 || nb > 20) return nullptr;
  if (K > 33 && (extras || N <= 224)) return nullptr;         // (three near tiles: four blocks at least per wave)
  return tab[nb - 5];
}
// the same with one series split over the FOUR waves of a workgroup (fista_mfma4.h): 641 .. 1 280 scans, A = ceil(N / 128)
// blocks per wave (6 .. 10); K <= 33 with two near tiles: the call shapes of the two-wave form; 34 <= K <= 65 with three: plain
// solves, the cost trace, the certificate and the _loops_deconv rule (`extras` = taps from device memory: two tiles only)
mfma2_launch_fn pick_mfma4(int N, int K, bool extras = false) {
  static const mfma2_launch_fn tab[] = {&pb::launch_mfma4<6>, &pb::launch_mfma4<7>, &pb::launch_mfma4<8>, &pb::launch_mfma4<9>,
                                        &pb::launch_mfma4<10>};
  if (K < 1 || K > 65 || (K > 33 && extras) || N <= 640 || N > 1280) return nullptr;
  return tab[(N + 127) / 128 - 6];
}
typedef int (*mfma_launch_fn)(const pb::FistaArgs&, const double*, int, bool, hipStream_t);
// the matrix-pipe form (fista_mfma.h): NB = ceil(N / 31) blocks of 31 samples + one sum slot, 129 <= N <= 310; K <= 33
// with two near tiles (every variant), 34 <= K <= 64 with three (`extras` = window-rule certificate:
// not built for those)
constexpr int MFMA_K2 = 33, MFMA_K3 = 64;
constexpr int MFMA1_NMAX = 10 * pb::MFMA_SPAN;   // longer series (up to 640 scans) run on the split form (fista_mfma2.h)
mfma_launch_fn pick_mfma(int N, int K, bool extras = false) {
  static const mfma_launch_fn tab[] = {&pb::launch_mfma<5>, &pb::launch_mfma<6>, &pb::launch_mfma<7>,
                                       &pb::launch_mfma<8>, &pb::launch_mfma<9>, &pb::launch_mfma<10>};
  const int nb = (N + pb::MFMA_SPAN - 1) / pb::MFMA_SPAN;
  if (K < 1 || K > MFMA_K3 || (K > MFMA_K2 && extras) || N <= 128 || nb > 10) return nullptr;   // (129 .. 310 scans: 5 .. 10 blocks)
  return tab[nb - 5];
}
}  // namespace
namespace pb {
#define PB_WIDE(S, KT)                                                                            \
  extern template int launch_wide<S, KT>(const FistaArgs&, const double*, int, bool, int, hipStream_t); \
  extern template int launch_wide_pp<S, KT>(const FistaArgs&, int, hipStream_t);
#include "wide_table.inc"
#undef PB_WIDE
}  // namespace pb
namespace {

typedef int (*wide_launch_fn)(const pb::FistaArgs&, const double* taps, int K, bool with_j, int stop,
                              hipStream_t);
struct WideEntry {
  int S, KT;
  wide_launch_fn fn;
  fast_launch_pp_fn fn_pp;
};
#define PB_WIDE(S, KT) {S, KT, &pb::launch_wide<S, KT>, &pb::launch_wide_pp<S, KT>},
const WideEntry kWide[] = {
#include "wide_table.inc"
};
#undef PB_WIDE

typedef int (*exact_launch_fn)(const pb::FistaArgs&, const double* taps, int K, bool with_j, int stop,
                               hipStream_t);
struct ExactEntry {
  int S, KT;
  exact_launch_fn fn;
};
}  // namespace
namespace pb {
#define PB_EXACT(S, KT) \
  extern template int launch_exact<S, KT>(const FistaArgs&, const double*, int, bool, int, hipStream_t);
#include "exact_table.inc"
#undef PB_EXACT
}  // namespace pb
namespace {
#define PB_EXACT(S, KT) {S, KT, &pb::launch_exact<S, KT>},
const ExactEntry kExact[] = {
#include "exact_table.inc"
};
#undef PB_EXACT

// window lengths the register-resident forms carry (increment ring of wind - 2 slots in LDS)
inline bool ring_wind(int wind) { return wind == 4 || wind == 6 || wind == 8; }

// all-float64 register-resident form (one problem per wave): cheapest entry that holds (N, K)
const ExactEntry* pick_exact(int N, int K) {
  const ExactEntry* best = nullptr;
  const int s_need = (N + 63) / 64;
  for (const ExactEntry& e : kExact) {
    if (e.S < s_need || e.KT < K) continue;
    if (!best || (int64_t)e.S * e.KT < (int64_t)best->S * best->KT) best = &e;
  }
  return best;
}

const WideEntry* pick_wide(int N, int K) {
  const WideEntry* best = nullptr;
  const int s_need = (N + 63) / 64;
  for (const WideEntry& e : kWide) {
    if (e.S < s_need || e.KT < K) continue;
    if (!best || (int64_t)e.S * e.KT < (int64_t)best->S * best->KT) best = &e;
  }
  return best;
}

template <int S, int KT>
constexpr pair_launch_fn pair_or_null() {
  if constexpr (S <= 20 && KT <= 32) return &pb::launch_pair<S, KT>; else return nullptr;
}
template <int S, int KT>
constexpr pair_launch_fn pair_ffa_or_null() {
  if constexpr (S <= 20 && KT <= 32) return &pb::launch_pair_ffa<S, KT>; else return nullptr;
}
template <int S, int KT>
constexpr int (*pair_dev_or_null())(const pb::FistaArgs&, hipStream_t) {
  if constexpr (S <= 20 && KT <= 32) return &pb::launch_pair_ffa_dev<S, KT>; else return nullptr;
}
template <int S, int KT>
constexpr pair_cert_fn pair_cert_or_null() {
  if constexpr (S <= 20 && KT <= 32) return &pb::launch_pair_ffa_cert<S, KT>; else return nullptr;
}
template <int S, int KT>
constexpr pair_split_fn pair_split_or_null() {
  if constexpr (S <= 20 && KT <= 32) return &pb::launch_pair_ffa_split<S, KT>; else return nullptr;
}
#define PB_FAST(S, KT)                                                                             \
  {S, KT, &pb::launch_fast<S, KT>, &pb::launch_fast_pp<S, KT>, pair_or_null<S, KT>(),             \
   pair_ffa_or_null<S, KT>(), pair_dev_or_null<S, KT>(), pair_cert_or_null<S, KT>(),              \
   pair_split_or_null<S, KT>()},
const FastEntry kFast[] = {
#include "fast_table.inc"
};
#undef PB_FAST

const FastEntry* pick_fast(int N, int K) {
  const FastEntry* best = nullptr;
  const int s_need = (N + 15) / 16;
  for (const FastEntry& e : kFast) {
    if (e.S < s_need || e.KT < K) continue;
    if (!best || (int64_t)e.S * e.KT < (int64_t)best->S * best->KT) best = &e;
  }
  return best;
}

// series of 16 S < N <= 32 S scans: the pair form with the series' two halves in the slots of a row
const FastEntry* pick_split(int N, int K) {
  const FastEntry* best = nullptr;
  const FastEntry* whole = pick_fast(N, K);
  if (whole && whole->fn_pair_ffa) return nullptr;      // the series fits a slot: two PROBLEMS per row
  for (const FastEntry& e : kFast) {
    if (!e.fn_pair_split || e.KT < K || N <= 16 * e.S || N > 32 * e.S) continue;
    if (!best || (int64_t)e.S * e.KT < (int64_t)best->S * best->KT) best = &e;
  }
  return best;
}
// below this many series the one-problem-per-wave / single-row forms finish first (latency-bound)
constexpr int SPLIT_MIN_P = 1024;

// Plain solves (no stop rule) of a shape the matrix-pipe form serves AND some vector form can back up
// (remainders, re-solves of what its guards hand back): they go to the matrix pipe before the split-pair form
// is considered -- 305..310 scans fit ten blocks of 31 samples but have no single-slot pair entry, and one
// matrix-pipe wave beats the pair form over two slots.
bool mfma_serves_plain(int N, int K) {
  return pick_mfma(N, K) != nullptr && (pick_fast(N, K) != nullptr || pick_wide(N, K) != nullptr);
}
// without a single-row entry (305..310 scans and more than 32 taps) the remainder of the whole rounds goes to the
// one-problem-per-wave form when it is small, else everything runs on the matrix pipe (a partial last pass)
double wave_slots();
int mfma_wide_base(int P, bool one_launch) {
  const int round = (int)wave_slots() * 8;
  const int base = (P / round) * round;
  return (one_launch || P - base > round / 4) ? P : base;
}

// ---- launch plans: plan.h (host + device); here the host-side wrappers with this device's wave slots --------
using pb::Piece; using pb::Plan;
using pb::FORM_GENERIC; using pb::FORM_FAST1; using pb::FORM_PAIR; using pb::FORM_WIDE; using pb::FORM_MFMA; using pb::FORM_MFMA2;
using pb::MFMA2_MIN_R; using pb::MFMA2_BESIDE_CHUNKS;

double wave_slots() {
  static const double slots = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
      return 2048.0;
    return (double)prop.multiProcessorCount * 4.0 * 2.0;
  }();
  return slots;
}
int best_form(int P, bool has_pair, bool has_wide, double* cost = nullptr) { return pb::best_form(P, has_pair, has_wide, wave_slots(), cost); }
Plan plan_plain(int P, bool has_pair, bool has_wide, bool one_launch) { return pb::plan_plain(P, has_pair, has_wide, one_launch, wave_slots()); }

// ---- a side stream per device for remainders that fit BESIDE the main launch -----------
// When a plain solve has between one and two half rounds of pair waves (8 192 < P < 16 384
// problems on MI355X), a single launch leaves some SIMDs with two 8-problem waves and the
// others with one.  Measured (tools/conc_probe.py): half a round of pair waves (one per SIMD)
// on the caller's stream with the remainder on a second stream -- single-row waves (4
// problems) or one-problem waves -- co-schedules one wave of each per SIMD: 12 288 problems in
// 0.80 of a round instead of 0.92-1.0, 10 000 in 0.74.  The side stream forks from and joins
// back into the caller's stream with events (no host synchronisation); it is created on first
// use, one per device.  Not used while the caller's stream is being captured into a graph.  A
// remainder is never started beside a multi-round launch (measured slower: it unbalances the
// last round); after whole rounds the same group closes the plan (plan_pieces).
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool ok = false;
};
std::mutex g_side_mutex;

SideStream* side_stream_locked() {      // call with g_side_mutex held
  static std::map<int, SideStream> per_dev;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  SideStream& ss = per_dev[dev];
  if (!ss.ok && ss.stream == nullptr) {
    // Its own priority level: HIP multiplexes the streams of one priority onto a few hardware
    // queues, and a side stream that lands on the caller's queue runs BEHIND the caller's
    // kernel instead of beside it (seen after an application had created half a dozen streams)
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; (void)hipGetLastError(); }
    if (hipStreamCreateWithPriority(&ss.stream, hipStreamNonBlocking, greatest) == hipSuccess &&
        hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess)
      ss.ok = true;
    else
      (void)hipGetLastError();
  }
  return ss.ok ? &ss : nullptr;
}

bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return true;                         // unknown: stay on one stream
  }
  return cs != hipStreamCaptureStatusNone;
}

int plan_pieces(int P, bool has_pair, bool has_wide, bool one_launch, bool one_stream, Piece* out) {
  return pb::plan_pieces(P, has_pair, has_wide, one_launch, one_stream, wave_slots(), out);
}
// ... for series of ten blocks; shorter series take one chunk at most: a pass of the one-wave form is cheaper for them
// relative to a chunk (round 4, sum-slot kernel: N = 240, 10 000 problems 1.36 ms as split pass + two chunks against
// 1.21 ms as one pass of the one-wave form; N = 300: 1.52 against 1.59 -- profiles/r4_split_form_passes.txt)
inline int beside_chunks_for(int N) { return N > 9 * pb::MFMA_SPAN ? MFMA2_BESIDE_CHUNKS : 1; }
// Series of 311 .. 640 scans (10 .. 20 blocks) run on the split form from this many problems on (below, the
// pair form over two slots or the latency-bound one-problem-per-wave form finish first: N = 600, 4 096 problems
// 1.58 ms against 1.93, 8 192 2.87 against 1.97 -- profiles/r4_split_form_passes.txt); whole passes of 8 192
// problems, a remainder above 5/16 of a pass too, a smaller one on the one-problem-per-wave form.
constexpr int MFMA2_LONG_MIN_P = 5120;
// (HRFs of 34+ taps have no pair form to compete with, and the one-problem-per-wave form pays for every tap: N = 600, K = 42,
// 4 096 problems 2.31 ms on the split form against 4.61 -- profiles/r5_long_series_42_taps.txt)
inline int mfma2_long_min_p(int K) { return K > 33 ? 2048 : MFMA2_LONG_MIN_P; }
bool mfma2_serves_long(int N, int K, bool extras = true) { return N > MFMA1_NMAX && pick_mfma2(N, K, extras) != nullptr && pick_wide(N, K) != nullptr; }
// 225 .. 310 scans with 34+ taps and the window rule: the one-wave form has no certificate beside three near tiles (its state does
// not fit), the split form has -- it takes such calls like a long series
bool mfma2_takes_short_cert(int N, int K, int stop_mode, int wind) {     // (... and the _loops_deconv rule, which the one-wave form lacks there too)
  return K > 33 && ((stop_mode == PB_STOP_WINDOW && wind == 6) || stop_mode == PB_STOP_LOOPS) && N > 224 && N <= MFMA1_NMAX &&
         pick_mfma2(N, K, false) != nullptr && pick_wide(N, K) != nullptr;
}
int mfma2_long_base(int P, bool one_launch) {
  const int pass = (int)wave_slots() * 4;            // 16 problems x (slots / 2 SIMDs / 2 waves)
  const int base = (P / pass) * pass;
  return (one_launch || P - base > pass * 5 / 16) ? P : base;
}
// Series of 641 .. 1 280 scans on the four-wave form: a pass is 16 problems per compute unit (4 096 on 256 of them) whatever
// the batch; whole passes, a remainder above MFMA4_MIN_R of a pass too, a smaller one -- and batches below it -- on the
// one-problem-per-wave form
constexpr int MFMA4_MIN_R_NUM = 10, MFMA4_MIN_R_DEN = 16;   // (N = 1 200: 2 048 problems 1.89 ms against 2.31, 3 072 2.59 against 2.30 -- profiles/r5_long_series_1200_scans.txt)
bool mfma4_serves(int N, int K, bool extras = false) { return pick_mfma4(N, K, extras) != nullptr && pick_wide(N, K) != nullptr; }
int mfma4_base(int P, bool one_launch) {
  const int pass = (int)wave_slots() * 2;            // 16 problems x (slots / 2 per SIMD / 4 SIMDs per workgroup)
  const int base = (P / pass) * pass;
  return (one_launch || (int64_t)(P - base) * MFMA4_MIN_R_DEN > (int64_t)pass * MFMA4_MIN_R_NUM) ? P : base;
}
int plan_pieces_mfma(int P, bool has_pair, bool has_wide, bool one_launch, bool one_stream, bool has_mfma2, int beside_chunks, Piece* out) {
  return pb::plan_pieces_mfma(P, has_pair, has_wide, one_launch, one_stream, has_mfma2, beside_chunks, wave_slots(), out);
}

// ---- workspace of a partitioned solve (int32 units) -----------------------------------------------------------
//   [0, P) the lists   [P] length of the front list   [P+1, P+1+nblk) block counts   ranges of the three lists'
//   candidate launches   lambda_max of every series (float64, when the caller has none)
struct WorkLayout { int64_t ranges, lmax, total; };
WorkLayout work_layout(int P, int V) {
  const int64_t nblk = ((int64_t)P + pb::PATH_PER_BLOCK - 1) / pb::PATH_PER_BLOCK;
  WorkLayout w;
  w.ranges = ((int64_t)P + 2 + 2 * nblk + 8 + 1) & ~(int64_t)1;   // (list, n_front, front / ill counts per block, n_ill: path.h)
  w.lmax = w.ranges + 3 * 2 * pb::CAND_COUNT + 4;           // (3 x candidate ranges + the ill range; even: 8-byte aligned when the buffer is)
  w.total = w.lmax + 2 * (int64_t)V + 8;
  return w;
}
// coherence bounds of the conditioning guard (path.h: path_class; calibrated on 5 120 series per length,
// profiles/r5_gamma_calibration_*.txt: above them the matrix-pipe form holds 3.3e-6 and the float32 vector forms 3e-6)
constexpr double PART_GAMMA_F64 = 1.0e-2, PART_GAMMA_MATRIX_PIPE = 7.0e-2;
// The bound below which a series stays off the matrix pipe, by shape: the error of the matrix-pipe forms at a given coherence falls
// with the length of the series (profiles/r5_gamma_calibration_*.txt, worst over the adversarial families per bin of gamma_2:
// 300 scans 4.8e-6 in [5e-2, 7e-2) and 6.7e-6 below; 600 scans 3.7e-6 in [3e-2, 5e-2), 4.9e-6 in [2e-2, 3e-2); 1 200 scans 5.4e-6 in
// [2e-2, 3e-2), 4.5e-6 in [1e-2, 2e-2); with 34+ taps 8.2e-6 in [5e-2, 7e-2) at 300 scans) -- and white noise, whose own error is
// 2e-6 at most, has a median gamma_2 of 6e-2 / 4e-2 / 3e-2 at 300 / 600 / 1 200 scans: one bound for every length kept nearly
// every noise-like series of 1 200 scans on the vector forms.
inline double part_gamma_matrix_pipe(int N, int K) {
  if (K > 33 || N <= 310) return PART_GAMMA_MATRIX_PIPE;
  return N <= 640 ? 3.0e-2 : 2.0e-2;
}
// below this many problems a call is latency-bound and keeps the host-side plan (a partition costs ~8 small launches)
constexpr int PART_MIN_P = 4096;

// pb_fista_solve without a caller's workspace: one buffer per (device, stream), grown on demand, never freed
int32_t* own_workspace(void* stream, int64_t need, int64_t* len) {
  static std::mutex mu;
  static std::map<std::pair<int, void*>, std::pair<int32_t*, int64_t>> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> lock(mu);
  auto& e = cache[std::make_pair(dev, stream)];
  if (e.second < need) {
    int32_t* fresh = nullptr;
    const int64_t want = need + need / 4;
    if (hipMalloc((void**)&fresh, (size_t)want * sizeof(int32_t)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    // (the old buffer may still be read by work in flight on this stream: it is kept, not freed -- buffers grow by a
    // quarter at least, so what is retired over a process's life is a small multiple of the largest one)
    e.first = fresh;
    e.second = want;
  }
  *len = e.second;
  return e.first;
}

template <int KIND>
int launch_op(const double* x, int64_t ldx, double* out, int64_t ldo, int V, int n_src, int n_dst,
              const double* taps, int K, void* stream, const char* name) {
  // sizes first; an empty batch is a no-op whatever the pointers are (torch hands out
  // data_ptr() == 0 for zero-element tensors)
  if (V < 0 || n_src < 1 || n_dst < 1 || K < 0)
    return fail(PB_ERR_INVALID, "%s: bad size (V=%d n_src=%d n_dst=%d K=%d)", name, V, n_src, n_dst, K);
  if (ldx < n_src || ldo < n_dst) return fail(PB_ERR_INVALID, "%s: leading dimension too small", name);
  const int nmax = n_src > n_dst ? n_src : n_dst;
  if (2 * (int64_t)nmax + K + 8 > LDS_DOUBLES_MAX)
    return fail(PB_ERR_INVALID, "%s: row of %d with %d taps exceeds LDS", name, nmax, K);
  if (V == 0) return PB_OK;
  if (!x || !out || (K > 0 && !taps)) return fail(PB_ERR_INVALID, "%s: NULL pointer", name);
  const size_t lds = (size_t)(2 * nmax + K + 8) * sizeof(double);
  hipLaunchKernelGGL((pb::op_kernel<KIND>), dim3(V), dim3(pb::GEN_THREADS), lds, (hipStream_t)stream,
                     x, ldx, out, ldo, n_src, n_dst, taps, K);
  return check_launch(name);
}

}  // namespace

namespace {
template <typename TY>
int stats_impl(const double* w_dev, int64_t ldw, const TY* y_dev, int64_t ldy, int y_rep, int P,
               int N, const double* taps_dev, int K, double* r2_dev, double* l1_dev, void* stream,
               const char* name) {
  if (P < 0 || N < 1 || K < 1 || y_rep < 1 || ldw < N || ldy < N)
    return fail(PB_ERR_INVALID, "%s: bad size", name);
  if (2 * (int64_t)N + K + 8 > LDS_DOUBLES_MAX)
    return fail(PB_ERR_INVALID, "%s: N=%d K=%d exceeds LDS", name, N, K);
  if (P == 0) return PB_OK;
  if (!w_dev || !y_dev || !taps_dev || !r2_dev || !l1_dev)
    return fail(PB_ERR_INVALID, "%s: NULL pointer", name);
  const size_t lds = (size_t)(2 * N + K + 8) * sizeof(double);
  hipLaunchKernelGGL((pb::stats_kernel<TY>), dim3(P), dim3(pb::GEN_THREADS), lds,
                     (hipStream_t)stream, w_dev, ldw, y_dev, ldy, y_rep, N, taps_dev, K, r2_dev,
                     l1_dev);
  return check_launch(name);
}

template <typename TY>
int hrf_cost_impl(const double* z_dev, int64_t ldz, const TY* y_dev, int64_t ldy, int V, int N,
                  const double* taps_dev, int K, int n_hrf, double* cost_dev, int per_voxel,
                  void* stream, const char* name) {
  if (V < 0 || N < 1 || K < 1 || n_hrf < 1 || ldz < N || ldy < N)
    return fail(PB_ERR_INVALID, "%s: bad size", name);
  if (n_hrf > 65535) return fail(PB_ERR_INVALID, "%s: more than 65535 candidate HRFs", name);
  if ((int64_t)N + K + 8 > LDS_DOUBLES_MAX) return fail(PB_ERR_INVALID, "%s: exceeds LDS", name);
  if (V == 0) return PB_OK;
  if (!z_dev || !y_dev || !taps_dev || !cost_dev) return fail(PB_ERR_INVALID, "%s: NULL pointer", name);
  const size_t lds = (size_t)(N + K + 8) * sizeof(double);
  hipLaunchKernelGGL((pb::hrf_cost_kernel<TY>), dim3(V, n_hrf), dim3(pb::GEN_THREADS), lds,
                     (hipStream_t)stream, z_dev, ldz, y_dev, ldy, V, N, taps_dev, K, cost_dev,
                     per_voxel);
  return check_launch(name);
}
}  // namespace

namespace {
template <typename TY>
int lambda_max_impl(const TY* y_dev, int64_t ldy, int V, int N, const double* taps_dev, int K,
                    double* out_dev, void* stream, const char* name) {
  if (V < 0 || N < 1 || K < 1 || ldy < N) return fail(PB_ERR_INVALID, "%s: bad size", name);
  if (2 * (int64_t)N + K + 8 > LDS_DOUBLES_MAX)
    return fail(PB_ERR_INVALID, "%s: N=%d K=%d exceeds LDS", name, N, K);
  if (V == 0) return PB_OK;
  if (!y_dev || !taps_dev || !out_dev) return fail(PB_ERR_INVALID, "%s: NULL pointer", name);
  const size_t lds = (size_t)(2 * N + K + 8) * sizeof(double);
  hipLaunchKernelGGL((pb::lambda_max_kernel<TY>), dim3(V), dim3(pb::GEN_THREADS), lds,
                     (hipStream_t)stream, y_dev, ldy, N, taps_dev, K, out_dev);
  return check_launch(name);
}

constexpr int NE_MAX_BLOCKS = 2048;
constexpr int NE_WAVE_BLOCKS = 1024;           // one-voxel-per-wave form: one resident pass (4 workgroups per CU)

// the one-voxel-per-wave form serves K <= 32 (tail entries in registers) while four staging areas fit
inline bool ne_wave_form(int N, int K) {
#ifdef PB_DEVELOPMENT                            // A/B aid of development builds only: the release library reads no environment
  if (getenv("PB_NE_BLOCK_FORM")) return false;
#endif
  return K <= 32 && (int64_t)pb::ne_wave_lds_doubles(N, K) <= LDS_DOUBLES_MAX;
}
inline int ne_wave_blocks(int V, int cap) {
  int b = (V + pb::GEN_WAVES - 1) / pb::GEN_WAVES;
  if (b > NE_WAVE_BLOCKS) b = NE_WAVE_BLOCKS;
  return b < cap ? b : cap;
}

template <typename TY>
int normal_eq_impl(const double* z_dev, int64_t ldz, const TY* y_dev, int64_t ldy, int V, int N,
                   int K, int per_voxel, double* work_dev, int64_t work_len, double* out_dev,
                   void* stream, const char* name) {
  if (V < 0 || N < 1 || K < 1 || K > 127 || ldz < N || ldy < N)
    return fail(PB_ERR_INVALID, "%s: bad size (V=%d N=%d K=%d)", name, V, N, K);
  const int ne = pb::ne_len(K);
  int sub_log2 = 0;                            // lanes per role: 4 for K <= 31, 2 for K <= 63
  while ((2 * K + 1) << (sub_log2 + 1) <= pb::NE_THREADS && sub_log2 < 2) ++sub_log2;
  const int64_t nd = per_voxel ? 2 * (int64_t)N : (int64_t)pb::ne_sum_lds_doubles(N, K);
  if (nd > LDS_DOUBLES_MAX) return fail(PB_ERR_INVALID, "%s: N=%d K=%d exceeds LDS", name, N, K);
  const size_t lds = (size_t)nd * sizeof(double);
  if (per_voxel) {
    if (V == 0) return PB_OK;
    if (!z_dev || !y_dev || !out_dev) return fail(PB_ERR_INVALID, "%s: NULL pointer", name);
    hipLaunchKernelGGL((pb::normal_eq_kernel<TY>), dim3(V < 65536 ? V : 65536),
                       dim3(pb::NE_THREADS), lds, (hipStream_t)stream, z_dev, ldz, y_dev, ldy, V, N, K,
                       sub_log2, out_dev);
    return check_launch(name);
  }
  // shared mode: block partials in work_dev, then a fixed-order sum (an empty shard yields
  // zeros, so that the rank still contributes to the all-reduce)
  if (!out_dev) return fail(PB_ERR_INVALID, "%s: NULL output", name);
  int blocks = V < NE_MAX_BLOCKS ? V : NE_MAX_BLOCKS;
  if (work_len / ne < blocks) blocks = (int)(work_len / ne);
  if (V > 0) {
    if (blocks < 1 || !work_dev)
      return fail(PB_ERR_INVALID, "%s: work buffer must hold at least %d doubles", name, ne);
    if (!z_dev || !y_dev) return fail(PB_ERR_INVALID, "%s: NULL pointer", name);
    if (ne_wave_form(N, K)) {                  // one voxel per wave (blind.h)
      blocks = ne_wave_blocks(V, blocks);
      hipLaunchKernelGGL((pb::normal_eq_wave_kernel<TY, false>), dim3(blocks), dim3(pb::NE_THREADS),
                         (size_t)pb::ne_wave_lds_doubles(N, K) * sizeof(double), (hipStream_t)stream, z_dev,
                         ldz, y_dev, ldy, V, N, K, work_dev);
    } else {
      hipLaunchKernelGGL((pb::normal_eq_sum_kernel<TY>), dim3(blocks), dim3(pb::NE_THREADS),
                         (size_t)pb::ne_sum_lds_doubles(N, K) * sizeof(double), (hipStream_t)stream, z_dev,
                         ldz, y_dev, ldy, V, N, K, work_dev);
    }
  } else {
    blocks = 0;
  }
  hipLaunchKernelGGL(pb::normal_eq_reduce_kernel, dim3(ne), dim3(pb::NE_THREADS), 0,
                     (hipStream_t)stream, work_dev, blocks, ne, out_dev);
  return check_launch(name);
}
}  // namespace

extern "C" {

int pb_version(void) { return 100; }

int pb_init(void) {
  std::lock_guard<std::mutex> lock(g_side_mutex);
  if (!side_stream_locked()) return fail(PB_ERR_HIP, "pb_init: no side stream (is a HIP device current?)");
  g_err[0] = 0;
  return PB_OK;
}

const char* pb_last_error(void) { return g_err; }

int pb_fista_has_fast_path(int N, int K) {
  return (N >= 1 && K >= 1 && (pick_fast(N, K) || pick_wide(N, K))) ? 1 : 0;
}

// the one-problem-per-wave entry worth using for SHORT series (the cheapest-per-problem tail
// form): only entries whose strips are at most 8 samples
static const WideEntry* pick_wide_small(int N, int K) {
  const WideEntry* we = pick_wide(N, K);
  return (we && we->S <= 8) ? we : nullptr;
}

// the pair form carries plain solves and, as a no-fire certificate with an exact re-solve of what
// it cannot clear, the window rule at the reference's wind = 6 (the queries assume the tolerance
// is small enough for that path: tol * n_iter < 0.5, see pb_fista_solve)
static bool pair_carries(const FastEntry* fe, int stop_mode, int wind) {
  if (stop_mode == PB_STOP_NONE) return fe->fn_pair != nullptr;
  return stop_mode == PB_STOP_WINDOW && wind == 6 && fe->fn_pair_cert != nullptr;
}

int pb_fista_which_kernel(int N, int K, int P, int with_cost_trace, int stop_mode, int wind) {
  if (N < 1 || K < 1 || P < 1) return 0;
  const bool mfma_plain = stop_mode == PB_STOP_NONE && mfma_serves_plain(N, K);
  const bool split_shape = stop_mode == PB_STOP_NONE || (stop_mode == PB_STOP_WINDOW && wind == 6) || (stop_mode == PB_STOP_LOOPS && !with_cost_trace);
  const bool mfma2_ok = (stop_mode == PB_STOP_NONE || (stop_mode == PB_STOP_WINDOW && wind == 6)) && pick_mfma2(N, K) != nullptr;
  if (split_shape && mfma4_serves(N, K, false) && (stop_mode != PB_STOP_WINDOW || pick_wide(N, K)->S <= 20))
    return mfma4_base(P, false) > 0 ? pb::FORM_MFMA4 : FORM_WIDE;
  if (split_shape && (mfma2_serves_long(N, K, false) || mfma2_takes_short_cert(N, K, stop_mode, wind)) && P >= mfma2_long_min_p(K) && (stop_mode != PB_STOP_WINDOW || pick_wide(N, K)->S <= 20))
    return mfma2_long_base(P, false) > 0 ? FORM_MFMA2 : ((pick_fast(N, K) && N <= 320) ? FORM_FAST1 : FORM_WIDE);   // (the solve's own backup form)
  if (const FastEntry* se = pick_split(N, K))
    if (!mfma_plain && P >= SPLIT_MIN_P && pair_carries(se, stop_mode, wind) && (stop_mode == PB_STOP_NONE || pick_wide(N, K)))
      return FORM_PAIR;
  const FastEntry* fe = pick_fast(N, K);
  if (!fe && mfma_plain) return mfma_wide_base(P, false) > 0 ? FORM_MFMA : FORM_WIDE;
  if (fe && (pair_carries(fe, stop_mode, wind) || stop_mode == PB_STOP_NONE || (stop_mode == PB_STOP_LOOPS && !with_cost_trace)) &&
      pick_mfma(N, K, stop_mode != PB_STOP_NONE)) {   // plain solves, the window-rule certificate, the _loops_deconv rule
    Piece pc[6];
    plan_pieces_mfma(P, fe->fn_pair != nullptr && stop_mode != PB_STOP_LOOPS, pick_wide_small(N, K) != nullptr, false, false, mfma2_ok, beside_chunks_for(N), pc);
    return pc[0].form;
  }
  if (fe && stop_mode == PB_STOP_WINDOW && (!ring_wind(wind) || fe->S > 20)) fe = nullptr;
  if (!fe) {
    const WideEntry* we = pick_wide(N, K);
    if (we && stop_mode == PB_STOP_WINDOW && (!ring_wind(wind) || we->S > 20)) we = nullptr;
    return we ? 3 : 0;
  }
  Piece pc[4];
  plan_pieces(P, pair_carries(fe, stop_mode, wind), pick_wide_small(N, K) != nullptr, false, false, pc);
  return pc[0].form;                                    // the form that carries most problems
}

int pb_fista_plan(int N, int K, int P, int stop_mode, int wind, int* n_main, int* main_form,
                  int* tail_form) {
  return pb_fista_plan_ex(N, K, P, stop_mode, wind, 0u, n_main, main_form, tail_form);
}

int pb_fista_plan_ex(int N, int K, int P, int stop_mode, int wind, unsigned flags, int* n_main,
                     int* main_form, int* tail_form) {
  int nm = 0, mf = 0, tf = 0;
  const bool no_mfma = (flags & (PB_FLAG_NO_MFMA | PB_FLAG_NO_PAIR | PB_FLAG_FORCE_PAIR | PB_FLAG_DIRECT_FIR)) != 0;
  const FastEntry* se = (N >= 1 && K >= 1 && P >= SPLIT_MIN_P) ? pick_split(N, K) : nullptr;
  const bool mfma_plain = N >= 1 && K >= 1 && stop_mode == PB_STOP_NONE && !no_mfma && mfma_serves_plain(N, K);
  const bool mfma2_ok = N >= 1 && K >= 1 && (stop_mode == PB_STOP_NONE || (stop_mode == PB_STOP_WINDOW && wind == 6)) && !no_mfma &&
                        pick_mfma2(N, K) != nullptr;
  const bool split_shape = stop_mode == PB_STOP_NONE || (stop_mode == PB_STOP_WINDOW && wind == 6) || stop_mode == PB_STOP_LOOPS;
  if (N >= 1 && K >= 1 && P >= 1 && split_shape && !no_mfma &&
      mfma4_serves(N, K, false) && (stop_mode != PB_STOP_WINDOW || pick_wide(N, K)->S <= 20)) {
    const int base = mfma4_base(P, (flags & (PB_FLAG_ONE_LAUNCH | PB_FLAG_FORCE_MFMA2)) != 0);
    if (base > 0 && base < P) { nm = base; mf = pb::FORM_MFMA4; tf = FORM_WIDE; }
    else tf = base > 0 ? pb::FORM_MFMA4 : FORM_WIDE;
  } else if (N >= 1 && K >= 1 && split_shape && !no_mfma && (mfma2_serves_long(N, K, false) || mfma2_takes_short_cert(N, K, stop_mode, wind)) && (P >= mfma2_long_min_p(K) || (flags & PB_FLAG_FORCE_MFMA2)) &&
      (stop_mode != PB_STOP_WINDOW || pick_wide(N, K)->S <= 20)) {
    const int base = mfma2_long_base(P, (flags & (PB_FLAG_ONE_LAUNCH | PB_FLAG_FORCE_MFMA2)) != 0);
    const int backup_form = (pick_fast(N, K) && N <= 320) ? FORM_FAST1 : FORM_WIDE;      // what pb_fista_solve uses behind the split form
    if (base > 0 && base < P) { nm = base; mf = FORM_MFMA2; tf = backup_form; }
    else tf = base > 0 ? FORM_MFMA2 : backup_form;
  } else if (mfma2_ok && (flags & PB_FLAG_FORCE_MFMA2) && P >= 1) {
    tf = FORM_MFMA2;
  } else if (se && !mfma_plain && pair_carries(se, stop_mode, wind) && (stop_mode == PB_STOP_NONE || pick_wide(N, K))) {
    tf = FORM_PAIR;                                     // one launch of the split pair form
  } else if (mfma_plain && P >= 1 && !pick_fast(N, K)) {
    const int base = mfma_wide_base(P, (flags & PB_FLAG_ONE_LAUNCH) != 0);
    if (base > 0 && base < P) { nm = base; mf = FORM_MFMA; tf = FORM_WIDE; }
    else tf = base > 0 ? FORM_MFMA : FORM_WIDE;
  } else if (N >= 1 && K >= 1 && P >= 1 && !no_mfma && pick_fast(N, K) &&
             (pair_carries(pick_fast(N, K), stop_mode, wind) || stop_mode == PB_STOP_NONE || stop_mode == PB_STOP_LOOPS) &&
             pick_mfma(N, K, stop_mode != PB_STOP_NONE)) {
    Piece pc[6];
    const int npc = plan_pieces_mfma(P, pick_fast(N, K)->fn_pair != nullptr && stop_mode != PB_STOP_LOOPS, pick_wide_small(N, K) != nullptr,
                                     (flags & PB_FLAG_ONE_LAUNCH) != 0, (flags & PB_FLAG_ONE_STREAM) != 0, mfma2_ok, beside_chunks_for(N), pc);
    int i = 1;
    while (i < npc && pc[i].form == pc[0].form) ++i;
    if (i < npc) {
      nm = pc[i - 1].p1;
      mf = pc[0].form;
      int big = i;
      for (int k = i + 1; k < npc; ++k)
        if (pc[k].p1 - pc[k].p0 > pc[big].p1 - pc[big].p0) big = k;
      tf = pc[big].form;
    } else { nm = 0; mf = 0; tf = pc[0].form; }
  } else if (N >= 1 && K >= 1 && P >= 1) {
    const FastEntry* fe = pick_fast(N, K);
    if (fe && stop_mode == PB_STOP_WINDOW && (!ring_wind(wind) || fe->S > 20)) fe = nullptr;
    if (fe) {
      Piece pc[4];
      const int npc = plan_pieces(P, pair_carries(fe, stop_mode, wind),
                                  pick_wide_small(N, K) != nullptr, false, false, pc);
      // leading pieces of one form = the "main" part; the first other form = the tail
      // (of several: the one that carries most of the remaining problems)
      int i = 1;
      while (i < npc && pc[i].form == pc[0].form) ++i;
      if (i < npc) {
        nm = pc[i - 1].p1;
        mf = pc[0].form;
        int big = i;
        for (int k = i + 1; k < npc; ++k)
          if (pc[k].p1 - pc[k].p0 > pc[big].p1 - pc[big].p0) big = k;
        tf = pc[big].form;
      } else { nm = 0; mf = 0; tf = pc[0].form; }
    } else {
      const WideEntry* we = pick_wide(N, K);
      if (we && stop_mode == PB_STOP_WINDOW && (!ring_wind(wind) || we->S > 20)) we = nullptr;
      tf = we ? FORM_WIDE : FORM_GENERIC;
    }
  }
  if (n_main) *n_main = nm;
  if (main_form) *main_form = mf;
  if (tail_form) *tail_form = tf;
  return PB_OK;
}

static int solve_impl(const float* y_dev, int64_t ldy, int y_rep, double* w_dev, int64_t ldw, int P,
                      int N, const double* taps_host, const double* taps_dev, int K, double step,
                      double lbda,
                      const double* lbda_dev, const double* betas_dev, int n_iter, float* J_dev,
                      int64_t ldj, int stop_mode, double tol, int wind, int32_t* n_done_dev,
                      unsigned flags, void* stream, const double* lmax_dev, double dense_ratio,
                      int32_t* work_dev, int64_t work_len) {
  if (P < 0 || N < 1 || K < 1 || n_iter < 0 || y_rep < 1)
    return fail(PB_ERR_INVALID, "pb_fista_solve: bad size (P=%d N=%d K=%d n_iter=%d y_rep=%d)", P,
                N, K, n_iter, y_rep);
  if (P > 0 && (!y_dev || !w_dev || !taps_host || (n_iter > 0 && !betas_dev)))
    return fail(PB_ERR_INVALID, "pb_fista_solve: NULL pointer");
  if (P > (1 << 25)) return fail(PB_ERR_INVALID, "pb_fista_solve: more than 2^25 problems per launch");
  if (ldy < N || ldw < N) return fail(PB_ERR_INVALID, "pb_fista_solve: leading dimension < N");
  if (J_dev && ldj < n_iter) return fail(PB_ERR_INVALID, "pb_fista_solve: ldj < n_iter");
  if (!(step > 0.0)) return fail(PB_ERR_INVALID, "pb_fista_solve: step must be positive");
  // the reference's prox with a negative threshold grows every entry (its lambda search gets there); only
  // pb_fista_solve_d restates that -- these kernels clamp
  if (!lbda_dev && lbda < 0.0) return fail(PB_ERR_INVALID, "pb_fista_solve: negative lbda (float64 entry point only)");
  if (stop_mode < PB_STOP_NONE || stop_mode > PB_STOP_WINDOW)
    return fail(PB_ERR_INVALID, "pb_fista_solve: unknown stop_mode %d", stop_mode);
  if (stop_mode == PB_STOP_WINDOW && wind < 2)
    return fail(PB_ERR_INVALID, "pb_fista_solve: wind must be >= 2");
  if (P == 0) return PB_OK;

  pb::FistaArgs a;
  a.y = y_dev; a.y64 = nullptr; a.ldy = ldy; a.w = w_dev; a.ldw = ldw; a.lbda_vec = lbda_dev;
  a.betas = betas_dev; a.J = J_dev; a.J64 = nullptr; a.ldj = ldj; a.n_done = n_done_dev;
  a.step = step; a.lbda = lbda; a.tol = tol;
  a.y_rep = y_rep; a.P = P; a.N = N; a.n_iter = n_iter; a.stop_mode = stop_mode;
  a.taps_pp = nullptr; a.ldt = 0; a.step_vec = nullptr; a.step_shared = 0; a.K = K; a.p0 = 0;
  a.cold = (flags & PB_FLAG_COLD_START) ? 1 : 0;
  a.rho_guard = (flags & PB_FLAG_NO_RHO_GUARD) ? 0 : 1;
  a.wind = wind;
#ifdef PB_DEVELOPMENT                            // (development builds only: the series scale of the matrix-pipe form)
  if (const char* yb = getenv("PB_MFMA_YBITS")) {
    const int v = atoi(yb);
    if (v >= 8 && v <= 15) a.ybits = v;
  }
#endif


  // ---- round 5: partition BEFORE solving (dense class -> matrix pipe, sparse class -> float32 vector forms) and a
  // compacted re-solve of what a guard or certificate hands back; list lengths and launch plans live on the device
  // (path.h, plan.h).  For every call a matrix-pipe form would carry: one lambda or one per problem, cost trace,
  // window-rule certificate, _loops_deconv rule; series of 129..310 scans (fista_mfma_kernel) and of 311..640
  // (fista_mfma2_kernel).  Without it a batch whose lambda lies near lambda_max was solved twice -- matrix pipe, then one
  // handed-back problem per wave (profiles/r4_path_partition.txt: 2.05 against 3.20e9).
  //   launch_form(form, args, stream, exact_rule) -> 0 / 1 (rejected);  has_form(form, list) says which candidates exist
  //   (list 0: the call's one list -- matrix-pipe forms for its dense head, vector forms for the rest --, 1 / 2: the
  //   measurement aids "matrix-pipe candidates only" / "vector candidates only", 3: handed-back problems);
  //   bound(cand) = grid bound of a candidate in slots
  const int V_series = (P + y_rep - 1) / y_rep;
  const bool part_ws = work_dev && P >= PART_MIN_P && K <= pb::LMAX_KT && N <= 1280 && n_done_dev &&
                       work_len >= work_layout(P, V_series).total &&
                       !(flags & (PB_FLAG_FORCE_GENERIC | PB_FLAG_FORCE_PAIR | PB_FLAG_FORCE_WIDE | PB_FLAG_NO_PAIR | PB_FLAG_DIRECT_FIR | PB_FLAG_NO_MFMA |
                                  PB_FLAG_ONE_LAUNCH | PB_FLAG_FORCE_MFMA2 | PB_FLAG_CERT_NO_RESOLVE | PB_FLAG_NO_PARTITION)) &&
                       !((flags & PB_FLAG_FORCE_MFMA) && !lmax_dev);      // ("everything on the matrix pipe", as before)
  // (the ill-conditioned class: register-resident float64 kernel where the shape has an entry; its window rule is wind = 6)
  auto ill_exact = [&]() -> const ExactEntry* {
    const ExactEntry* e = pick_exact(N, K);
    return (e && stop_mode == PB_STOP_WINDOW && wind != 6) ? nullptr : e;
  };
  bool no_dense_class = false;                     // a partitioned call without a matrix-pipe form: every problem in the vector class
  auto run_partition = [&](const pb::PlanSpec& dense, const pb::PlanSpec& sparse, const pb::PlanSpec& flagged,
                           auto&& launch_form, auto&& has_form, auto&& bound) -> int {
    const WorkLayout wl = work_layout(P, V_series);
    hipStream_t user = (hipStream_t)stream;
    // lambda_max of every series (the caller's lmax_dev is not needed any more: the pass also sees max|y| and marks the
    // ill-conditioned series, which the caller's numbers do not tell) -- float32, ~55 us per 100 k series
    double* lm = reinterpret_cast<double*>(work_dev + wl.lmax);
    {
      pb::LmaxTaps lt;
      double run = 0.0, csum = 0.0;                   // sum|c| over the N lags of the operator's step response c = cumsum(h)
      for (int k = 0; k < pb::LMAX_KT; ++k) lt.h[k] = k < K ? (float)taps_host[k] : 0.0f;
      for (int t = 0; t < N; ++t) { if (t < K) run += taps_host[t]; csum += std::fabs(run); }
      // (the marked series: float64 register-resident kernel, or the LDS one, which needs the taps in device memory and the row in LDS)
      const int64_t nd_g = 3 * (int64_t)N + K + 2 * pb::GEN_WAVES + (stop_mode == PB_STOP_WINDOW ? (int64_t)wind * N : 0);
      const bool guard = !(flags & PB_FLAG_NO_ILL_GUARD);
      const double unit = csum / std::sqrt((double)N);
      const float f64_bound = (guard && (ill_exact() || (taps_dev && nd_g <= LDS_DOUBLES_MAX))) ? (float)(PART_GAMMA_F64 * unit) : 0.0f;
      const float vec_bound = guard ? (float)(part_gamma_matrix_pipe(N, K) * unit) : 0.0f;
      const dim3 grid((unsigned)((V_series + 3) / 4)), block(256);
      if (N <= 320) hipLaunchKernelGGL((pb::lmax_wave_kernel<5>), grid, block, 4 * (64 * 5 + pb::LMAX_KT) * sizeof(float), user, y_dev, ldy, V_series, N, lt, K, lm, f64_bound, vec_bound);
      else if (N <= 640) hipLaunchKernelGGL((pb::lmax_wave_kernel<10>), grid, block, 4 * (64 * 10 + pb::LMAX_KT) * sizeof(float), user, y_dev, ldy, V_series, N, lt, K, lm, f64_bound, vec_bound);
      else hipLaunchKernelGGL((pb::lmax_wave_kernel<21>), grid, block, 4 * (64 * 21 + pb::LMAX_KT) * sizeof(float), user, y_dev, ldy, V_series, N, lt, K, lm, f64_bound, vec_bound);   // (odd strips: conflict-free)
    }
    const double* lmax = lm;
    const int nblk = (P + pb::PATH_PER_BLOCK - 1) / pb::PATH_PER_BLOCK;
    int32_t* rg_dense = work_dev + wl.ranges;
    int32_t* rg_sparse = rg_dense + 2 * pb::CAND_COUNT;
    int32_t* rg_flag = rg_sparse + 2 * pb::CAND_COUNT;
    int32_t* rg_ill = rg_flag + 2 * pb::CAND_COUNT;
    {
      pb::ClassPred cp{lbda_dev, lbda, lmax, y_rep, no_dense_class ? 0.0 : (dense_ratio > 0.0 ? dense_ratio : (N > 640 ? PB_PATH_DENSE_RATIO_LONGER : (N > MFMA1_NMAX ? PB_PATH_DENSE_RATIO_LONG : PB_PATH_DENSE_RATIO))), nullptr};
      hipLaunchKernelGGL(pb::path_count_kernel, dim3(nblk), dim3(pb::PATH_THREADS), 0, user, cp, P, work_dev);
      pb::PlanSpec front = dense;
      front.merged = 1;
      hipLaunchKernelGGL(pb::path_scan_kernel, dim3(1), dim3(pb::PATH_THREADS), 0, user, P, nblk, work_dev, front, sparse, rg_dense, rg_sparse, rg_ill);
      hipLaunchKernelGGL(pb::path_scatter_kernel, dim3(nblk), dim3(pb::PATH_THREADS), 0, user, cp, P, work_dev);
      const int rc = check_launch("partition");
      if (rc != PB_OK) return rc;
    }
    SideStream* ss = nullptr;
    std::unique_lock<std::mutex> lock(g_side_mutex, std::defer_lock);
    if (!dense.one_stream) {
      lock.lock();
      ss = side_stream_locked();
    }
    // a whole list: its candidates in their static order, the side-stream ones forked after the whole rounds
    auto solve_list = [&](const int32_t* ranges, int side, int list, bool exact_rule) -> int {
      int rc = PB_OK;
      bool forked = false;
      for (int c = 0; c < pb::CAND_COUNT && rc == PB_OK; ++c) {
        const int form = pb::cand_form(c);
        if (!has_form(form, list) || (pb::cand_side(c) && !ss)) continue;
        // (a plan on one stream has no groups, and a second piece of a form continues the first: plan_to_candidates merges them)
        if ((c == pb::CAND_PAIR1 || c == pb::CAND_FAST1) && !ss) continue;
        const int bd = bound(c);
        if (bd <= 0) continue;
        if (c >= pb::CAND_FIRST_AFTER_FORK && !forked && ss) {
          if (hipEventRecord(ss->fork, user) != hipSuccess || hipStreamWaitEvent(ss->stream, ss->fork, 0) != hipSuccess)
            return fail(PB_ERR_HIP, "pb_fista_solve: fork to the side stream failed");
          forked = true;
        }
        pb::FistaArgs b = a;
        b.perm = work_dev;
        b.n_dense = work_dev + P;
        b.perm_side = side;
        b.range = ranges + 2 * c;
        b.grid_slots = bd;
        if (launch_form(form, b, (pb::cand_side(c) && ss) ? ss->stream : user, exact_rule) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: a kernel form rejected its launch over a device-side list (form %d)", form);
        rc = check_launch("fista kernel (device-side list)");
      }
      if (forked && (hipEventRecord(ss->join, ss->stream) != hipSuccess || hipStreamWaitEvent(user, ss->join, 0) != hipSuccess))
        return fail(PB_ERR_HIP, "pb_fista_solve: join of the side stream failed");
      return rc;
    };
    // ONE list for the call: positions [0, P) of the list array (dense problems first), one plan (plan.h: plan_partitioned)
    const bool only_dense = (flags & PB_FLAG_ONLY_DENSE) != 0, only_sparse = (flags & PB_FLAG_ONLY_SPARSE) != 0;   // (measurement aids)
    int rc = solve_list(rg_dense, 1, only_dense ? 1 : (only_sparse ? 2 : 0), false);
    if (rc != PB_OK || only_dense || only_sparse) return rc;
    // the ill-conditioned series (marked by the lambda_max pass, the tail of the list array): float64 LDS kernel, any stop
    // rule; its workgroups stride over the list, which is empty for ordinary data
    {
      pb::FistaArgs b = a;
      b.perm = work_dev;
      b.perm_side = 1;
      b.range = rg_ill;
      const int64_t nd_g = 3 * (int64_t)N + K + 2 * pb::GEN_WAVES + (stop_mode == PB_STOP_WINDOW ? (int64_t)wind * N : 0);
      // (the register-resident float64 kernel where the shape has an entry: 1.0e9 voxel-iterations/s against 0.17e9)
      const ExactEntry* ee = ill_exact();
      if (ee) {
        b.grid_slots = P;
        if (ee->fn(b, taps_host, K, J_dev != nullptr, stop_mode, user) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: float64 kernel rejected the launch (ill-conditioned series)");
        rc = check_launch("fista_exact_kernel(ill-conditioned series)");
        if (rc != PB_OK) return rc;
      } else if (taps_dev && nd_g <= LDS_DOUBLES_MAX) {
        const int wgs = P < 2048 ? P : 2048;
        if (J_dev) hipLaunchKernelGGL((pb::fista_generic_kernel<true>), dim3(wgs), dim3(pb::GEN_THREADS), (size_t)nd_g * sizeof(double), user, b, taps_dev, K, wind);
        else hipLaunchKernelGGL((pb::fista_generic_kernel<false>), dim3(wgs), dim3(pb::GEN_THREADS), (size_t)nd_g * sizeof(double), user, b, taps_dev, K, wind);
        rc = check_launch("fista_generic_kernel(ill-conditioned series)");
        if (rc != PB_OK) return rc;
      }
    }
    // what the guards / certificates handed back (n_done = -1): compacted, then the exact vector forms at full occupancy
    pb::ClassPred cp{nullptr, 0.0, nullptr, 1, 0.0, n_done_dev};
    const pb::PlanSpec none{0, 0, 0, 0, 1, 0, 0, dense.slots};
    hipLaunchKernelGGL(pb::path_count_kernel, dim3(nblk), dim3(pb::PATH_THREADS), 0, user, cp, P, work_dev);
    hipLaunchKernelGGL(pb::path_scan_kernel, dim3(1), dim3(pb::PATH_THREADS), 0, user, P, nblk, work_dev, flagged, none, rg_flag, (int32_t*)nullptr, (int32_t*)nullptr);
    hipLaunchKernelGGL(pb::path_scatter_kernel, dim3(nblk), dim3(pb::PATH_THREADS), 0, user, cp, P, work_dev);
    rc = check_launch("partition(handed back)");
    if (rc != PB_OK) return rc;
    ss = nullptr;                                  // (one stream: a few per cent of the batch at most)
    return solve_list(rg_flag, 3, 3, true);
  };

  // series of 16 S < N <= 32 S scans (the reference's 600-scan demo): the pair form with the two
  // halves of ONE series in the slots of a row, in one launch; the window rule as a certificate,
  // re-solved on the one-problem-per-wave form
  const bool mfma_plain = stop_mode == PB_STOP_NONE && n_done_dev && (!lbda_dev || (flags & PB_FLAG_FORCE_MFMA)) &&
                          !(flags & (PB_FLAG_FORCE_GENERIC | PB_FLAG_NO_PAIR | PB_FLAG_FORCE_PAIR | PB_FLAG_FORCE_WIDE |
                                     PB_FLAG_DIRECT_FIR | PB_FLAG_NO_MFMA)) && mfma_serves_plain(N, K);
  // The matrix-pipe form with every series split over two waves (fista_mfma2.h): plain solves without cost trace.
  // Series of 311..640 scans run on it from MFMA2_LONG_MIN_P problems on: whole passes (and a remainder above a
  // quarter of a pass), the rest and whatever its guards hand back on the one-problem-per-wave form.  Shorter series
  // meet it as a piece of the plan below (small batches, remainders) or through PB_FLAG_FORCE_MFMA2.
  // (the window rule rides it as the no-fire certificate of the one-wave form: wind = 6, far from firing)
  const bool mfma2_cert = stop_mode == PB_STOP_WINDOW && wind == 6 && n_done_dev && !(flags & PB_FLAG_NO_CERT) &&
                          ((flags & PB_FLAG_FORCE_CERT) || tol * (double)n_iter < 0.02);
  // (the _loops_deconv rule in full inside the kernel, as on the one-wave form: no cost trace)
  const bool split_loops = stop_mode == PB_STOP_LOOPS && !J_dev;
  const mfma2_launch_fn mfma2 =
      ((stop_mode == PB_STOP_NONE || mfma2_cert || split_loops) && n_done_dev && (!lbda_dev || (flags & (PB_FLAG_FORCE_MFMA | PB_FLAG_FORCE_MFMA2))) &&
       !(flags & (PB_FLAG_FORCE_GENERIC | PB_FLAG_NO_PAIR | PB_FLAG_FORCE_PAIR | PB_FLAG_FORCE_WIDE | PB_FLAG_DIRECT_FIR |
                  PB_FLAG_NO_MFMA))) ? pick_mfma2(N, K, false) : nullptr;
  // (round 5) the same call shapes at 311..640 scans, partitioned on the device: dense class on whole passes of the split
  // form, sparse class on the pair form over two slots (or the backup form), handed-back problems compacted
  {
    const bool four = N > 640;                       // 641 .. 1 280 scans: the form split over four waves (fista_mfma4.h)
    const bool shape_ok = (stop_mode == PB_STOP_NONE || mfma2_cert || split_loops) && n_done_dev &&
                          !(flags & (PB_FLAG_FORCE_GENERIC | PB_FLAG_NO_PAIR | PB_FLAG_FORCE_PAIR | PB_FLAG_FORCE_WIDE | PB_FLAG_DIRECT_FIR | PB_FLAG_NO_MFMA));
    const mfma2_launch_fn mfma2_l = four ? ((shape_ok && mfma4_serves(N, K, false)) ? pick_mfma4(N, K, false) : nullptr)
                                         : ((mfma2 || !lbda_dev || !(stop_mode == PB_STOP_NONE || mfma2_cert || split_loops)) ? mfma2 : pick_mfma2(N, K, false));
    if (mfma2_l && part_ws && (four || ((mfma2_serves_long(N, K, false) || mfma2_takes_short_cert(N, K, stop_mode, wind)) && P >= mfma2_long_min_p(K)))) {
      const FastEntry* fe1 = pick_fast(N, K);
      const WideEntry* we1 = pick_wide(N, K);
      const bool use_wide = we1 && (!fe1 || N > 320);
      const bool backup_ok = (fe1 || we1) && (stop_mode != PB_STOP_WINDOW || (use_wide ? we1->S <= 20 : fe1->S <= 20));
      const FastEntry* se = pick_split(N, K);
      const bool scert_l = se && stop_mode == PB_STOP_WINDOW && wind == 6 && we1 && we1->S <= 20 && !(flags & PB_FLAG_NO_CERT) &&
                           ((flags & PB_FLAG_FORCE_CERT) || tol * (double)n_iter < 0.5);
      const bool split_ok = se != nullptr && (stop_mode == PB_STOP_NONE || scert_l);
      if (backup_ok) {
        const double slots = wave_slots();
        pb::PlanSpec dense{3, 0, 0, 0, 1, 1, 0, slots};
        if (four) { dense.pass_mult = 2; dense.rem_num16 = MFMA4_MIN_R_NUM; }
        pb::PlanSpec sparse{4, split_ok ? 1 : 0, 0, 0, 1, 0, 0, slots};
        pb::PlanSpec flagged{4, 0, 0, 0, 1, 0, 0, slots};
        dense.backup_form = sparse.backup_form = flagged.backup_form = use_wide ? FORM_WIDE : FORM_FAST1;
        sparse.min_pair = SPLIT_MIN_P;
        const bool wj = J_dev != nullptr;
        return run_partition(dense, sparse, flagged,
          [&](int form, const pb::FistaArgs& b, hipStream_t st, bool exact_rule) -> int {
            if (form == FORM_MFMA2) return mfma2_l(b, taps_host, K, wj, st);
            if (form == FORM_PAIR) return se->fn_pair_split(b, taps_host, K, wj, scert_l && !exact_rule, st);
            if (form == FORM_WIDE) return we1->fn(b, taps_host, K, wj, stop_mode, st);
            return fe1->fn(b, taps_host, K, wj, stop_mode, st);
          },
          [&](int form, int list) -> bool {
            if (form == FORM_MFMA2) return list <= 1;
            if (list == 1) return false;
            if (form == FORM_PAIR) return list != 3 && split_ok;
            if (form == FORM_WIDE) return use_wide;
            if (form == FORM_FAST1) return !use_wide;
            return false;
          },
          [&](int c) -> int { return (c == pb::CAND_MFMA2 || c == pb::CAND_PAIR0 || c == pb::CAND_FAST0 || c == pb::CAND_WIDE) ? P : 0; });
      }
    }
  }
  // 641 .. 1 280 scans: the same call shapes on the form split over four waves (fista_mfma4.h)
  const mfma2_launch_fn mfma4 =
      ((stop_mode == PB_STOP_NONE || mfma2_cert || split_loops) && n_done_dev && (!lbda_dev || (flags & (PB_FLAG_FORCE_MFMA | PB_FLAG_FORCE_MFMA2))) &&
       !(flags & (PB_FLAG_FORCE_GENERIC | PB_FLAG_NO_PAIR | PB_FLAG_FORCE_PAIR | PB_FLAG_FORCE_WIDE | PB_FLAG_DIRECT_FIR |
                  PB_FLAG_NO_MFMA)) && mfma4_serves(N, K, false)) ? pick_mfma4(N, K, false) : nullptr;
  if (mfma4) {
    const WideEntry* we1 = pick_wide(N, K);
    if (stop_mode != PB_STOP_WINDOW || we1->S <= 20) {     // (the window rule's re-solve needs the rule's increment ring)
      auto backup = [&](const pb::FistaArgs& b) -> int { return we1->fn(b, taps_host, K, J_dev != nullptr, stop_mode, (hipStream_t)stream); };
      const int base = (flags & PB_FLAG_FORCE_MFMA2) ? P : mfma4_base(P, (flags & PB_FLAG_ONE_LAUNCH) != 0);
      pb::FistaArgs b = a;
      if (base > 0) {
        b.P = base;
        if (mfma4(b, taps_host, K, J_dev != nullptr, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: four-wave matrix-pipe kernel rejected the launch");
        const int rc = check_launch("fista_mfma4_kernel");
        if (rc != PB_OK) return rc;
      }
      if (base < P) {
        b = a;
        b.p0 = base;
        if (backup(b) != 0) return fail(PB_ERR_INVALID, "pb_fista_solve: no vector form for the remainder");
        const int rc = check_launch("fista_fast_kernel(wide, remainder)");
        if (rc != PB_OK) return rc;
      }
      if (base > 0 && !(flags & PB_FLAG_CERT_NO_RESOLVE)) {
        b = a;
        b.P = base;
        b.only_flagged = 1;
        if (backup(b) != 0) return fail(PB_ERR_INVALID, "pb_fista_solve: no vector form for the re-solve");
        return check_launch("fista_fast_kernel(wide, re-solve)");
      }
      return PB_OK;
    }
  }
  if (mfma2 && ((flags & PB_FLAG_FORCE_MFMA2) || ((mfma2_serves_long(N, K, false) || mfma2_takes_short_cert(N, K, stop_mode, wind)) && P >= mfma2_long_min_p(K)))) {
    const FastEntry* fe1 = pick_fast(N, K);
    const WideEntry* we1 = pick_wide(N, K);
    const bool use_wide = we1 && (!fe1 || N > 320);
    // the exact vector form behind it (remainder, re-solve): single row, else one per wave -- with the window rule it
    // must hold the rule's increment ring (strips of at most 20 samples)
    const bool backup_ok = (fe1 || we1) && (stop_mode != PB_STOP_WINDOW || (use_wide ? we1->S <= 20 : fe1->S <= 20));
    if (backup_ok) {
      auto backup = [&](const pb::FistaArgs& b) -> int {
        return use_wide ? we1->fn(b, taps_host, K, J_dev != nullptr, stop_mode, (hipStream_t)stream)
                        : fe1->fn(b, taps_host, K, J_dev != nullptr, stop_mode, (hipStream_t)stream);
      };
      const int base = (flags & PB_FLAG_FORCE_MFMA2) ? P : mfma2_long_base(P, (flags & PB_FLAG_ONE_LAUNCH) != 0);
      pb::FistaArgs b = a;
      if (base > 0) {
        b.P = base;
        if (mfma2(b, taps_host, K, J_dev != nullptr, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: split matrix-pipe kernel rejected the launch");
        const int rc = check_launch("fista_mfma2_kernel");
        if (rc != PB_OK) return rc;
      }
      if (base < P) {
        b = a;
        b.p0 = base;
        if (backup(b) != 0) return fail(PB_ERR_INVALID, "pb_fista_solve: no vector form for the remainder");
        const int rc = check_launch("fista_fast_kernel(remainder)");
        if (rc != PB_OK) return rc;
      }
      if (base > 0 && !(flags & PB_FLAG_CERT_NO_RESOLVE)) {
        b = a;
        b.P = base;
        b.only_flagged = 1;
        if (backup(b) != 0) return fail(PB_ERR_INVALID, "pb_fista_solve: no vector form for the re-solve");
        return check_launch("fista_fast_kernel(re-solve)");
      }
      return PB_OK;
    }
  }
  if (!mfma_plain && !(flags & (PB_FLAG_FORCE_GENERIC | PB_FLAG_NO_PAIR | PB_FLAG_FORCE_WIDE | PB_FLAG_DIRECT_FIR))) {
    const FastEntry* se = pick_split(N, K);
    const WideEntry* wre = se ? pick_wide(N, K) : nullptr;
    const bool scert = se && stop_mode == PB_STOP_WINDOW && wind == 6 && n_done_dev && wre && wre->S <= 20 &&
                       !(flags & PB_FLAG_NO_CERT) && ((flags & PB_FLAG_FORCE_CERT) || tol * (double)n_iter < 0.5);
    if (se && (P >= SPLIT_MIN_P || (flags & PB_FLAG_FORCE_PAIR)) && (stop_mode == PB_STOP_NONE || scert)) {
      if (se->fn_pair_split(a, taps_host, K, J_dev != nullptr, scert, (hipStream_t)stream) != 0)
        return fail(PB_ERR_INVALID, "pb_fista_solve: split pair kernel rejected the launch");
      int rc = check_launch("fista_pair_ffa_kernel(split)");
      if (rc == PB_OK && scert && !(flags & PB_FLAG_CERT_NO_RESOLVE)) {
        pb::FistaArgs b = a;
        b.only_flagged = 1;
        if (wre->fn(b, taps_host, K, J_dev != nullptr, PB_STOP_WINDOW, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: no one-problem-per-wave form for the re-solve");
        rc = check_launch("fista_fast_kernel(wide, re-solve)");
      }
      return rc;
    }
  }
  const FastEntry* fe = (flags & PB_FLAG_FORCE_GENERIC) ? nullptr : pick_fast(N, K);
  // the register-resident window rule keeps wind-1 = 5 iterates in VGPRs: wind = 6
  // (the reference default) on entries small enough to hold them; else LDS kernel
  if (fe && stop_mode == PB_STOP_WINDOW && (!ring_wind(wind) || fe->S > 20)) fe = nullptr;
  if (fe && (flags & PB_FLAG_FORCE_WIDE)) fe = nullptr;
  // Window rule on the pair form: a per-iteration no-fire certificate (fista_pair_ffa.h), then an
  // exact re-solve of the problems it could not clear (n_done = -1) on the single-row form.  Worth
  // it when the rule is not expected to fire: the criterion decays like ~0.9/k on this problem
  // class, so it cannot pass below tol before k ~ 0.9/tol.
  const bool cert = fe && stop_mode == PB_STOP_WINDOW && wind == 6 && fe->fn_pair_cert && n_done_dev &&
                    P >= 2 && !(flags & (PB_FLAG_NO_PAIR | PB_FLAG_NO_CERT | PB_FLAG_DIRECT_FIR)) &&
                    ((flags & PB_FLAG_FORCE_CERT) || tol * (double)n_iter < 0.5);
  // plain solves (cost trace or not) of 129..310 scans, HRFs up to 33 taps (34..65: no certificate): both operators on the
  // matrix pipe (fista_mfma.h).  Needs n_done_dev: a problem whose scaled operands left the float16
  // range comes back with n_done = -1 and is re-solved on the single-row form.
  // Not with one lambda per problem, unless asked for (PB_FLAG_FORCE_MFMA): along a regularisation
  // path a third of the problems (lambda near lambda_max) fail that kernel's accuracy guard and
  // would be solved twice.
  // The window rule rides it as the same no-fire certificate as on the pair form (`cert` above).
  // (its bound rests on four tracked samples per problem instead of sixteen: only where the rule is
  // far from firing, tol * n_iter < 0.02; closer calls stay on the pair form)
  const bool mfma_cert = cert && ((flags & PB_FLAG_FORCE_MFMA) || tol * (double)n_iter < 0.02);
  // (the _loops_deconv rule rides it too, evaluated exactly inside the kernel: no cost trace, K <= 33)
  const bool mfma_loops = stop_mode == PB_STOP_LOOPS && !J_dev && K <= MFMA_K2;
  const mfma_launch_fn mfma = (fe && (stop_mode == PB_STOP_NONE || mfma_cert || mfma_loops) && n_done_dev &&
                               (!lbda_dev || (flags & PB_FLAG_FORCE_MFMA)) &&
                               !(flags & (PB_FLAG_NO_PAIR | PB_FLAG_FORCE_PAIR | PB_FLAG_DIRECT_FIR | PB_FLAG_NO_MFMA)))
                                  ? pick_mfma(N, K, stop_mode != PB_STOP_NONE) : nullptr;
  if (fe) {
    auto run = [&](int form, int p0, int p1) -> int {
      pb::FistaArgs b = a;
      b.p0 = p0;
      b.P = p1;
      if (form == FORM_MFMA) {
        if (!mfma || mfma(b, taps_host, K, J_dev != nullptr, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: matrix-pipe kernel rejected the launch");
        return check_launch("fista_mfma_kernel");
      }
      if (form == FORM_MFMA2) {
        if (!mfma2 || mfma2(b, taps_host, K, J_dev != nullptr, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: split matrix-pipe kernel rejected the launch");
        return check_launch("fista_mfma2_kernel");
      }
      if (form == FORM_PAIR && cert) {
        if (fe->fn_pair_cert(b, taps_host, K, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: certificate kernel rejected the launch");
        return check_launch("fista_pair_ffa_kernel(cert)");
      }
      if (form == FORM_PAIR) {
        const pair_launch_fn fn = (fe->fn_pair_ffa && !(flags & PB_FLAG_DIRECT_FIR)) ? fe->fn_pair_ffa
                                                                                      : fe->fn_pair;
        if (fn(b, taps_host, K, J_dev != nullptr, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: pair kernel rejected the launch");
        return check_launch("fista_pair_kernel");
      }
      if (form == FORM_WIDE) {
        const WideEntry* we = pick_wide_small(N, K);
        if (!we || we->fn(b, taps_host, K, J_dev != nullptr, stop_mode, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: no one-problem-per-wave form");
        return check_launch("fista_fast_kernel(wide)");
      }
      if (fe->fn(b, taps_host, K, J_dev != nullptr, stop_mode, (hipStream_t)stream) != 0)
        return fail(PB_ERR_INVALID, "pb_fista_solve: no register-resident form for this stop rule");
      return check_launch("fista_fast_kernel");
    };
    // flagged problems of the pair pieces [q0, q1): exact window rule, on `stream` (after the join)
    auto resolve = [&](int q0, int q1) -> int {
      if (flags & PB_FLAG_CERT_NO_RESOLVE) return PB_OK;
      pb::FistaArgs b = a;
      b.p0 = q0;
      b.P = q1;
      b.only_flagged = 1;
      if (fe->fn(b, taps_host, K, J_dev != nullptr, stop_mode, (hipStream_t)stream) != 0)
        return fail(PB_ERR_INVALID, "pb_fista_solve: no single-row form for the re-solve");
      return check_launch("fista_fast_kernel(re-solve)");
    };
    if (flags & PB_FLAG_NO_PAIR) return run(FORM_FAST1, 0, P);
    // (round 5: the call partitioned on the device -- run_partition above)
    const bool part_mfma = mfma != nullptr || (fe && (stop_mode == PB_STOP_NONE || mfma_cert || mfma_loops) && lbda_dev &&
                                               pick_mfma(N, K, stop_mode != PB_STOP_NONE) != nullptr);
    // (... and a call no matrix-pipe form carries -- another window, a cost trace beside the _loops_deconv rule, a long HRF --
    // is partitioned all the same, with an empty dense class: the conditioning guard is the partition's, and float32 vector
    // forms need it too (ill-conditioned series: 3e-5 .. 5e-3 without it))
    const bool part_guard_only = !part_mfma && !(flags & PB_FLAG_NO_ILL_GUARD);
    if ((part_mfma || part_guard_only) && part_ws) {
      no_dense_class = !part_mfma;
      const mfma_launch_fn mfma_p = !part_mfma ? nullptr : (mfma ? mfma : pick_mfma(N, K, stop_mode != PB_STOP_NONE));
      const mfma2_launch_fn mfma2_p = !part_mfma ? nullptr : ((mfma2 || !lbda_dev) ? mfma2 : (((stop_mode == PB_STOP_NONE || mfma2_cert) && K <= MFMA_K2) ? pick_mfma2(N, K) : nullptr));
      const bool one_stream = (flags & PB_FLAG_ONE_STREAM) != 0 || stream_is_capturing((hipStream_t)stream);
      const double slots = wave_slots();
      const bool has_wide = pick_wide_small(N, K) != nullptr;
      const bool has_pair = (fe->fn_pair != nullptr && stop_mode == PB_STOP_NONE) || cert;
      const bool has_mfma2 = mfma2_p != nullptr && (stop_mode == PB_STOP_NONE || mfma_cert);
      const pb::PlanSpec dense{1, has_pair ? 1 : 0, has_wide ? 1 : 0, 0, one_stream ? 1 : 0, has_mfma2 ? 1 : 0, beside_chunks_for(N), slots};
      const pb::PlanSpec sparse{2, has_pair ? 1 : 0, has_wide ? 1 : 0, 0, one_stream ? 1 : 0, 0, 0, slots};
      const pb::PlanSpec flagged{2, 0, has_wide ? 1 : 0, 0, 1, 0, 0, slots};
      const bool wj = J_dev != nullptr;
      return run_partition(dense, sparse, flagged,
        [&](int form, const pb::FistaArgs& b, hipStream_t st, bool exact_rule) -> int {
          if (form == FORM_MFMA) return mfma_p(b, taps_host, K, wj, st);
          if (form == FORM_MFMA2) return mfma2_p(b, taps_host, K, wj, st);
          if (form == FORM_PAIR && cert && !exact_rule) return fe->fn_pair_cert(b, taps_host, K, st);
          if (form == FORM_PAIR) return fe->fn_pair_ffa(b, taps_host, K, wj, st);
          if (form == FORM_WIDE) return pick_wide_small(N, K)->fn(b, taps_host, K, wj, stop_mode, st);
          return fe->fn(b, taps_host, K, wj, stop_mode, st);
        },
        [&](int form, int list) -> bool {
          if (form == FORM_MFMA) return list <= 1 && mfma_p != nullptr;
          if (form == FORM_MFMA2) return list <= 1 && has_mfma2;
          if (list == 1) return false;                                // (aid: matrix-pipe candidates only)
          if (form == FORM_PAIR) return list != 3 && has_pair;
          if (form == FORM_WIDE) return has_wide;
          return true;
        },
        [&](int c) -> int { return pb::cand_max_slots(c, P, slots); });
    }
    if (flags & PB_FLAG_FORCE_PAIR) {
      if (cert) {
        const int rc = run(FORM_PAIR, 0, P);
        return rc != PB_OK ? rc : resolve(0, P);
      }
      if (!fe->fn_pair || P < 2 || stop_mode != PB_STOP_NONE) return run(FORM_FAST1, 0, P);
      return run(FORM_PAIR, 0, P);
    }
    // whole rounds on the densest form, the remainder on the cheapest (the pair form has no
    // stop rules; the one-problem-per-wave form has them all); a remainder that fits beside
    // half a round of pair waves runs on the side stream
    const bool pair_ok = (fe->fn_pair != nullptr && stop_mode == PB_STOP_NONE) || cert;
    Piece pc[6];
    const bool one_stream = (flags & PB_FLAG_ONE_STREAM) != 0 || stream_is_capturing((hipStream_t)stream);
    const int npc = mfma ? plan_pieces_mfma(P, pair_ok, pick_wide_small(N, K) != nullptr,
                                            (flags & PB_FLAG_ONE_LAUNCH) != 0, one_stream,
                                            mfma2 != nullptr && (stop_mode == PB_STOP_NONE || mfma_cert), beside_chunks_for(N), pc)
                         : plan_pieces(P, pair_ok, pick_wide_small(N, K) != nullptr,
                                       (flags & PB_FLAG_ONE_LAUNCH) != 0, one_stream, pc);
    bool any_side = false;
    int q0 = P, q1 = 0;                          // range of the pieces that may leave n_done = -1 (contiguous)
    for (int i = 0; i < npc; ++i) {
      any_side |= pc[i].side;
      if ((pc[i].form == FORM_PAIR && cert) || pc[i].form == FORM_MFMA || pc[i].form == FORM_MFMA2) {
        q0 = pc[i].p0 < q0 ? pc[i].p0 : q0;
        q1 = pc[i].p1 > q1 ? pc[i].p1 : q1;
      }
    }
    auto finish = [&](int rc) -> int { return (rc == PB_OK && q1 > q0) ? resolve(q0, q1) : rc; };
    if (!any_side) {
      for (int i = 0; i < npc; ++i) {
        const int rc = run(pc[i].form, pc[i].p0, pc[i].p1);
        if (rc != PB_OK) return rc;
      }
      return finish(PB_OK);
    }
    std::lock_guard<std::mutex> lock(g_side_mutex);
    SideStream* ss = side_stream_locked();
    if (!ss) {                                   // no side stream: same pieces, one after the other
      for (int i = 0; i < npc; ++i) {
        const int rc = run(pc[i].form, pc[i].p0, pc[i].p1);
        if (rc != PB_OK) return rc;
      }
      return finish(PB_OK);
    }
    hipStream_t user = (hipStream_t)stream;
    int rc_all = PB_OK;
    bool forked = false;
    for (int i = 0; i < npc && rc_all == PB_OK; ++i) {
      if (pc[i].group && !forked) {              // whole rounds are in the queue: the group starts here
        if (hipEventRecord(ss->fork, user) != hipSuccess || hipStreamWaitEvent(ss->stream, ss->fork, 0) != hipSuccess)
          return fail(PB_ERR_HIP, "pb_fista_solve: fork to the side stream failed");
        forked = true;
      }
      stream = pc[i].side ? (void*)ss->stream : (void*)user;      // `run` launches on `stream`
      rc_all = run(pc[i].form, pc[i].p0, pc[i].p1);
    }
    stream = (void*)user;
    if (!forked) return finish(rc_all);
    // join even after an error so that the caller's stream never runs ahead of the side stream
    if (hipEventRecord(ss->join, ss->stream) != hipSuccess || hipStreamWaitEvent(user, ss->join, 0) != hipSuccess)
      return fail(PB_ERR_HIP, "pb_fista_solve: join of the side stream failed");
    return finish(rc_all);
  }
  // 305..310 scans with more than 32 taps: no single-row entry, but ten blocks of 31 samples fit the matrix-pipe
  // form -- whole rounds (or everything) on it, a small remainder and the problems its guards hand back on the
  // one-problem-per-wave form
  if (!fe && mfma_plain) {
    const WideEntry* we = pick_wide(N, K);
    const mfma_launch_fn mf = pick_mfma(N, K);
    if (we && mf) {
      const int base = mfma_wide_base(P, (flags & PB_FLAG_ONE_LAUNCH) != 0);
      pb::FistaArgs b = a;
      if (base > 0) {
        b.P = base;
        if (mf(b, taps_host, K, J_dev != nullptr, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: matrix-pipe kernel rejected the launch");
        const int rc = check_launch("fista_mfma_kernel");
        if (rc != PB_OK) return rc;
      }
      if (base < P) {
        b = a;
        b.p0 = base;
        if (we->fn(b, taps_host, K, J_dev != nullptr, stop_mode, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: no one-problem-per-wave form");
        const int rc = check_launch("fista_fast_kernel(wide)");
        if (rc != PB_OK) return rc;
      }
      if (base > 0 && !(flags & PB_FLAG_CERT_NO_RESOLVE)) {
        b = a;
        b.P = base;
        b.only_flagged = 1;
        if (we->fn(b, taps_host, K, J_dev != nullptr, stop_mode, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve: no one-problem-per-wave form for the re-solve");
        return check_launch("fista_fast_kernel(wide, re-solve)");
      }
      return PB_OK;
    }
  }
  // long series: one problem per wave (window rule: wind = 6 and S <= 20, as above)
  if (!(flags & PB_FLAG_FORCE_GENERIC)) {
    const WideEntry* we = pick_wide(N, K);
    if (we && stop_mode == PB_STOP_WINDOW && (!ring_wind(wind) || we->S > 20)) we = nullptr;
    if (we) {
      if (we->fn(a, taps_host, K, J_dev != nullptr, stop_mode, (hipStream_t)stream) != 0)
        return fail(PB_ERR_INVALID, "pb_fista_solve: no one-problem-per-wave form for this stop rule");
      return check_launch("fista_fast_kernel(wide)");
    }
  }
  if (flags & PB_FLAG_FORCE_FAST)
    return fail(PB_ERR_INVALID, "pb_fista_solve: no register-resident kernel for N=%d K=%d stop=%d",
                N, K, stop_mode);

  // generic path (any N, K that fit LDS; all stop rules)
  if (!taps_dev) return fail(PB_ERR_INVALID, "pb_fista_solve: taps_dev required for the generic kernel");
  const int64_t nd = 3 * (int64_t)N + K + 2 * pb::GEN_WAVES +
                     (stop_mode == PB_STOP_WINDOW ? (int64_t)wind * N : 0);
  if (nd > LDS_DOUBLES_MAX)
    return fail(PB_ERR_INVALID, "pb_fista_solve: N=%d K=%d wind=%d exceeds LDS", N, K, wind);
  const size_t lds = (size_t)nd * sizeof(double);
  if (J_dev)
    hipLaunchKernelGGL((pb::fista_generic_kernel<true>), dim3(P), dim3(pb::GEN_THREADS), lds,
                       (hipStream_t)stream, a, taps_dev, K, wind);
  else
    hipLaunchKernelGGL((pb::fista_generic_kernel<false>), dim3(P), dim3(pb::GEN_THREADS), lds,
                       (hipStream_t)stream, a, taps_dev, K, wind);
  return check_launch("fista_generic_kernel");
}

int pb_fista_list_plan(int kind, int n, int n_max, int has_pair, int has_wide, int one_stream, int has_mfma2,
                       int beside_chunks, int32_t* ranges, int32_t* bounds) {
  if (kind < 1 || kind > 3 || n < 0 || n_max < n) return fail(PB_ERR_INVALID, "pb_fista_list_plan: bad argument");
  Piece pc[pb::MAX_PIECES];
  int npc = 0;
  const double slots = wave_slots();
  if (n > 0 && kind == 1) npc = pb::plan_pieces_mfma(n, has_pair != 0, has_wide != 0, false, one_stream != 0, has_mfma2 != 0, beside_chunks, slots, pc);
  else if (n > 0 && kind == 2) npc = pb::plan_pieces(n, has_pair != 0, has_wide != 0, false, one_stream != 0, slots, pc);
  else if (kind == 3 && n_max > 0)      // a partitioned call of n_max problems, n of them dense
    npc = pb::plan_partitioned(n, n_max, has_pair != 0, has_wide != 0, one_stream != 0, has_mfma2 != 0, beside_chunks, slots, pc);
  int32_t rg[2 * pb::CAND_COUNT];
  const int rc = pb::plan_to_candidates(pc, npc, rg);
  for (int c = 0; c < pb::CAND_COUNT; ++c) {
    if (ranges) { ranges[2 * c] = rg[2 * c]; ranges[2 * c + 1] = rg[2 * c + 1]; }
    if (bounds) bounds[c] = pb::cand_max_slots(c, n_max, slots);
  }
  if (rc != 0) return fail(PB_ERR_INVALID, "pb_fista_list_plan: a piece of the plan found its candidate launch taken (n=%d)", n);
  g_err[0] = 0;
  return PB_OK;
}

int64_t pb_fista_work_len(int P, int y_rep) {
  if (P < 0 || y_rep < 1) return 0;
  return work_layout(P, (P + y_rep - 1) / y_rep).total;
}

int pb_fista_solve_ex(const float* y_dev, int64_t ldy, int y_rep, double* w_dev, int64_t ldw, int P,
                      int N, const double* taps_host, const double* taps_dev, int K, double step,
                      double lbda, const double* lbda_dev, const double* betas_dev, int n_iter, float* J_dev,
                      int64_t ldj, int stop_mode, double tol, int wind, int32_t* n_done_dev,
                      unsigned flags, void* stream, const double* lmax_dev, double dense_ratio,
                      int32_t* work_dev, int64_t work_len) {
  return solve_impl(y_dev, ldy, y_rep, w_dev, ldw, P, N, taps_host, taps_dev, K, step, lbda, lbda_dev, betas_dev, n_iter,
                    J_dev, ldj, stop_mode, tol, wind, n_done_dev, flags, stream, lmax_dev, dense_ratio, work_dev, work_len);
}

int pb_fista_solve(const float* y_dev, int64_t ldy, int y_rep, double* w_dev, int64_t ldw, int P,
                   int N, const double* taps_host, const double* taps_dev, int K, double step,
                   double lbda,
                   const double* lbda_dev, const double* betas_dev, int n_iter, float* J_dev,
                   int64_t ldj, int stop_mode, double tol, int wind, int32_t* n_done_dev,
                   unsigned flags, void* stream) {
  // a workspace of the library's own for the partition (grown on demand, one per device and stream; never while the
  // stream is being captured: an allocation cannot be captured -- such calls, and calls that fail to get memory, run
  // without the partition.  pb_fista_solve_ex takes the caller's workspace instead and allocates nothing.)
  int32_t* work = nullptr;
  int64_t len = 0;
  if (P >= PART_MIN_P && N <= 1280 && n_done_dev && !(flags & PB_FLAG_NO_PARTITION) && !stream_is_capturing((hipStream_t)stream))
    work = own_workspace(stream, work_layout(P, (P + (y_rep > 0 ? y_rep : 1) - 1) / (y_rep > 0 ? y_rep : 1)).total, &len);
  return solve_impl(y_dev, ldy, y_rep, w_dev, ldw, P, N, taps_host, taps_dev, K, step, lbda, lbda_dev, betas_dev, n_iter,
                    J_dev, ldj, stop_mode, tol, wind, n_done_dev, flags, stream, nullptr, 0.0, work, len);
}

int pb_fista_solve_d(const double* y_dev, int64_t ldy, int y_rep, double* w_dev, int64_t ldw,
                     int P, int N, const double* taps_host, const double* taps_dev, int K,
                     double step, double lbda, const double* lbda_dev, const double* betas_dev,
                     int n_iter, double* J_dev, int64_t ldj, int stop_mode, double tol, int wind,
                     int32_t* n_done_dev, unsigned flags, void* stream) {
  if (P < 0 || N < 1 || K < 1 || n_iter < 0 || y_rep < 1)
    return fail(PB_ERR_INVALID, "pb_fista_solve_d: bad size (P=%d N=%d K=%d n_iter=%d y_rep=%d)", P,
                N, K, n_iter, y_rep);
  if (ldy < N || ldw < N) return fail(PB_ERR_INVALID, "pb_fista_solve_d: leading dimension < N");
  if (J_dev && ldj < n_iter) return fail(PB_ERR_INVALID, "pb_fista_solve_d: ldj < n_iter");
  if (!(step > 0.0)) return fail(PB_ERR_INVALID, "pb_fista_solve_d: step must be positive");
  if (stop_mode < PB_STOP_NONE || stop_mode > PB_STOP_WINDOW)
    return fail(PB_ERR_INVALID, "pb_fista_solve_d: unknown stop_mode %d", stop_mode);
  if (stop_mode == PB_STOP_WINDOW && wind < 2)
    return fail(PB_ERR_INVALID, "pb_fista_solve_d: wind must be >= 2");
  // register-resident float64 form (one problem per wave) when the shape has an entry and
  // the host copy of the taps is given; the window rule there is the reference default wind = 6
  const ExactEntry* ee = (taps_host && !(flags & PB_FLAG_FORCE_GENERIC)) ? pick_exact(N, K) : nullptr;
  if (ee && stop_mode == PB_STOP_WINDOW && wind != 6) ee = nullptr;
  if (!ee && (flags & PB_FLAG_FORCE_FAST))
    return fail(PB_ERR_INVALID, "pb_fista_solve_d: no register-resident float64 kernel for N=%d K=%d", N, K);
  const int64_t nd = 3 * (int64_t)N + K + 2 * pb::GEN_WAVES +
                     (stop_mode == PB_STOP_WINDOW ? (int64_t)wind * N : 0);
  if (!ee && nd > LDS_DOUBLES_MAX)
    return fail(PB_ERR_INVALID, "pb_fista_solve_d: N=%d K=%d wind=%d exceeds LDS", N, K, wind);
  if (P == 0) return PB_OK;
  if (!y_dev || !w_dev || (!ee && !taps_dev) || (n_iter > 0 && !betas_dev))
    return fail(PB_ERR_INVALID, "pb_fista_solve_d: NULL pointer");
  if (P > (1 << 25)) return fail(PB_ERR_INVALID, "pb_fista_solve_d: more than 2^25 problems per launch");
  pb::FistaArgs a;
  a.y = nullptr; a.y64 = y_dev; a.ldy = ldy; a.w = w_dev; a.ldw = ldw; a.lbda_vec = lbda_dev;
  a.betas = betas_dev; a.J = nullptr; a.J64 = J_dev; a.ldj = ldj; a.n_done = n_done_dev;
  a.step = step; a.lbda = lbda; a.tol = tol;
  a.y_rep = y_rep; a.P = P; a.N = N; a.n_iter = n_iter; a.stop_mode = stop_mode;
  a.taps_pp = nullptr; a.ldt = 0; a.step_vec = nullptr; a.step_shared = 0; a.K = K; a.p0 = 0;
  a.cold = (flags & PB_FLAG_COLD_START) ? 1 : 0;
  a.rho_guard = (flags & PB_FLAG_NO_RHO_GUARD) ? 0 : 1;
  if (ee) {
    if (ee->fn(a, taps_host, K, J_dev != nullptr, stop_mode, (hipStream_t)stream) != 0)
      return fail(PB_ERR_INVALID, "pb_fista_solve_d: launch rejected");
    return check_launch("fista_exact_kernel");
  }
  const size_t lds = (size_t)nd * sizeof(double);
  if (J_dev)
    hipLaunchKernelGGL((pb::fista_generic_kernel<true, true>), dim3(P), dim3(pb::GEN_THREADS), lds,
                       (hipStream_t)stream, a, taps_dev, K, wind);
  else
    hipLaunchKernelGGL((pb::fista_generic_kernel<false, true>), dim3(P), dim3(pb::GEN_THREADS), lds,
                       (hipStream_t)stream, a, taps_dev, K, wind);
  return check_launch("fista_generic_kernel(f64)");
}

int pb_fista_solve_backtrack_d(const double* y_dev, int64_t ldy, int y_rep, double* w_dev, int64_t ldw, int P, int N,
                               const double* taps_dev, int K, double step0, double eta, int max_halvings_per_iter,
                               double lbda, const double* lbda_dev, const double* betas_dev, int n_iter,
                               int32_t* n_done_dev, double* step_out_dev, int32_t* halvings_out_dev, unsigned flags, void* stream) {
  if (P < 0 || N < 1 || K < 1 || n_iter < 0 || y_rep < 1)
    return fail(PB_ERR_INVALID, "pb_fista_solve_backtrack_d: bad size (P=%d N=%d K=%d n_iter=%d y_rep=%d)", P, N, K, n_iter, y_rep);
  if (ldy < N || ldw < N) return fail(PB_ERR_INVALID, "pb_fista_solve_backtrack_d: leading dimension < N");
  if (!(step0 > 0.0) || !(eta > 0.0 && eta < 1.0) || max_halvings_per_iter < 0)
    return fail(PB_ERR_INVALID, "pb_fista_solve_backtrack_d: step0 > 0, 0 < eta < 1 and max_halvings_per_iter >= 0 are required");
  const int64_t nd = 5 * (int64_t)N + K + 2 * pb::GEN_WAVES;
  if (nd > LDS_DOUBLES_MAX) return fail(PB_ERR_INVALID, "pb_fista_solve_backtrack_d: N=%d K=%d exceeds LDS", N, K);
  if (P == 0) return PB_OK;
  if (!y_dev || !w_dev || !taps_dev || (n_iter > 0 && !betas_dev)) return fail(PB_ERR_INVALID, "pb_fista_solve_backtrack_d: NULL pointer");
  pb::FistaArgs a;
  a.y = nullptr; a.y64 = y_dev; a.ldy = ldy; a.w = w_dev; a.ldw = ldw; a.lbda_vec = lbda_dev;
  a.betas = betas_dev; a.J = nullptr; a.J64 = nullptr; a.ldj = 0; a.n_done = n_done_dev;
  a.step = step0; a.lbda = lbda; a.tol = 0.0;
  a.y_rep = y_rep; a.P = P; a.N = N; a.n_iter = n_iter; a.stop_mode = PB_STOP_NONE;
  a.taps_pp = nullptr; a.ldt = 0; a.step_vec = nullptr; a.step_shared = 0; a.K = K; a.p0 = 0;
  a.cold = (flags & PB_FLAG_COLD_START) ? 1 : 0;
  hipLaunchKernelGGL(pb::fista_backtrack_kernel, dim3(P), dim3(pb::GEN_THREADS), (size_t)nd * sizeof(double), (hipStream_t)stream,
                     a, taps_dev, K, eta, max_halvings_per_iter, step_out_dev, halvings_out_dev);
  return check_launch("fista_backtrack_kernel");
}

int pb_fista_outputs(const double* w_dev, int64_t ldw, int P, int N, const double* taps_dev, int K,
                     double* z_dev, int64_t ldz, double* x_dev, int64_t ldx, void* stream) {
  if (P < 0 || N < 1 || K < 1 || ldw < N || (z_dev && ldz < N) || (x_dev && ldx < N))
    return fail(PB_ERR_INVALID, "pb_fista_outputs: bad size");
  if (2 * (int64_t)N + K + 8 > LDS_DOUBLES_MAX)
    return fail(PB_ERR_INVALID, "pb_fista_outputs: N=%d K=%d exceeds LDS", N, K);
  if (P == 0 || (!z_dev && !x_dev)) return PB_OK;
  if (!w_dev || !taps_dev) return fail(PB_ERR_INVALID, "pb_fista_outputs: NULL pointer");
  const size_t lds = (size_t)(2 * N + K + 8) * sizeof(double);
  hipLaunchKernelGGL(pb::outputs_kernel, dim3(P), dim3(pb::GEN_THREADS), lds, (hipStream_t)stream,
                     w_dev, ldw, N, taps_dev, (int64_t)0, K, z_dev, ldz, x_dev, ldx);
  return check_launch("outputs_kernel");
}

int pb_fista_outputs_pp(const double* w_dev, int64_t ldw, int P, int N, const double* taps_dev,
                        int64_t ldt, int K, double* z_dev, int64_t ldz, double* x_dev, int64_t ldx,
                        void* stream) {
  if (P < 0 || N < 1 || K < 1 || ldw < N || ldt < K || (z_dev && ldz < N) || (x_dev && ldx < N))
    return fail(PB_ERR_INVALID, "pb_fista_outputs_pp: bad size");
  if (2 * (int64_t)N + K + 8 > LDS_DOUBLES_MAX)
    return fail(PB_ERR_INVALID, "pb_fista_outputs_pp: N=%d K=%d exceeds LDS", N, K);
  if (P == 0 || (!z_dev && !x_dev)) return PB_OK;
  if (!w_dev || !taps_dev) return fail(PB_ERR_INVALID, "pb_fista_outputs_pp: NULL pointer");
  const size_t lds = (size_t)(2 * N + K + 8) * sizeof(double);
  hipLaunchKernelGGL(pb::outputs_kernel, dim3(P), dim3(pb::GEN_THREADS), lds, (hipStream_t)stream,
                     w_dev, ldw, N, taps_dev, ldt, K, z_dev, ldz, x_dev, ldx);
  return check_launch("outputs_kernel(pp)");
}

int pb_spm_hrf(const double* deltas_dev, int M, const double* t_dev, int K, double a_peak,
               double loc_peak, double a_under, double loc_under, double ratio, double* out_dev,
               void* stream) {
  if (M < 0 || K < 1 || !(a_peak > 0.0) || !(a_under > 0.0))
    return fail(PB_ERR_INVALID, "pb_spm_hrf: bad argument");
  if (M == 0) return PB_OK;
  if (!deltas_dev || !t_dev || !out_dev) return fail(PB_ERR_INVALID, "pb_spm_hrf: NULL pointer");
  const int64_t total = (int64_t)M * K;
  const unsigned blocks = (unsigned)((total + pb::GEN_THREADS - 1) / pb::GEN_THREADS);
  hipLaunchKernelGGL(pb::spm_hrf_kernel, dim3(blocks), dim3(pb::GEN_THREADS), 0, (hipStream_t)stream,
                     deltas_dev, M, t_dev, K, a_peak, loc_peak, lgamma(a_peak), a_under, loc_under,
                     lgamma(a_under), ratio, out_dev);
  return check_launch("spm_hrf_kernel");
}


int pb_fista_stats(const double* w_dev, int64_t ldw, const float* y_dev, int64_t ldy, int y_rep,
                   int P, int N, const double* taps_dev, int K, double* r2_dev, double* l1_dev,
                   void* stream) {
  return stats_impl<float>(w_dev, ldw, y_dev, ldy, y_rep, P, N, taps_dev, K, r2_dev, l1_dev, stream,
                           "pb_fista_stats");
}
int pb_fista_stats_d(const double* w_dev, int64_t ldw, const double* y_dev, int64_t ldy, int y_rep,
                     int P, int N, const double* taps_dev, int K, double* r2_dev, double* l1_dev,
                     void* stream) {
  return stats_impl<double>(w_dev, ldw, y_dev, ldy, y_rep, P, N, taps_dev, K, r2_dev, l1_dev, stream,
                            "pb_fista_stats_d");
}

int pb_spectral_radius(const double* x0_dev, int N, const double* taps_dev, int K, int nb_iter,
                       double tol, double* out_dev, void* stream) {
  if (N < 1 || K < 1 || nb_iter < 0) return fail(PB_ERR_INVALID, "pb_spectral_radius: bad size");
  if (!x0_dev || !taps_dev || !out_dev) return fail(PB_ERR_INVALID, "pb_spectral_radius: NULL pointer");
  if (3 * (int64_t)N + K + 8 > LDS_DOUBLES_MAX)
    return fail(PB_ERR_INVALID, "pb_spectral_radius: N=%d K=%d exceeds LDS", N, K);
  const size_t lds = (size_t)(3 * N + K + 8) * sizeof(double);
  hipLaunchKernelGGL(pb::power_iter_kernel, dim3(1), dim3(pb::GEN_THREADS), lds, (hipStream_t)stream,
                     x0_dev, N, taps_dev, K, nb_iter, tol, out_dev);
  return check_launch("power_iter_kernel");
}

int pb_integ_op(const double* x, int64_t ldx, double* out, int64_t ldo, int V, int N, void* st) {
  return launch_op<pb::OP_INTEG>(x, ldx, out, ldo, V, N, N, nullptr, 0, st, "pb_integ_op");
}
int pb_integ_adj(const double* x, int64_t ldx, double* out, int64_t ldo, int V, int N, void* st) {
  return launch_op<pb::OP_INTEG_ADJ>(x, ldx, out, ldo, V, N, N, nullptr, 0, st, "pb_integ_adj");
}
int pb_conv(const double* x, int64_t ldx, double* out, int64_t ldo, int V, int n_in, int n_out,
            const double* taps, int K, void* st) {
  return launch_op<pb::OP_CONV>(x, ldx, out, ldo, V, n_in, n_out, taps, K, st, "pb_conv");
}
int pb_corr(const double* r, int64_t ldr, double* out, int64_t ldo, int V, int n_in, int n_out,
            const double* taps, int K, void* st) {
  // r has n_out samples (the range of the Toeplitz matrix), the result n_in
  return launch_op<pb::OP_CORR>(r, ldr, out, ldo, V, n_out, n_in, taps, K, st, "pb_corr");
}
int pb_op_forward(const double* x, int64_t ldx, double* out, int64_t ldo, int V, int n_in, int n_out,
                  const double* taps, int K, void* st) {
  return launch_op<pb::OP_FWD>(x, ldx, out, ldo, V, n_in, n_out, taps, K, st, "pb_op_forward");
}
int pb_op_adjoint(const double* r, int64_t ldr, double* out, int64_t ldo, int V, int n_in, int n_out,
                  const double* taps, int K, void* st) {
  return launch_op<pb::OP_ADJ>(r, ldr, out, ldo, V, n_out, n_in, taps, K, st, "pb_op_adjoint");
}

int pb_hrf_cost(const double* z_dev, int64_t ldz, const float* y_dev, int64_t ldy, int V, int N,
                const double* taps_dev, int K, int n_hrf, double* cost_dev, void* stream) {
  return hrf_cost_impl<float>(z_dev, ldz, y_dev, ldy, V, N, taps_dev, K, n_hrf, cost_dev, 0, stream,
                              "pb_hrf_cost");
}
int pb_hrf_cost_d(const double* z_dev, int64_t ldz, const double* y_dev, int64_t ldy, int V, int N,
                  const double* taps_dev, int K, int n_hrf, double* cost_dev, void* stream) {
  return hrf_cost_impl<double>(z_dev, ldz, y_dev, ldy, V, N, taps_dev, K, n_hrf, cost_dev, 0, stream,
                               "pb_hrf_cost_d");
}
int pb_hrf_cost_pv(const double* z_dev, int64_t ldz, const float* y_dev, int64_t ldy, int V, int N,
                   const double* taps_dev, int K, int n_hrf, double* cost_dev, void* stream) {
  return hrf_cost_impl<float>(z_dev, ldz, y_dev, ldy, V, N, taps_dev, K, n_hrf, cost_dev, 1, stream,
                              "pb_hrf_cost_pv");
}
int pb_hrf_cost_pv_d(const double* z_dev, int64_t ldz, const double* y_dev, int64_t ldy, int V,
                     int N, const double* taps_dev, int K, int n_hrf, double* cost_dev,
                     void* stream) {
  return hrf_cost_impl<double>(z_dev, ldz, y_dev, ldy, V, N, taps_dev, K, n_hrf, cost_dev, 1, stream,
                               "pb_hrf_cost_pv_d");
}

int pb_gram_frobenius(const double* taps_dev, int64_t ldt, int P, int K, int N, double* out_dev,
                      void* stream) {
  if (P < 0 || K < 1 || N < 1 || (P > 1 && ldt < K))
    return fail(PB_ERR_INVALID, "pb_gram_frobenius: bad size");
  if ((int64_t)N + 8 > LDS_DOUBLES_MAX) return fail(PB_ERR_INVALID, "pb_gram_frobenius: exceeds LDS");
  if (P == 0) return PB_OK;
  if (!taps_dev || !out_dev) return fail(PB_ERR_INVALID, "pb_gram_frobenius: NULL pointer");
  if (K <= N && K <= 1024) {                     // the FIR form: O(K^2 + N) per HRF
    hipLaunchKernelGGL(pb::gram_frobenius_fir_kernel, dim3(P), dim3(64), (size_t)3 * K * sizeof(double),
                       (hipStream_t)stream, taps_dev, ldt, K, N, out_dev);
    return check_launch("gram_frobenius_fir_kernel");
  }
  const size_t lds = (size_t)(N + 8) * sizeof(double);
  hipLaunchKernelGGL(pb::gram_frobenius_kernel, dim3(P), dim3(pb::GEN_THREADS), lds,
                     (hipStream_t)stream, taps_dev, ldt, K < N ? K : N, N, out_dev);
  return check_launch("gram_frobenius_kernel");
}

int pb_lambda_max(const float* y_dev, int64_t ldy, int V, int N, const double* taps_dev, int K,
                  double* out_dev, void* stream) {
  return lambda_max_impl<float>(y_dev, ldy, V, N, taps_dev, K, out_dev, stream, "pb_lambda_max");
}
int pb_lambda_max_d(const double* y_dev, int64_t ldy, int V, int N, const double* taps_dev, int K,
                    double* out_dev, void* stream) {
  return lambda_max_impl<double>(y_dev, ldy, V, N, taps_dev, K, out_dev, stream, "pb_lambda_max_d");
}

int pb_inf_norm(const double* x_dev, int64_t ldx, double* out_dev, int64_t ldo, int V, int64_t n,
                void* stream) {
  if (V < 0 || n < 1 || (V > 1 && (ldx < n || ldo < n)))
    return fail(PB_ERR_INVALID, "pb_inf_norm: bad size");
  if (V == 0) return PB_OK;
  if (!x_dev || !out_dev) return fail(PB_ERR_INVALID, "pb_inf_norm: NULL pointer");
  hipLaunchKernelGGL(pb::inf_norm_kernel, dim3(V), dim3(pb::GEN_THREADS), 0, (hipStream_t)stream,
                     x_dev, ldx, out_dev, ldo, n);
  return check_launch("pb_inf_norm");
}

int64_t pb_hrf_normal_eq_len(int K) { return K >= 1 ? (int64_t)pb::ne_len(K) : 0; }

int pb_hrf_normal_eq(const double* z_dev, int64_t ldz, const float* y_dev, int64_t ldy, int V,
                     int N, int K, int per_voxel, double* work_dev, int64_t work_len,
                     double* out_dev, void* stream) {
  return normal_eq_impl<float>(z_dev, ldz, y_dev, ldy, V, N, K, per_voxel, work_dev, work_len,
                               out_dev, stream, "pb_hrf_normal_eq");
}
int pb_hrf_normal_eq_d(const double* z_dev, int64_t ldz, const double* y_dev, int64_t ldy, int V,
                       int N, int K, int per_voxel, double* work_dev, int64_t work_len,
                       double* out_dev, void* stream) {
  return normal_eq_impl<double>(z_dev, ldz, y_dev, ldy, V, N, K, per_voxel, work_dev, work_len,
                                out_dev, stream, "pb_hrf_normal_eq_d");
}

int pb_theta_fit(const double* ne_dev, int64_t ldne, int M, int K, const double* t_dev,
                 double a_peak, double loc_peak, double a_under, double loc_under, double ratio,
                 double lo, double hi, int n_refine, double* theta_dev, double* cost_dev,
                 double* taps_dev, int64_t ldt, void* stream) {
  if (M < 0 || K < 1 || K > 127 || n_refine < 1 || !(a_peak > 0.0) || !(a_under > 0.0) ||
      !(lo <= hi) || (M > 1 && ldne < pb::ne_len(K)) || (taps_dev && M > 1 && ldt < K))
    return fail(PB_ERR_INVALID, "pb_theta_fit: bad argument");
  const int64_t nd = (int64_t)pb::ne_len(K) + 64 * (int64_t)K + 256;
  if (nd > LDS_DOUBLES_MAX) return fail(PB_ERR_INVALID, "pb_theta_fit: K=%d exceeds LDS", K);
  if (M == 0) return PB_OK;
  if (!ne_dev || !t_dev || !theta_dev || !cost_dev)
    return fail(PB_ERR_INVALID, "pb_theta_fit: NULL pointer");
  pb::HrfModel hm{a_peak, loc_peak, lgamma(a_peak), a_under, loc_under, lgamma(a_under), ratio,
                  pb::hrf_int_power(a_peak), pb::hrf_int_power(a_under), std::exp(-lgamma(a_peak)),
                  std::exp(-lgamma(a_under))};
  hipLaunchKernelGGL(pb::theta_fit_kernel, dim3(M), dim3(256), (size_t)nd * sizeof(double),
                     (hipStream_t)stream, ne_dev, ldne, M, K, t_dev, hm, lo, hi, n_refine, theta_dev,
                     cost_dev, taps_dev, ldt, 0, (double*)nullptr, 0.0, (double*)nullptr);
  return check_launch("pb_theta_fit");
}

int pb_theta_fit_step(const double* msg_dev, int K, const double* t_dev, double a_peak,
                      double loc_peak, double a_under, double loc_under, double ratio, double lo,
                      double hi, int n_refine, int N, double lbda, double* theta_dev,
                      double* cost_dev, double* taps_dev, double* step_dev, double* jcost_dev,
                      void* stream) {
  if (K < 1 || K > 127 || N < K || n_refine < 1 || !(a_peak > 0.0) || !(a_under > 0.0) || !(lo <= hi))
    return fail(PB_ERR_INVALID, "pb_theta_fit_step: bad argument");
  const int64_t nd = (int64_t)pb::ne_len(K) + 1 + 64 * (int64_t)K + 256;
  if (nd > LDS_DOUBLES_MAX) return fail(PB_ERR_INVALID, "pb_theta_fit_step: K=%d exceeds LDS", K);
  if (!msg_dev || !t_dev || !theta_dev || !cost_dev || !taps_dev || !step_dev || !jcost_dev)
    return fail(PB_ERR_INVALID, "pb_theta_fit_step: NULL pointer");
  pb::HrfModel hm{a_peak, loc_peak, lgamma(a_peak), a_under, loc_under, lgamma(a_under), ratio,
                  pb::hrf_int_power(a_peak), pb::hrf_int_power(a_under), std::exp(-lgamma(a_peak)),
                  std::exp(-lgamma(a_under))};
  hipLaunchKernelGGL(pb::theta_fit_kernel, dim3(1), dim3(256), (size_t)nd * sizeof(double),
                     (hipStream_t)stream, msg_dev, (int64_t)pb::ne_len(K) + 1, 1, K, t_dev, hm, lo, hi,
                     n_refine, theta_dev, cost_dev, taps_dev, (int64_t)K, N, step_dev, lbda, jcost_dev);
  return check_launch("pb_theta_fit_step");
}

int pb_hrf_normal_eq_w(const double* w_dev, int64_t ldw, const float* y_dev, int64_t ldy, int V,
                       int N, int K, double* work_dev, int64_t work_len, double* out_dev,
                       void* stream) {
  if (V < 0 || N < 1 || K < 1 || K > 127 || ldw < N || ldy < N)
    return fail(PB_ERR_INVALID, "pb_hrf_normal_eq_w: bad size (V=%d N=%d K=%d)", V, N, K);
  const int ne = pb::ne_len(K) + 1;
  const int64_t nd = (int64_t)pb::ne_sum_lds_doubles(N, K);
  if (nd > LDS_DOUBLES_MAX) return fail(PB_ERR_INVALID, "pb_hrf_normal_eq_w: N=%d K=%d exceeds LDS", N, K);
  if (!out_dev) return fail(PB_ERR_INVALID, "pb_hrf_normal_eq_w: NULL output");
  int blocks = V < NE_MAX_BLOCKS ? V : NE_MAX_BLOCKS;
  if (work_len / ne < blocks) blocks = (int)(work_len / ne);
  if (V > 0) {
    if (blocks < 1 || !work_dev)
      return fail(PB_ERR_INVALID, "pb_hrf_normal_eq_w: work buffer must hold at least %d doubles", ne);
    if (!w_dev || !y_dev) return fail(PB_ERR_INVALID, "pb_hrf_normal_eq_w: NULL pointer");
    if (ne_wave_form(N, K)) {
      blocks = ne_wave_blocks(V, blocks);
      const size_t lds = (size_t)pb::ne_wave_lds_doubles(N, K) * sizeof(double);
      if (K * K <= 64 * 12 && N <= 64 * 5)
        hipLaunchKernelGGL((pb::normal_eq_wave_kernel<float, true, 12, 5>), dim3(blocks), dim3(pb::NE_THREADS), lds,
                           (hipStream_t)stream, w_dev, ldw, y_dev, ldy, V, N, K, work_dev);
      else
        hipLaunchKernelGGL((pb::normal_eq_wave_kernel<float, true>), dim3(blocks), dim3(pb::NE_THREADS), lds,
                           (hipStream_t)stream, w_dev, ldw, y_dev, ldy, V, N, K, work_dev);
    } else {
      hipLaunchKernelGGL((pb::normal_eq_sum_kernel<float, true>), dim3(blocks), dim3(pb::NE_THREADS),
                         (size_t)nd * sizeof(double), (hipStream_t)stream, w_dev, ldw, y_dev, ldy, V, N, K,
                         work_dev);
    }
  } else {
    blocks = 0;
  }
  hipLaunchKernelGGL(pb::normal_eq_reduce_kernel, dim3(ne), dim3(pb::NE_THREADS), 0,
                     (hipStream_t)stream, work_dev, blocks, ne, out_dev);
  return check_launch("pb_hrf_normal_eq_w");
}


int pb_fista_solve_pp(const float* y_dev, int64_t ldy, double* w_dev, int64_t ldw, int P, int N,
                      const double* taps_dev, int64_t ldt, int K, const double* step_dev,
                      double lbda, const double* lbda_dev, const double* betas_dev, int n_iter,
                      int stop_mode, double tol, int32_t* n_done_dev, unsigned flags, void* stream) {
  if (P < 0 || N < 1 || K < 1 || n_iter < 0)
    return fail(PB_ERR_INVALID, "pb_fista_solve_pp: bad size (P=%d N=%d K=%d n_iter=%d)", P, N, K, n_iter);
  if (P > (1 << 25)) return fail(PB_ERR_INVALID, "pb_fista_solve_pp: more than 2^25 problems per launch");
  if (ldy < N || ldw < N || (ldt != 0 && ldt < K))
    return fail(PB_ERR_INVALID, "pb_fista_solve_pp: leading dimension too small");
  if (P == 0) return PB_OK;
  if (!y_dev || !w_dev || !taps_dev || !step_dev || (n_iter > 0 && !betas_dev))
    return fail(PB_ERR_INVALID, "pb_fista_solve_pp: NULL pointer");
  if (stop_mode != PB_STOP_NONE && stop_mode != PB_STOP_LOOPS)
    return fail(PB_ERR_INVALID, "pb_fista_solve_pp: stop_mode must be PB_STOP_NONE or PB_STOP_LOOPS");

  pb::FistaArgs a;
  a.y = y_dev; a.y64 = nullptr; a.ldy = ldy; a.w = w_dev; a.ldw = ldw; a.lbda_vec = lbda_dev;
  a.betas = betas_dev; a.J = nullptr; a.J64 = nullptr; a.ldj = 0; a.n_done = n_done_dev;
  a.step = 0.0; a.lbda = lbda; a.tol = tol;
  a.y_rep = 1; a.P = P; a.N = N; a.n_iter = n_iter; a.stop_mode = stop_mode;
  a.taps_pp = taps_dev; a.ldt = ldt; a.step_vec = step_dev; a.step_shared = (ldt == 0); a.K = K;
  a.p0 = 0;
  a.cold = (flags & PB_FLAG_COLD_START) ? 1 : 0;
  a.rho_guard = (flags & PB_FLAG_NO_RHO_GUARD) ? 0 : 1;

  // ONE shared HRF, no stop rule, series of 311 .. 1 280 scans (round 5): whole passes on the matrix-pipe form split over two
  // (up to 640 scans) or four waves, which read the HRF and its step from device memory; the rest, and what the guards hand
  // back, on the one-problem-per-wave form
  if (N > MFMA1_NMAX && ldt == 0 && stop_mode == PB_STOP_NONE && n_done_dev && K <= 65 &&
      !(flags & (PB_FLAG_FORCE_GENERIC | PB_FLAG_FORCE_PAIR | PB_FLAG_NO_PAIR | PB_FLAG_NO_MFMA | PB_FLAG_FORCE_WIDE | PB_FLAG_DIRECT_FIR))) {
    const bool four = N > 640;
    const mfma2_launch_fn split = four ? pick_mfma4(N, K, false) : pick_mfma2(N, K, false);
    const WideEntry* we = pick_wide(N, K);
    if (split && we && (four || P >= mfma2_long_min_p(K) || (flags & PB_FLAG_FORCE_MFMA2))) {
      const bool all = (flags & (PB_FLAG_ONE_LAUNCH | PB_FLAG_FORCE_MFMA2)) != 0;
      const int base = four ? mfma4_base(P, all) : mfma2_long_base(P, all);
      pb::FistaArgs b = a;
      if (base > 0) {
        b.P = base;
        if (split(b, nullptr, K, false, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve_pp: split matrix-pipe kernel rejected the launch");
        const int rc = check_launch(four ? "fista_mfma4_kernel(shared taps)" : "fista_mfma2_kernel(shared taps)");
        if (rc != PB_OK) return rc;
      }
      if (base < P) {
        b = a;
        b.p0 = base;
        if (we->fn_pp(b, stop_mode, (hipStream_t)stream) != 0) return fail(PB_ERR_INVALID, "pb_fista_solve_pp: launch rejected");
        const int rc = check_launch("fista_fast_kernel(wide, pp, remainder)");
        if (rc != PB_OK) return rc;
      }
      if (base > 0 && !(flags & PB_FLAG_CERT_NO_RESOLVE)) {
        b = a;
        b.P = base;
        b.only_flagged = 1;
        if (we->fn_pp(b, stop_mode, (hipStream_t)stream) != 0) return fail(PB_ERR_INVALID, "pb_fista_solve_pp: launch rejected");
        return check_launch("fista_fast_kernel(wide, pp, re-solve)");
      }
      return PB_OK;
    }
  }
  const FastEntry* fe = (flags & PB_FLAG_FORCE_GENERIC) ? nullptr : pick_fast(N, K);
  if (fe) {
    // ONE shared HRF, no stop rule: the pair form (taps read from device memory) wherever the
    // dispatch model of plain solves would use it, the per-problem-taps kernel elsewhere
    if (ldt == 0 && stop_mode == PB_STOP_NONE && fe->fn_pair_dev && P >= 2 &&
        !(flags & (PB_FLAG_NO_PAIR | PB_FLAG_DIRECT_FIR))) {
      // (the remainder of the whole rounds on the single-row or the one-problem-per-wave kernel,
      // both reading the shared HRF through their per-problem-taps form)
      const WideEntry* ws = pick_wide_small(N, K);
      auto run = [&](int form, int p0, int p1) -> int {
        pb::FistaArgs b = a;
        b.p0 = p0;
        b.P = p1;
        const int bad = (form == FORM_PAIR)   ? fe->fn_pair_dev(b, (hipStream_t)stream)
                        : (form == FORM_WIDE) ? ws->fn_pp(b, stop_mode, (hipStream_t)stream)
                                              : fe->fn_pp(b, stop_mode, (hipStream_t)stream);
        if (bad) return fail(PB_ERR_INVALID, "pb_fista_solve_pp: launch rejected");
        return check_launch(form == FORM_PAIR ? "fista_pair_ffa_kernel(shared taps)" : "fista_fast_kernel(pp)");
      };
      // whole rounds (and a remainder above half a round) on the matrix-pipe form, which reads the
      // shared HRF and its step from device memory like the pair form; what the whole rounds leave goes to the
      // split form (fista_mfma2.h: half the latency for up to half a round) where plan_pieces_mfma would put it
      int base = 0;
      const mfma_launch_fn mfma = (n_done_dev && !(flags & (PB_FLAG_FORCE_PAIR | PB_FLAG_NO_MFMA))) ? pick_mfma(N, K) : nullptr;
      const mfma2_launch_fn mfma2 = (mfma && K <= MFMA_K2) ? pick_mfma2(N, K) : nullptr;
      int base2 = 0;                               // problems [base, base2) on the split form
      if (mfma) {
        const int round = (int)wave_slots() * 8, half = round / 2;
        base = (P / round) * round;
        const int R = P - base;
        base2 = base;
        if (flags & PB_FLAG_ONE_LAUNCH) base = base2 = P;
        else if (mfma2 && R > MFMA2_MIN_R && R <= half) base2 = P;
        else if (mfma2 && ws && R > half && R - half <= (int)wave_slots()) base2 = base + half;
        else if (R > half) base = base2 = P;
        if (base > 0) {
          pb::FistaArgs b = a;
          b.P = base;
          if (mfma(b, nullptr, K, false, (hipStream_t)stream) != 0)
            return fail(PB_ERR_INVALID, "pb_fista_solve_pp: matrix-pipe kernel rejected the launch");
          const int rc = check_launch("fista_mfma_kernel(shared taps)");
          if (rc != PB_OK) return rc;
        }
        if (base2 > base) {
          pb::FistaArgs b = a;
          b.p0 = base;
          b.P = base2;
          if (mfma2(b, nullptr, K, false, (hipStream_t)stream) != 0)
            return fail(PB_ERR_INVALID, "pb_fista_solve_pp: split matrix-pipe kernel rejected the launch");
          const int rc = check_launch("fista_mfma2_kernel(shared taps)");
          if (rc != PB_OK) return rc;
          base = base2;
        }
      }
      if (base < P) {
        Plan pl{0, FORM_GENERIC, FORM_PAIR};
        if (!(flags & PB_FLAG_FORCE_PAIR))
          pl = plan_plain(P - base, P - base >= 2, ws != nullptr, (flags & PB_FLAG_ONE_LAUNCH) != 0);
        if (pl.n_main > 0) {
          const int rc = run(pl.main_form, base, base + pl.n_main);
          if (rc != PB_OK) return rc;
        }
        const int rc = run(pl.tail_form, base + pl.n_main, P);
        if (rc != PB_OK) return rc;
      }
      if (base > 0 && !(flags & PB_FLAG_CERT_NO_RESOLVE)) {   // problems the matrix-pipe form handed back (n_done = -1)
        pb::FistaArgs b = a;
        b.P = base;
        b.only_flagged = 1;
        if (fe->fn_pp(b, stop_mode, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve_pp: launch rejected");
        return check_launch("fista_fast_kernel(pp, re-solve)");
      }
      return PB_OK;
    }
    // one HRF per problem: single-row form, or one problem per wave where that finishes first
    // (small batches are latency-bound: 0.37 ms against 0.93 ms per 500 iterations up to 2 048)
    if (!(flags & (PB_FLAG_NO_PAIR | PB_FLAG_FORCE_PAIR | PB_FLAG_ONE_LAUNCH))) {
      const WideEntry* ws = pick_wide_small(N, K);
      if (ws && best_form(P, false, true) == FORM_WIDE) {
        if (ws->fn_pp(a, stop_mode, (hipStream_t)stream) != 0)
          return fail(PB_ERR_INVALID, "pb_fista_solve_pp: launch rejected");
        return check_launch("fista_fast_kernel(wide, pp)");
      }
    }
    if (fe->fn_pp(a, stop_mode, (hipStream_t)stream) != 0)
      return fail(PB_ERR_INVALID, "pb_fista_solve_pp: launch rejected");
    return check_launch("fista_fast_kernel(pp)");
  }
  if (!(flags & PB_FLAG_FORCE_GENERIC)) {
    if (const WideEntry* we = pick_wide(N, K)) {
      if (we->fn_pp(a, stop_mode, (hipStream_t)stream) != 0)
        return fail(PB_ERR_INVALID, "pb_fista_solve_pp: launch rejected");
      return check_launch("fista_fast_kernel(wide, pp)");
    }
  }
  if (flags & PB_FLAG_FORCE_FAST)
    return fail(PB_ERR_INVALID, "pb_fista_solve_pp: no register-resident kernel for N=%d K=%d", N, K);
  const int64_t nd = 3 * (int64_t)N + K + 2 * pb::GEN_WAVES;
  if (nd > LDS_DOUBLES_MAX) return fail(PB_ERR_INVALID, "pb_fista_solve_pp: N=%d K=%d exceeds LDS", N, K);
  hipLaunchKernelGGL((pb::fista_generic_kernel<false>), dim3(P), dim3(pb::GEN_THREADS),
                     (size_t)nd * sizeof(double), (hipStream_t)stream, a, taps_dev, K, 0);
  return check_launch("fista_generic_kernel(pp)");
}

int64_t pb_fista_path_work_len(int P) { return P >= 0 ? work_layout(P, P).total : 0; }     // (enough for any y_rep)

int pb_fista_solve_path(const float* y_dev, int64_t ldy, int y_rep, double* w_dev, int64_t ldw, int P, int N,
                        const double* taps_host, const double* taps_dev, int K, double step,
                        const double* lbda_dev, const double* lmax_dev, double dense_ratio,
                        const double* betas_dev, int n_iter, int32_t* n_done_dev, int32_t* work_dev,
                        int64_t work_len, unsigned flags, void* stream) {
  // (round 4's entry point for regularisation paths; since round 5 every call shape is partitioned:
  // pb_fista_solve_ex with per-problem lambdas, the caller's lambda_max and workspace)
  if (P > 0 && (!lbda_dev || !n_done_dev))
    return fail(PB_ERR_INVALID, "pb_fista_solve_path: NULL pointer (per-problem lambdas and n_done are required)");
  return solve_impl(y_dev, ldy, y_rep, w_dev, ldw, P, N, taps_host, taps_dev, K, step, 0.0, lbda_dev, betas_dev, n_iter,
                    nullptr, 0, PB_STOP_NONE, 0.0, 6, n_done_dev, flags, stream, lmax_dev, dense_ratio, work_dev, work_len);
}

}  // extern "C"
