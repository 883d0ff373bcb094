/*
 * pybold_hip.h -- C ABI of the MI355X (gfx950) implementation of pyBOLD's
 * deconvolution hot path.
 *
 * The reference (hcherkaoui/pybold) has no FFI: its "plugin surface" is Python
 * duck typing (objects with .op(x)/.adj(x)) and plain functions on 1-D NumPy
 * arrays.  Each entry point below therefore cites the reference *Python*
 * interface it stands in for (file:line in the reference checkout); the
 * binding a maintainer would add on the reference side is the ctypes stub in
 * INTEGRATION.md.
 *
 * Conventions
 *   - every pointer named *_dev is a DEVICE pointer (HBM of the current HIP
 *     device); pointers named *_host are host pointers read synchronously
 *     before the call returns; nothing else is dereferenced on the host;
 *   - matrices are row-major, one row per problem ("voxel"), leading dimension
 *     ld* counted in ELEMENTS;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     all work is enqueued asynchronously in its order, no device memory is allocated, no
 *     host synchronisation is performed (graph-capture safe).  One exception to "on it":
 *     pb_fista_solve may run the remainder of a plain solve on an internal side stream
 *     (one per device, created on first use) that forks from and joins back into `stream`
 *     through events, so the caller sees ordinary stream order; not while `stream` is being
 *     captured, and never with PB_FLAG_ONE_STREAM;
 *   - return value 0 = success; a negative value = error, text available from
 *     pb_last_error() (thread-local).  No entry point ever falls back to a
 *     CPU computation.
 */
#ifndef PYBOLD_HIP_H
#define PYBOLD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PB_OK 0
#define PB_ERR_INVALID (-1)   /* bad argument (shape, NULL pointer, size limit) */
#define PB_ERR_HIP (-2)       /* HIP runtime error at launch */

/* flags for pb_fista_solve */
#define PB_FLAG_FORCE_GENERIC 1u  /* use the any-size LDS kernel even when a
                                     register-resident specialisation exists */
#define PB_FLAG_FORCE_FAST 2u     /* fail instead of using the generic kernel */
#define PB_FLAG_NO_PAIR 4u        /* plain solves: always one problem per DPP row
                                     (fista_fast_kernel) */
#define PB_FLAG_FORCE_PAIR 8u     /* plain solves: always two problems per DPP row
                                     (fista_pair_kernel); default = whichever the dispatch
                                     model expects to finish first for this problem count */
#define PB_FLAG_FORCE_WIDE 16u    /* always one problem per wave (fista_fast_kernel, 64 lanes) */
#define PB_FLAG_DIRECT_FIR 64u    /* pair form: direct K-tap FIRs (fista_pair_kernel) instead of the
                                     2-parallel fast FIRs (fista_pair_ffa_kernel) */
#define PB_FLAG_ONE_STREAM 128u    /* never use the internal side stream (see pb_fista_solve) */
#define PB_FLAG_COLD_START 256u   /* the iterate starts from 0 (what deconv does, bold_signal.py:57): w_dev
                                     is output only -- no memset by the caller, no read by the kernel */
#define PB_FLAG_NO_CERT 512u       /* PB_STOP_WINDOW: always evaluate the rule in full (fista_fast_kernel), never
                                     the no-fire certificate of the pair form + re-solve (see pb_fista_solve) */
#define PB_FLAG_CERT_NO_RESOLVE 2048u /* diagnostic: certificate launch only; uncleared problems keep n_done = -1 */
#define PB_FLAG_FORCE_MFMA 16384u  /* the matrix-pipe form also with per-problem lambdas (see pb_fista_solve) */
#define PB_FLAG_NO_RHO_GUARD 32768u /* matrix-pipe form: keep sparse solutions (threshold / max|w| > 0.02) instead of handing them
                                     back to the float32 operators.  For INTERMEDIATE solves of an outer loop (the z-steps
                                     of the blind loop but the last): their errors, relative 1e-5..1e-4 of a still tiny
                                     iterate, are forgotten by the warm-started solves that follow */
#define PB_FLAG_FORCE_MFMA2 65536u /* the matrix-pipe form with every series split over the waves of a workgroup, one launch:
                                     two waves (fista_mfma2_kernel, 129..640 scans) or four (fista_mfma4_kernel, 641..1280);
                                     HRFs of up to 33 taps: plain, cost trace, window certificate, _loops_deconv rule; 34..65 taps
                                     (three near tiles, series of 225+ scans): the same */
#define PB_FLAG_NO_MFMA 8192u      /* plain solves: never the matrix-pipe form (fista_mfma_kernel), vector forms only */
#define PB_FLAG_FORCE_CERT 1024u   /* PB_STOP_WINDOW, wind = 6: certificate path whatever tol * n_iter is */
#define PB_FLAG_NO_PARTITION 4096u  /* never partition a call on the device (see pb_fista_solve_ex): the host-side plan of round 4 */
#define PB_FLAG_ONLY_DENSE 131072u  /* measurement aids for a partitioned call: only the dense class / only the sparse */
#define PB_FLAG_ONLY_SPARSE 262144u /*   class is solved; the other class's rows of w and n_done are left untouched */
#define PB_FLAG_NO_ILL_GUARD 524288u /* partitioned calls: do not single out ill-conditioned series (see pb_fista_solve_ex) */
#define PB_FLAG_ONE_LAUNCH 32u    /* never split a plain solve into a full-rounds launch and a
                                     remainder launch (see pb_fista_solve) */

/* early-stop rules (evaluated per problem, inside the kernel) */
#define PB_STOP_NONE 0
#define PB_STOP_LOOPS 1   /* _loops_deconv rule, pybold/bold_signal.py:267-273 */
#define PB_STOP_WINDOW 2  /* deconv windowed rule, pybold/bold_signal.py:82-95 */

/* Library version (major*10000 + minor*100 + patch). */
int pb_version(void);

/* Optional.  Creates the internal side stream of the current device now instead of at the
 * first solve that uses it (see pb_fista_solve).  Worth calling before the application creates
 * streams of its own: HIP multiplexes streams onto a few hardware queues in creation order, and
 * a side stream created after half a dozen others was measured to overlap less with the caller's
 * stream (2.52 ms instead of 2.38 ms for 12 500 problems).  Idempotent, thread-safe. */
int pb_init(void);

/* Text of the last error on the calling thread ("" if none). */
const char* pb_last_error(void);

/* 1 if a register-resident specialisation exists for (N scans, K taps), else 0 (the
 * generic kernel is used).  Host-only query. */
int pb_fista_has_fast_path(int N, int K);

/* Which kernel pb_fista_solve will run for this call shape (no flags): 0 = generic
 * LDS kernel, 1 = register-resident, one problem per 16-lane row (fista_fast_kernel),
 * 2 = register-resident, two problems per row (fista_pair_kernel), 3 = register-resident,
 * one problem per wave (long series), 4 = register-resident, 16 problems per wave, both operators
 * on the matrix pipe (fista_mfma_kernel: 129..310 scans; HRFs of up to 48 taps, the window-rule certificate up to
 * 33; one lambda for the batch; assumes n_done_dev is given), 5 = the same with every series split over the two waves
 * of a workgroup (fista_mfma2_kernel: small batches, and series of 311..640 scans), 6 = split over the four waves of a
 * workgroup (fista_mfma4_kernel: series of 641..1280 scans; HRFs of up to 65 taps).  Host-only query. */
int pb_fista_which_kernel(int N, int K, int P, int with_cost_trace, int stop_mode, int wind);

/* How pb_fista_solve (no flags) lays P problems out: problems [0, *n_main) in one launch of
 * kernel form *main_form (whole rounds of waves; 0 problems = no such launch) and
 * [*n_main, P) in a launch of *tail_form (forms numbered as pb_fista_which_kernel).
 * Host-only query; any output pointer may be NULL. */
int pb_fista_plan(int N, int K, int P, int stop_mode, int wind, int* n_main, int* main_form,
                  int* tail_form);
/* The same for a call that passes `flags` (PB_FLAG_NO_MFMA: the plan of the vector forms). */
int pb_fista_plan_ex(int N, int K, int P, int stop_mode, int wind, unsigned flags, int* n_main,
                     int* main_form, int* tail_form);

/*
 * Fused FISTA-like solver: n_iter iterations of the recurrence of
 *   deconv (fixed-lambda loop)   pybold/bold_signal.py:62-72
 *   _loops_deconv                pybold/bold_signal.py:259-276
 * for P independent problems, state resident on chip, in ONE launch -- or, for a plain
 * solve (no stop rule) whose problem count does not fill the machine evenly, in up to four:
 * the whole rounds of waves on the densest kernel form, then the remainder on whichever form
 * finishes it first (PB_FLAG_ONE_LAUNCH turns that off).  A remainder above half a round of
 * pair waves becomes half a round on `stream` with the rest beside it on the side stream
 * (PB_FLAG_ONE_STREAM turns that off; pb_fista_plan reports the split):
 *
 *   u = w - step * H^T (H w - y)      H = toeplitz(taps) . cumsum
 *   p = soft(u, lbda_p * step)        pybold/linear.py:73-113, convolution.py:105-132
 *   w = p + beta_k (p - u)            (beta_0 = 0; see SURVEY.md 8a for the aliasing)
 *
 * y_dev      float32 [ceil(P / y_rep)][ldy]: observed series; problem p reads
 *            row p / y_rep (y_rep > 1 = several lambdas per voxel).
 * w_dev      float64 [P][ldw]: in = warm start (w_0), out = final iterate.  P <= 2^25.
 * taps_host  float64 [K]: HRF taps (shared by all problems), host copy: the
 *            register-resident kernel receives them as kernel arguments.
 * taps_dev   the same taps in device memory (used by the generic LDS kernel;
 *            may be NULL when pb_fista_has_fast_path(N, K) and the stop rule is
 *            not PB_STOP_WINDOW).
 * step       1/L, L = 0.9*rho (deconv, :52-53) or ||A^T A||_F (_loops_deconv, :253-254).
 * lbda, lbda_dev  regularisation: per-problem float64 [P] if lbda_dev != NULL,
 *            else the scalar.
 * betas_dev  float64 [n_iter]: momentum factors (t_k - 1)/t_{k+1} for the
 *            iterations of THIS launch (:68-69).
 * J_dev      optional float32 [P][ldj], ldj >= n_iter: un-normalised cost
 *            0.5||H w_{k+1} - y||^2 + lbda ||w_{k+1}||_1 after each iteration (:74-77).
 * stop_mode, tol, wind, n_done_dev: optional per-problem early stopping
 *            (PB_STOP_*); n_done_dev int32 [P] receives the number of
 *            iterations executed.  PB_STOP_NONE ignores tol/wind; n_done_dev
 *            may be NULL.
 *            PB_STOP_WINDOW with wind = 6 (the reference default), n_done_dev given and
 *            tol * n_iter < 0.5 (the rule is not expected to fire: its criterion decays like
 *            ~0.9/k): the problems run on the two-problems-per-row form, which PROVES per
 *            iteration that the rule does not fire (a lower bound of its numerator from one
 *            tracked sample per lane, an upper bound of its denominator from norms of the
 *            iterate); problems it cannot clear come out with n_done = -1, iterate untouched,
 *            and are re-solved at once, exactly, by a second launch of the single-row form --
 *            results are those of the full rule either way.  PB_FLAG_NO_CERT / _FORCE_CERT
 *            override the choice.  Other wind values, or series above 16*20 scans, evaluate the
 *            rule in full (single-row form; LDS kernel beyond its limits).
 *
 * ARITHMETIC.  The iterate, the gradient step, the threshold and the momentum are float64 in every
 * kernel form; y is float32.  The two linear operators (H, H^T) run in one of two arithmetics:
 *   (a) float32 on the vector pipe (fista_fast / fista_pair(_ffa) kernels), or
 *   (b) on the matrix pipe (fista_mfma_kernel): operands split into two float16 parts (22 bits, low part
 *       rounded to nearest), three v_mfma_f32_16x16x32_f16 products per tile, float32 accumulation; every
 *       series scaled by a per-problem power of two into the float16 range (exact: the problem is
 *       scale-covariant).
 * WHICH SHAPE RUNS WHERE (calls of a few thousand problems or more; pb_fista_which_kernel / pb_fista_plan_ex answer for a shape):
 *   scans N      taps K    plain / cost trace   window rule, wind 6      _loops_deconv rule    form
 *   <= 128       <= 32     (a)                  (a) certificate          (a)                   two problems per row / single row
 *   129 .. 310   <= 33     (b)                  (b) certificate          (b) in full           fista_mfma_kernel, one wave per 16 problems
 *   129 .. 310   34 .. 48  (b)                  (b) from 225 scans on    (b) from 225 scans on ... with three near tiles (stop rules: split form)
 *   311 .. 640   <= 33     (b)                  (b) certificate          (b) in full           fista_mfma2_kernel, two waves per 16 problems
 *   311 .. 640   34 .. 48  (b)                  (b) certificate          (b) in full           ... with three near tiles
 *   641 .. 1280  <= 33     (b)                  (b) certificate          (b) in full           fista_mfma4_kernel, four waves per 16 problems
 *   641 .. 1280  34 .. 48  (b)                  (b) certificate          (b) in full           ... with three near tiles
 *   longer series, longer HRFs, other windows, a cost trace beside the _loops_deconv rule: (a), one problem per wave up to
 *   2 432 scans, the LDS kernel beyond (and for the window rule beyond 1 280 scans, for 34+ taps beyond 1 280).  Per-problem HRFs (pb_fista_solve_pp, ldt != 0): (a).  A machine-filling batch that lands
 *   on (a) although (b) serves neighbouring shapes is 1.5 .. 4x below the matrix-pipe rate; the Python layer says so once.
 *
 * (b) is chosen for plain solves (PB_STOP_NONE, or PB_STOP_WINDOW as a certificate with tol * n_iter < 0.02)
 * of 129..1280 scans with ONE lambda for the call (lbda_dev == NULL, or per-problem lambdas with the dense
 * part of a regularisation path: see below) -- AND ONLY IF n_done_dev IS GIVEN: the matrix-pipe kernel
 * reports through n_done the problems it must not keep, namely
 *   - range: a residual fragment reached 2^15 or |sigma w| reached 60000 (checked after the first pass of a
 *     warm start, every 8th iteration and at the end), and
 *   - scale: an all-zero series with a warm start (no scale to bring the iterate into the float16 range), and
 *   - accuracy: the solution is too sparse for 22-bit operators, lbda * step > 0.02 * max|w| at the end (an
 *     error eps in the gradient moves a solution entry by ~eps * threshold, so the relative error of ANY
 *     arithmetic grows with threshold / max|w|; measured <= 3e-6 on diff_z below that bound, up to 1.5e-5
 *     above 0.1; PB_FLAG_NO_RHO_GUARD switches this one off),
 * leaves their iterate untouched (n_done = -1) and the same call re-solves them on (a) before it returns,
 * so the caller only ever sees finished problems (n_done = n_iter).  With n_done_dev == NULL the call
 * silently stays on (a): ~1.5x slower at 100k voxels, same results within 1e-6.  Either way the result
 * is within 1e-5 (relative L2 per problem on diff_z, z, x) of the float64 reference recurrence
 * (tests/test_gpu_round4.py: DC baselines up to 1000x the fluctuation, heavy tails, SNR -10..+30 dB,
 * lambda / lambda_max in [1e-3, 1]; profiles/r4_adversarial_sweep.txt).
 */
int pb_fista_solve(const float* y_dev, int64_t ldy, int y_rep,
                   double* w_dev, int64_t ldw, int P, int N,
                   const double* taps_host, const double* taps_dev, int K,
                   double step, double lbda, const double* lbda_dev,
                   const double* betas_dev, int n_iter,
                   float* J_dev, int64_t ldj,
                   int stop_mode, double tol, int wind, int32_t* n_done_dev,
                   unsigned flags, void* stream);

/*
 * pb_fista_solve with the caller's workspace: the call that NEVER solves a problem twice (round 5).
 *
 * The matrix-pipe form (b) holds eps only where the solution is dense (threshold <= 0.02 max|w|, see above); for a
 * lambda near lambda_max,v = || H^T y_v ||_inf it used to be solved on (b), handed back and solved again on (a), one
 * handed-back problem per wave -- slower than never using (b) (profiles/r4_path_partition.txt).  Since round 5 every
 * call that (b) can carry (129..1280 scans; plain, cost trace, window-rule certificate, _loops_deconv rule; ONE lambda or
 * one per problem) is PARTITIONED ON THE DEVICE before it is solved, without a host synchronisation:
 *     lbda_p < dense_ratio * lmax[p / y_rep]    -> dense class:  matrix-pipe form (b)
 *     otherwise                                 -> sparse class: float32 vector forms (a)
 * (lmax = pb_lambda_max of every series: lmax_dev if the caller has it -- a regularisation path does --, else one extra
 * pass over y, ~0.05 ms per 100 k series).  Three small launches build two order-preserving index lists in work_dev; a
 * one-thread kernel turns their lengths into the launch plans pb_fista_plan would give a call of that many problems
 * (csrc/plan.h: the same functions compiled for the device); every kernel form is then launched with a worst-case
 * grid and reads the slots it solves from device memory -- the waves of an empty launch leave at once.  What a guard
 * or certificate still hands back (n_done = -1) is compacted the same way and re-solved by the exact vector forms at
 * full occupancy.  Results, n_done and J are those of pb_fista_solve; the call allocates nothing.
 *
 * CONDITIONING GUARD.  The same pass sees ||y_v||_2, and the call classes every series on its coherence
 *     gamma_2 = lmax_v / (||y_v||_2 sum|c| / sqrt(N)),   c = cumsum(taps)        (block signals at SNR 1 dB: 0.16 .. 0.5)
 * -- a series most of whose energy the operator does not see (alternating signs, fast sinusoids, a signal under a strong fast
 * carrier) loses digits in every arithmetic narrower than float64: gamma_2 < 1e-2 -> float64 (fista_exact_kernel on their list; the LDS kernel, which needs taps_dev, for a window other than 6),
 * gamma_2 < 7e-2 (3e-2 for 311..640 scans, 2e-2 for 641..1280, HRFs of up to 33 taps) -> the float32 vector forms whatever lambda, never (b).  5 120 such series per length through this entry point:
 * worst 2.9e-6 / 3.7e-6 on diff_z, z, x; without the guard 4e-4 on (b) and 5e-3 on (a) (profiles/r5_gamma_calibration_*.txt).
 * PB_FLAG_NO_ILL_GUARD switches it off.  Calls that are not partitioned are not guarded.
 *
 * work_dev    int32 scratch of at least pb_fista_work_len(P, y_rep) entries, 8-byte aligned (contents meaningless
 *             afterwards); NULL or too small: no partition (the plan of pb_fista_solve with PB_FLAG_NO_PARTITION).
 * lmax_dev    ignored since the conditioning guard (the call makes its own lambda_max, which also carries the guard's marks);
 *             kept in the signature.       dense_ratio <= 0: PB_PATH_DENSE_RATIO(_LONG).
 * pb_fista_solve itself partitions too, on a workspace of the library's own (one per device and stream, grown on
 * demand with hipMalloc -- the one allocation this library makes; never under stream capture, where it runs unpartitioned).
 * Calls of fewer than 4 096 problems, shapes outside form (b), PB_FLAG_FORCE_* / _ONE_LAUNCH / _NO_PARTITION: the
 * host-side plan as before.
 */
int64_t pb_fista_work_len(int P, int y_rep);
/* Host-only query: the device-side plan of a LIST of n problems (kind 1: a dense class, matrix-pipe form + vector
 * remainder; kind 2: vector forms only; kind 3: a PARTITIONED CALL of n_max problems, n of them dense -- ranges are
 * positions in the call's list array, dense problems first) as the slots [ranges[2c], ranges[2c+1]) of each of the PB_CAND_COUNT candidate
 * launches (csrc/plan.h: MFMA, PAIR0, FAST0 | fork | MFMA2, PAIR1, FAST1, WIDE | side stream: WIDE0, WIDE1, FAST), and
 * the grid bound of each candidate for lists of at most n_max problems.  The same functions run on the device. */
#define PB_CAND_COUNT 10
int pb_fista_list_plan(int kind, int n, int n_max, int has_pair, int has_wide, int one_stream, int has_mfma2,
                       int beside_chunks, int32_t* ranges, int32_t* bounds);
int pb_fista_solve_ex(const float* y_dev, int64_t ldy, int y_rep,
                      double* w_dev, int64_t ldw, int P, int N,
                      const double* taps_host, const double* taps_dev, int K,
                      double step, double lbda, const double* lbda_dev,
                      const double* betas_dev, int n_iter,
                      float* J_dev, int64_t ldj,
                      int stop_mode, double tol, int wind, int32_t* n_done_dev,
                      unsigned flags, void* stream,
                      const double* lmax_dev, double dense_ratio, int32_t* work_dev, int64_t work_len);

/*
 * Regularisation path: P = V * y_rep problems (voxel v, lambda_{v,i}), lbda_dev[p] the lambda of problem p, all
 * lambdas of a voxel sharing its series (row p / y_rep of y_dev).  The reference has no such routine: its lambda
 * lists are hard-coded "already grid-search" values (examples/icassp_2019/simulation.py:113-114,
 * validation.py:60-62); each problem equals one reference call deconv(y_v, t_r, hrf, lbda=lambda_{v,i}).
 * Same recurrence, arguments and results as pb_fista_solve with lbda_dev (plain solve: no stop rule, no cost
 * trace), but the problems are PARTITIONED on the device, without a host synchronisation, by
 *     lbda_dev[p] < dense_ratio * lmax_dev[p / y_rep]         (lmax_dev: pb_lambda_max of every series)
 * -- the dense class runs on the matrix-pipe form, the sparse class (solutions of a few small entries, where the
 * 22-bit operators of that form would be handed back by its accuracy guard and solved twice) straight on the
 * two-problems-per-row float32 form; whatever a guard still hands back is re-solved exactly as in pb_fista_solve.
 * dense_ratio <= 0 selects PB_PATH_DENSE_RATIO (calibrated on block-signal paths: the guard hands back 1.9 % of the
 * problems at lambda / lambda_max = 0.13, 35 % at 0.20, 90 % at 0.30 for 300 scans, 0.2 % / 14 % / 70 % for 600; a problem
 * is worth trying on the matrix pipe while its chance of coming back is below 1 - t_matrix / t_vector ~ 0.35).  work_dev: int32 scratch of at least pb_fista_path_work_len(P)
 * entries (index lists and counts; contents meaningless afterwards).  lmax_dev == NULL, work_dev == NULL or a
 * shape outside the matrix-pipe form (129..310 scans, <= 33 taps): the call is pb_fista_solve with lbda_dev.
 * (Round 5: = pb_fista_solve_ex with lbda_dev, no cost trace, no stop rule; kept for callers of round 4.)
 */
/* (round 5: 0.19 for series of up to 310 scans, 0.22 for 311..640, 0.26 for 641..1280 (the guard hands back 7 % at
 * lambda / lambda_max = 0.20 there, 50 % at 0.30) -- with the handed-back problems compacted the break-even moved up from
 * round 4's 0.13: profiles/r5_dense_ratio_sweep.txt, r5_lambda_sweep_1200_scans.txt) */
#define PB_PATH_DENSE_RATIO 0.19
#define PB_PATH_DENSE_RATIO_LONG 0.22
#define PB_PATH_DENSE_RATIO_LONGER 0.26
int64_t pb_fista_path_work_len(int P);
int pb_fista_solve_path(const float* y_dev, int64_t ldy, int y_rep, double* w_dev, int64_t ldw, int P, int N,
                        const double* taps_host, const double* taps_dev, int K, double step,
                        const double* lbda_dev, const double* lmax_dev, double dense_ratio,
                        const double* betas_dev, int n_iter, int32_t* n_done_dev, int32_t* work_dev,
                        int64_t work_len, unsigned flags, void* stream);

/*
 * z = cumsum(w), x = taps * z (causal, truncated): the outputs deconv returns
 * next to diff_z (pybold/bold_signal.py:74-75,97).  float64 in, float64 out.
 * Either output pointer may be NULL.
 */
int pb_fista_outputs(const double* w_dev, int64_t ldw, int P, int N,
                     const double* taps_dev, int K,
                     double* z_dev, int64_t ldz, double* x_dev, int64_t ldx,
                     void* stream);

/*
 * Per-problem residual and sparsity of an iterate:
 *   r2[p] = || taps * cumsum(w_p) - y_p ||^2 ,  l1[p] = || w_p ||_1
 * -- the quantities the noise-driven lambda search of deconv(lbda=None) tracks
 * and feeds back into lambda (pybold/bold_signal.py:141-157).  float64 [P] each.
 */
int pb_fista_stats(const double* w_dev, int64_t ldw, const float* y_dev, int64_t ldy,
                   int y_rep, int P, int N, const double* taps_dev, int K,
                   double* r2_dev, double* l1_dev, void* stream);

/*
 * Operator surface (float64, any size that fits LDS: 3*max(n_in,n_out)+K <= 20000).
 * All of them act row-wise on V rows.
 *
 * pb_integ_op / pb_integ_adj   DiscretInteg.op / .adj      pybold/linear.py:15-43
 * pb_conv                      toeplitz_from_kernel(k, n_in, n_out) @ x
 *                                                          pybold/convolution.py:105-132
 * pb_corr                      toeplitz_from_kernel(k, n_in, n_out).T @ r
 * pb_op_forward / pb_op_adjoint  ConvAndLinear(DiscretInteg(), k, n_in, n_out).op / .adj
 *                                                          pybold/linear.py:73-113
 */
int pb_integ_op(const double* x_dev, int64_t ldx, double* out_dev, int64_t ldo,
                int V, int N, void* stream);
int pb_integ_adj(const double* x_dev, int64_t ldx, double* out_dev, int64_t ldo,
                 int V, int N, void* stream);
int pb_conv(const double* x_dev, int64_t ldx, double* out_dev, int64_t ldo,
            int V, int n_in, int n_out, const double* taps_dev, int K, void* stream);
int pb_corr(const double* r_dev, int64_t ldr, double* out_dev, int64_t ldo,
            int V, int n_in, int n_out, const double* taps_dev, int K, void* stream);
int pb_op_forward(const double* x_dev, int64_t ldx, double* out_dev, int64_t ldo,
                  int V, int n_in, int n_out, const double* taps_dev, int K, void* stream);
int pb_op_adjoint(const double* r_dev, int64_t ldr, double* out_dev, int64_t ldo,
                  int V, int n_in, int n_out, const double* taps_dev, int K, void* stream);

/*
 * Per-voxel HRF fit error  cost[v] = 0.5 || y_v - taps * z_v ||^2
 * (hrf_fit_err, pybold/bold_signal.py:217-222) for n_hrf candidate HRFs at
 * once: taps_dev float64 [n_hrf][K], cost_dev float64 [n_hrf][V].  The caller
 * sums over voxels (and all-reduces over ranks) for the shared-HRF step.
 */
int pb_hrf_cost(const double* z_dev, int64_t ldz, const float* y_dev, int64_t ldy,
                int V, int N, const double* taps_dev, int K, int n_hrf,
                double* cost_dev, void* stream);

/*
 * Spectral radius of H^T H for H = toeplitz(taps, N, N) . cumsum by the power
 * iteration of spectral_radius_est (pybold/utils.py:94-109), fused in one launch:
 * x0_dev float64 [N] start vector (the caller draws it, as the reference does with
 * np.random.randn), out_dev float64 [2] = { ||x_new||, iterations done }.
 */
int pb_spectral_radius(const double* x0_dev, int N, const double* taps_dev, int K,
                       int nb_iter, double tol, double* out_dev, void* stream);

/*
 * Per-voxel HRFs (blind deconvolution with one HRF dilation per voxel, the loop
 * the reference fans out over voxels: pybold/bold_signal.py:281-382,
 * examples/icassp_2019/simulation.py:62-72).
 *
 * pb_fista_solve_pp   the solver of pb_fista_solve with per-problem taps
 *                     taps_dev float64 [P][ldt] (K used) and per-problem step
 *                     step_dev float64 [P]; stop rule NONE or LOOPS; no cost trace.
 *                     ldt = 0: ONE HRF (taps_dev [K]) and ONE step (step_dev [1]) in device
 *                     memory shared by every problem -- the shared-HRF blind step, whose
 *                     taps come out of pb_theta_fit without passing through the host.  With
 *                     n_done_dev, no stop rule and K <= 48 that form runs on the matrix-pipe
 *                     kernels (129 .. 1 280 scans: one wave, two or four waves per 16 problems),
 *                     what they hand back on the vector forms; per-problem HRFs: vector forms.
 * pb_hrf_cost_pv      pb_hrf_cost with one HRF per (candidate, voxel):
 *                     taps_dev float64 [n_hrf][V][K], cost_dev float64 [n_hrf][V].
 * pb_gram_frobenius   out[p] = || A_p^T A_p ||_F with A_p = toeplitz(taps_p, N, N) tril(1):
 *                     the step constant of _loops_deconv (pybold/bold_signal.py:249-254)
 *                     for P different HRFs at once.
 */
int pb_fista_solve_pp(const float* y_dev, int64_t ldy, double* w_dev, int64_t ldw, int P, int N,
                      const double* taps_dev, int64_t ldt, int K, const double* step_dev,
                      double lbda, const double* lbda_dev, const double* betas_dev, int n_iter,
                      int stop_mode, double tol, int32_t* n_done_dev, unsigned flags, void* stream);
int pb_hrf_cost_pv(const double* z_dev, int64_t ldz, const float* y_dev, int64_t ldy,
                   int V, int N, const double* taps_dev, int K, int n_hrf,
                   double* cost_dev, void* stream);
int pb_gram_frobenius(const double* taps_dev, int64_t ldt, int P, int K, int N,
                      double* out_dev, void* stream);
/* pb_fista_outputs with one HRF per row: taps_dev float64 [P][ldt]. */
int pb_fista_outputs_pp(const double* w_dev, int64_t ldw, int P, int N,
                        const double* taps_dev, int64_t ldt, int K,
                        double* z_dev, int64_t ldz, double* x_dev, int64_t ldx, void* stream);
/*
 * Un-normalised two-gamma SPM HRF (pybold/hrf_model.py:25-31) for M dilations at
 * the K sample times t_dev (seconds, already decimated): out_dev float64 [M][K] =
 * gamma_pdf(d t; a_peak, loc_peak) - ratio * gamma_pdf(d t; a_under, loc_under).
 */
int pb_spm_hrf(const double* deltas_dev, int M, const double* t_dev, int K,
               double a_peak, double loc_peak, double a_under, double loc_under,
               double ratio, double* out_dev, void* stream);

/*
 * Float64-`y` forms (suffix _d).  pb_fista_solve stores y as float32 in HBM (the batch
 * layout); the entry points below take y as float64 and compute in float64 end to end --
 * the reference's own arithmetic -- for the single-voxel / small-batch calls of the
 * reference's API (deconv, _loops_deconv, bd, hrf_fit_err on 1-D arrays) and for the
 * theta-step, whose finite-difference L-BFGS-B is sensitive to the last digits of the cost
 * (pybold/bold_signal.py:217-222, :329-333).
 *
 * pb_fista_solve_d    pb_fista_solve in float64 end to end: y float64, cost trace J float64
 *                     [P][ldj].  The only entry point that takes NEGATIVE lambdas: the reference's
 *                     noise-driven search (pybold/bold_signal.py:141-145) drives alpha, hence
 *                     lambda = 1 / (2 alpha), below zero, and `sign(u) max(|u| - th, 0)` (:66) then
 *                     GROWS every non-zero entry by |th|; these kernels restate that expression,
 *                     the float32-y kernels clamp (pb_fista_solve rejects a negative scalar lbda).  Register-resident kernel (one problem per wave,
 *                     fista_exact_kernel) for series of up to 640 scans with HRFs of up to
 *                     32 taps when taps_host is given (window rule: wind = 6); otherwise, or
 *                     with PB_FLAG_FORCE_GENERIC, the any-size LDS kernel, which reads the taps
 *                     from taps_dev.  Either taps pointer may be NULL if the other kernel runs.
 * pb_fista_stats_d, pb_hrf_cost_d, pb_hrf_cost_pv_d   as their float32-y namesakes.
 */
int pb_fista_solve_d(const double* y_dev, int64_t ldy, int y_rep, double* w_dev, int64_t ldw,
                     int P, int N, const double* taps_host, const double* taps_dev, int K,
                     double step, double lbda, const double* lbda_dev, const double* betas_dev,
                     int n_iter, double* J_dev, int64_t ldj, int stop_mode, double tol, int wind,
                     int32_t* n_done_dev, unsigned flags, void* stream);
/*
 * OPT-IN EXTRA, never part of a parity run: the recurrence of pb_fista_solve_d with a BACKTRACKED step.  The reference
 * has a constant step only (pybold/bold_signal.py:52-53 / :253-254: 1 / (0.9 rho) or 1 / ||A^T A||_F); this is
 * BASELINE's "Lipschitz-backtracked step": per iteration, with g = H^T (H w - y) at the extrapolated point w,
 *     repeat   u = w - s g ;  p = soft(u, lbda s) ;
 *              accept if  F(p) <= F(w) + <p - w, g> + ||p - w||^2 / (2 s)     (F = 0.5 ||H . - y||^2)
 *              else s <- eta s                       (at most max_halvings_per_iter times per iteration; s never grows)
 *     w <- p + beta_k (p - u)                        (the reference's momentum, on the accepted gradient point)
 * so a caller without a Lipschitz constant starts from any step0; with step0 <= 1 / L the test passes at once and the
 * iterates are those of the constant-step solver.  float64 end to end (y float64), one workgroup per problem, any
 * 5 N + K <= 20000.  step_out_dev (float64 [P]) / halvings_out_dev (int32 [P]): the final step and the number of
 * reductions of every problem, or NULL.  Checked against its own NumPy statement (oracle: fista_backtrack_batch).
 */
int pb_fista_solve_backtrack_d(const double* y_dev, int64_t ldy, int y_rep, double* w_dev, int64_t ldw, int P, int N,
                               const double* taps_dev, int K, double step0, double eta, int max_halvings_per_iter,
                               double lbda, const double* lbda_dev, const double* betas_dev, int n_iter,
                               int32_t* n_done_dev, double* step_out_dev, int32_t* halvings_out_dev,
                               unsigned flags, void* stream);
int pb_fista_stats_d(const double* w_dev, int64_t ldw, const double* y_dev, int64_t ldy,
                     int y_rep, int P, int N, const double* taps_dev, int K,
                     double* r2_dev, double* l1_dev, void* stream);
int pb_hrf_cost_d(const double* z_dev, int64_t ldz, const double* y_dev, int64_t ldy,
                  int V, int N, const double* taps_dev, int K, int n_hrf,
                  double* cost_dev, void* stream);
int pb_hrf_cost_pv_d(const double* z_dev, int64_t ldz, const double* y_dev, int64_t ldy,
                     int V, int N, const double* taps_dev, int K, int n_hrf,
                     double* cost_dev, void* stream);

/*
 * out[v] = || H^T y_v ||_inf with H = toeplitz(taps, N, N) . cumsum (ConvAndLinear.adj,
 * pybold/linear.py:95-113, followed by max|.|): the smallest lambda whose solution is
 * diff_z = 0, i.e. the top of a per-voxel regularisation path `lambda = c * lambda_max,v`.
 * The reference hard-codes its lambda lists (examples/icassp_2019/simulation.py:113-114).
 */
int pb_lambda_max(const float* y_dev, int64_t ldy, int V, int N, const double* taps_dev, int K,
                  double* out_dev, void* stream);
int pb_lambda_max_d(const double* y_dev, int64_t ldy, int V, int N, const double* taps_dev, int K,
                    double* out_dev, void* stream);

/*
 * Row-wise inf-norm normalisation out = x / (max|x| + 1e-12) (inf_norm,
 * pybold/utils.py:112-138): V rows of n elements (n of any size; a 1-D or 3-D array of the
 * reference is one row of all its elements).  NaN propagates like np.max.
 */
int pb_inf_norm(const double* x_dev, int64_t ldx, double* out_dev, int64_t ldo, int V, int64_t n,
                void* stream);

/*
 * Theta-step of the blind solver without per-candidate passes over the data.
 * hrf_fit_err(theta) = 0.5||y - h(theta) * z||^2 (pybold/bold_signal.py:217-222) is the
 * quadratic form 0.5 yy - h^T b + 0.5 h^T G h in the K taps with
 *   G[m][m'] = sum_i z[i-m] z[i-m'],  b[m] = sum_i z[i-m] y[i],  yy = sum_i y[i]^2.
 *
 * pb_hrf_normal_eq   one pass over (z, y): a set = G row-major [K][K], b [K], yy
 *                    (pb_hrf_normal_eq_len(K) = K*K+K+1 float64).
 *                    per_voxel = 0: out_dev [len] = sum over the V voxels (shared-HRF
 *                      variant, BASELINE config 4; an empty shard V = 0 writes zeros so the
 *                      rank still contributes to the all-reduce); work_dev = scratch of
 *                      work_len float64 (>= len; more, up to 2048*len, = more workgroups);
 *                      the sum order is fixed: results are reproducible.
 *                    per_voxel = 1: out_dev [V][len], one set per voxel; work_dev unused.
 * pb_theta_fit       argmin over theta in [lo, hi] of the quadratic form for M sets
 *                    (ne_dev [M][ldne]) with h(theta) the un-normalised two-gamma SPM HRF
 *                    at the K sample times t_dev (parameters as pb_spm_hrf): section search
 *                    -- one scan of 64 candidates over [lo, hi], then, per further level
 *                    of n_refine, two scans of 16 candidates around the best one (bracket
 *                    / 56 per level) -- closed by a parabola vertex (n_refine = 3: about
 *                    1e-9 on theta); replaces the reference's fmin_l_bfgs_b call
 *                    (:329-333).  theta_dev [M], cost_dev [M] = F(theta*), taps_dev
 *                    [M][ldt] = h(theta*) (may be NULL).  Entirely on the device: the
 *                    next z-step can read taps_dev through pb_fista_solve_pp (ldt = 0).
 */
int64_t pb_hrf_normal_eq_len(int K);
int pb_hrf_normal_eq(const double* z_dev, int64_t ldz, const float* y_dev, int64_t ldy,
                     int V, int N, int K, int per_voxel, double* work_dev, int64_t work_len,
                     double* out_dev, void* stream);
int pb_hrf_normal_eq_d(const double* z_dev, int64_t ldz, const double* y_dev, int64_t ldy,
                       int V, int N, int K, int per_voxel, double* work_dev, int64_t work_len,
                       double* out_dev, void* stream);
int pb_theta_fit(const double* ne_dev, int64_t ldne, int M, int K, const double* t_dev,
                 double a_peak, double loc_peak, double a_under, double loc_under, double ratio,
                 double lo, double hi, int n_refine, double* theta_dev, double* cost_dev,
                 double* taps_dev, int64_t ldt, void* stream);

/*
 * The two launches of an outer iteration of the shared-HRF blind loop beside its z-step
 * (bd, pybold/bold_signal.py:320-342, with ONE dilation for all voxels: BASELINE config 4):
 *
 * pb_hrf_normal_eq_w   pb_hrf_normal_eq (shared form, float32 y) straight from the INNOVATION
 *                      w = diff_z: the cumulative sum z = cumsum(w) (:326) is taken inside the pass
 *                      (same summation tree as pb_integ_op: bit-identical sets) and ||w||_1 (the
 *                      regularisation term of the cost, :337-342) is summed in the same pass.
 *                      out_dev [len + 1] = the normal equations followed by sum_v ||w_v||_1: the
 *                      message of the outer iteration's one all-reduce.  work_dev >= len + 1.
 * pb_theta_fit_step    pb_theta_fit on ONE such message (M = 1) that also leaves what the next
 *                      z-step and the cost trace need on the device: step_dev [1] =
 *                      1 / ||A^T A||_F for h(theta*) on N scans (:249-254, pb_gram_frobenius's
 *                      closed form) and jcost_dev [1] = (2 F(theta*) + lbda ||w||_1) / ||y||^2.
 */
int pb_hrf_normal_eq_w(const double* w_dev, int64_t ldw, const float* y_dev, int64_t ldy,
                       int V, int N, int K, double* work_dev, int64_t work_len, double* out_dev,
                       void* stream);
int pb_theta_fit_step(const double* msg_dev, int K, const double* t_dev,
                      double a_peak, double loc_peak, double a_under, double loc_under, double ratio,
                      double lo, double hi, int n_refine, int N, double lbda, double* theta_dev,
                      double* cost_dev, double* taps_dev, double* step_dev, double* jcost_dev,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYBOLD_HIP_H */
