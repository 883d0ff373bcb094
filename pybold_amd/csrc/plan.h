// Launch plans of a solve: how P problems are laid over the kernel forms (whole rounds of waves on the densest form,
// the remainder on whichever finishes it first, part of it beside the main launch on a side stream).  The functions
// are pure integer / double arithmetic and compile for the HOST (pb_fista_plan_ex, the plans of calls whose problem
// counts the host knows) and for the DEVICE (round 5: a call that is partitioned on the device -- dense class on the
// matrix pipe, sparse class on the vector forms, re-solves of what a guard handed back -- knows its list lengths only
// there; `plan_kernel` in path.h runs the same functions on them, so a list of n problems gets exactly the plan a
// host-planned call of n problems gets).
//
// Measurements behind the constants: DESIGN.md 5.1d (tools/ab_forms.py, profiles/r2_ab_forms.txt,
// profiles/r4_split_form_passes.txt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pb {

#define PB_HD __host__ __device__ inline

enum Form { FORM_GENERIC = 0, FORM_FAST1 = 1, FORM_PAIR = 2, FORM_WIDE = 3, FORM_MFMA = 4, FORM_MFMA2 = 5, FORM_MFMA4 = 6 };

// ---- dispatch of a plain solve (no stop rule) over the register-resident vector forms --------
// All forms keep two waves per SIMD and are VALU-issue bound, so a launch costs "rounds":
// waves / (CUs x 4 SIMDs x 2), a last partial round at most half full costing ~0.56 of a
// round (its waves run alone on their SIMDs), a fuller one a whole round, plus a fixed
// ~0.03-0.05 for launch, prologue and epilogue.  Unit = one full round of the pair kernel
// (16 384 problems, ~2.73 ms for 500 iterations of N = 300, K = 30 on MI355X).  Measured
// (tools/ab_forms.py, profiles/r2_ab_forms.txt):
//   pair   8 problems per wave   1.00 per round (2-parallel fast FIRs)
//   fast1  4 problems per wave   0.63 per round of half as many problems
//   wide   1 problem  per wave   0.19 per round of an eighth as many (short series only)
// A problem count that is not a whole number of rounds is therefore split: whole rounds on
// the pair (or single-row) kernel, the remainder on whichever form finishes it first, as a
// second launch on the same stream (the first launch ends with every SIMD draining at once,
// so running the remainder after it costs what overlapping would) -- or, when the remainder
// exceeds half a round of pair waves, as a concurrent group (plan_pieces below).
constexpr double COST_FAST1 = 0.63, COST_WIDE = 0.19, COST_PARTIAL = 0.56;
constexpr double COST_LAUNCH = 0.03, COST_LAUNCH_WIDE = 0.05;

PB_HD int problems_per_wave(int form) { return form == FORM_PAIR ? 8 : (form == FORM_FAST1 ? 4 : 1); }

// `slots` = waves the device holds at two per SIMD (CUs x 4 x 2: 2 048 on MI355X)
PB_HD double form_cost(int form, int P, double slots) {
  const double unit = form == FORM_PAIR ? 1.0 : (form == FORM_FAST1 ? COST_FAST1 : COST_WIDE);
  const int ppw = problems_per_wave(form);
  const double r = (double)((P + ppw - 1) / ppw) / slots;
  const double whole = floor(r), part = r - whole;
  return (form == FORM_WIDE ? COST_LAUNCH_WIDE : COST_LAUNCH) +
         unit * (whole + (part == 0.0 ? 0.0 : (part <= 0.5 ? COST_PARTIAL : 1.0)));
}

// cheapest single form for P problems among those available
PB_HD int best_form(int P, bool has_pair, bool has_wide, double slots, double* cost = nullptr) {
  int best = FORM_FAST1;
  double c = form_cost(FORM_FAST1, P, slots);
  if (has_pair && P >= 2 && form_cost(FORM_PAIR, P, slots) < c) { best = FORM_PAIR; c = form_cost(FORM_PAIR, P, slots); }
  if (has_wide && form_cost(FORM_WIDE, P, slots) < c) { best = FORM_WIDE; c = form_cost(FORM_WIDE, P, slots); }
  if (cost) *cost = c;
  return best;
}

struct Plan {          // problems [0, n_main) on `main_form`, [n_main, P) on `tail_form`
  int n_main;
  int main_form;
  int tail_form;
};

PB_HD Plan plan_plain(int P, bool has_pair, bool has_wide, bool one_launch, double slots) {
  double c_best;
  Plan best{0, FORM_GENERIC, best_form(P, has_pair, has_wide, slots, &c_best)};
  if (one_launch) return best;
  for (int mi = 0; mi < 2; ++mi) {
    const int main_form = mi == 0 ? FORM_PAIR : FORM_FAST1;
    if (main_form == FORM_PAIR && !has_pair) continue;
    // main launch = a whole number of rounds, or of half rounds (every wave alone on its SIMD)
    const int half = (int)slots * problems_per_wave(main_form) / 2;
    for (int ui = 0; ui < 2; ++ui) {
      const int unit = ui == 0 ? 2 * half : half;
      const int n_main = (P / unit) * unit;
      if (n_main == 0 || n_main == P) continue;
      double c_tail;
      const int tail = best_form(P - n_main, has_pair, has_wide, slots, &c_tail);
      const double c = form_cost(main_form, n_main, slots) + c_tail;
      if (c < c_best) {
        c_best = c;
        best = Plan{n_main, main_form, tail};
      }
    }
  }
  return best;
}

struct Piece {
  int form, p0, p1;
  bool side;                             // runs on the side stream ...
  bool group;                            // ... as part of the concurrent group that closes the plan
};
constexpr int MAX_PIECES = 8;

// Plan of a plain solve as up to four pieces.  Sequential part: plan_plain.  If what remains
// after the whole rounds of pair waves lies between one and two half rounds, the whole rounds
// go first (one launch) and the rest becomes a concurrent group: half a round of pair waves
// with the remainder beside it on the side stream (forked after the whole rounds).
PB_HD int plan_pieces(int P, bool has_pair, bool has_wide, bool one_launch, bool one_stream, double slots, Piece* out) {
  int n = 0;
  const int half = (int)slots * 4;                 // problems in half a round of pair waves
  const int round = 2 * half;
  const int whole = (P / round) * round;
  const int R = P - whole;                         // what the whole rounds leave
  if (has_pair && !one_launch && !one_stream && R > half) {
    const int rest = R - half;                     // < half
    const int quarter = half / 2;                  // single-row waves: one per SIMD
    const int wide_round = (int)slots / 2;         // one-problem waves: one per SIMD
    const int g0 = whole, g1 = whole + half;
    int m = 0;
    Piece grp[3];
    if (has_wide && rest < 2 * wide_round) {       // at most two one-problem waves per SIMD
      grp[m++] = Piece{FORM_PAIR, g0, g1, false, true};
      grp[m++] = Piece{FORM_WIDE, g1, P, true, true};
    } else if (rest <= quarter) {                  // at most one single-row wave per SIMD
      grp[m++] = Piece{FORM_PAIR, g0, g1, false, true};
      grp[m++] = Piece{FORM_FAST1, g1, P, true, true};
    } else if (has_wide && rest - quarter <= 2 * wide_round) {
      // (both on the one side stream, in a fixed order.  On a stream of their own the left-overs
      // land wherever the dispatcher happens to put them: 2.46-2.78 ms, tools/conc_probe4.py)
      // The left-overs go FIRST on the side stream: they are latency-bound (0.37 ms whatever
      // their number) and so are the pair waves while alone on their SIMDs, so the two overlap
      // for free; behind the single-row waves they would run alone at the very end
      // (tools/conc_probe5.py: 2.33 ms instead of 2.44 for 12 500 problems).
      grp[m++] = Piece{FORM_PAIR, g0, g1, false, true};
      grp[m++] = Piece{FORM_WIDE, g1 + quarter, P, true, true};
      grp[m++] = Piece{FORM_FAST1, g1, g1 + quarter, true, true};
    }
    if (m > 0) {
      if (whole > 0) out[n++] = Piece{FORM_PAIR, 0, whole, false, false};
      for (int i = 0; i < m; ++i) out[n++] = grp[i];
      return n;
    }
  }
  // Between a quarter and three eighths of a round (4 096 < R <= 6 144): one single-row wave per
  // SIMD with up to two one-problem waves beside it, instead of pair waves alone on their SIMDs
  // (tools/conc_probe7.py: 5 000 problems 1.14 ms against 1.31, 6 000 1.39 against 1.59).
  if (has_pair && has_wide && !one_launch && !one_stream) {
    const int quarter = half / 2, wide_round = (int)slots / 2;
    // (behind whole rounds only with ONE left-over wave per SIMD: two measured no gain there)
    if (R > quarter && R - quarter <= (whole > 0 ? 1 : 2) * wide_round) {
      if (whole > 0) out[n++] = Piece{FORM_PAIR, 0, whole, false, false};
      out[n++] = Piece{FORM_FAST1, whole, whole + quarter, false, true};
      out[n++] = Piece{FORM_WIDE, whole + quarter, P, true, true};
      return n;
    }
  }
  const Plan pl = plan_plain(P, has_pair, has_wide, one_launch, slots);
  if (pl.n_main > 0) out[n++] = Piece{pl.main_form, 0, pl.n_main, false, false};
  out[n++] = Piece{pl.tail_form, pl.n_main, P, false, false};
  return n;
}

// Plan with the matrix-pipe form as the main form (plain solves, cost trace or not): one wave per
// SIMD carries 16 problems, so a round is the same 16 384 problems as a round of pair waves but takes
// ~0.68 of its time (measured, tools/r3_mfma_probe.py), and -- the waves being alone on their SIMDs --
// ANY remainder launched on it costs a full round.  Whole rounds therefore go to the matrix pipe,
// a remainder above half a round too; a smaller one keeps the plan of the vector forms
// (plan_pieces: pair waves alone on their SIMDs, single-row and one-problem waves beside them).
// A pass of the split form (fista_mfma2.h: 16 problems on TWO SIMDs, 8 192 problems per pass) lasts about 0.55 of a
// one-wave pass (measured: profiles/r4_split_form_passes.txt).  With it (`has_mfma2`: plain solves without cost trace)
// what the whole rounds leave is closed as
//   R <= MFMA2_MIN_R                     the vector plan (latency-bound forms finish a few thousand problems sooner)
//   MFMA2_MIN_R < R <= half a round      one pass of the split form
//   half < R <= half + one-problem waves half a round on the split form, then the left-overs one problem per wave
//   beyond                               one more pass of the one-wave form, as before
constexpr int MFMA2_MIN_R = 4608;
constexpr int MFMA2_BESIDE_CHUNKS = 2;         // chunks of one one-problem wave per SIMD beside a split-form pass

PB_HD int plan_pieces_mfma(int P, bool has_pair, bool has_wide, bool one_launch, bool one_stream, bool has_mfma2,
                           int beside_chunks, double slots, Piece* out) {
  const int round = (int)slots * 8;                  // 16 problems x (slots / 2) waves
  const int whole = (P / round) * round;
  const int R = P - whole;
  int n = 0;
  if (has_mfma2 && !one_launch && R > 0) {
    const int half = round / 2, wide_max = (int)slots;
    if (R > MFMA2_MIN_R && R <= half) {
      if (whole > 0) out[n++] = Piece{FORM_MFMA, 0, whole, false, false};
      out[n++] = Piece{FORM_MFMA2, whole, P, false, false};
      return n;
    }
    if (R > half && has_wide && R - half <= beside_chunks * (wide_max / 2)) {
      // The left-overs as one-problem waves BESIDE the split-form pass, on the side stream: a wave of the N <= 320
      // split form holds 355 registers, ONE 88-register one-problem wave fits next to it on a SIMD and issues in the
      // gaps the pass's barriers leave.  One per SIMD and no more: with two, 355 + 2 x 88 registers no longer fit and
      // the waves that wait block the placement of the two-wave workgroups (measured: 1 808 left-overs in one launch
      // beside the pass 2.25 ms, one after the other 1.62 ms; 808 beside it 1.15 ms) -- so they go in chunks of one
      // per SIMD, one chunk after the other on the side stream (a chunk beside the pass lasts about twice what it lasts
      // alone: 10 000 problems 1.52 ms, 9 000 1.14 ms; three chunks, 11 000 problems, 1.87 ms: slower than one pass of
      // the one-wave form, so two at most).
      if (whole > 0) out[n++] = Piece{FORM_MFMA, 0, whole, false, false};
      out[n++] = Piece{FORM_MFMA2, whole, whole + half, false, !one_stream};
      if (one_stream) {
        out[n++] = Piece{FORM_WIDE, whole + half, P, false, false};
      } else {
        for (int c0 = whole + half; c0 < P; c0 += wide_max / 2)
          out[n++] = Piece{FORM_WIDE, c0, c0 + wide_max / 2 < P ? c0 + wide_max / 2 : P, true, true};
      }
      return n;
    }
  }
  // a remainder costs one matrix-pipe pass (1.74 ms per 500 iterations at N = 300) whatever its size; the
  // vector plan closes up to half a round of pair waves + one one-problem wave per SIMD beside them in
  // 1.55 ms, so it keeps remainders up to round/2 + round/16
  const bool has_side = has_pair && has_wide && !one_stream;
  if (one_launch || R == 0 || R > round / 2 + (has_side ? round / 16 : 0)) {
    out[n++] = Piece{FORM_MFMA, 0, P, false, false};
    return n;
  }
  // (the remainder as one-problem waves on the side stream from the START -- an 84-register wave
  // fits beside a matrix-pipe wave's 416 -- measured slower: 12.67 against 11.88 ms per step of
  // config 3, profiles/r3_remainder_beside_mfma_ab.txt: it goes behind the whole rounds)
  if (whole > 0) out[n++] = Piece{FORM_MFMA, 0, whole, false, false};
  Piece sub[4];
  const int m = plan_pieces(R, has_pair, has_wide, false, one_stream, slots, sub);
  for (int i = 0; i < m; ++i) out[n++] = Piece{sub[i].form, sub[i].p0 + whole, sub[i].p1 + whole, sub[i].side, sub[i].group};
  return n;
}

// ---- the plan of a PARTITIONED call (round 5): n_d dense problems at positions [0, n_d) of the list array, the sparse
// ones behind them.  No sparse problem: the plan of a dense call (plan_pieces_mfma).  Otherwise the matrix-pipe form takes
// the whole rounds of the dense class (and what is left of it, if that is worth a pass of its own: above half a round on
// the one-wave form, above MFMA2_MIN_R on the split form), and whatever it does not take joins the sparse class in ONE
// list for the vector forms, positions [covered, P), planned like a vector call of that many problems.
PB_HD int plan_partitioned(int n_d, int P, bool has_pair, bool has_wide, bool one_stream, bool has_mfma2, int beside_chunks,
                           double slots, Piece* out) {
  if (n_d >= P) return plan_pieces_mfma(P, has_pair, has_wide, false, one_stream, has_mfma2, beside_chunks, slots, out);
  int n = 0;
  const int round = (int)slots * 8;
  const int whole = (n_d / round) * round, R = n_d - whole;
  int covered = whole;
  if (R > round / 2) {
    covered = n_d;
    out[n++] = Piece{FORM_MFMA, 0, n_d, false, false};
  } else {
    if (whole > 0) out[n++] = Piece{FORM_MFMA, 0, whole, false, false};
    if (has_mfma2 && R > MFMA2_MIN_R) {
      out[n++] = Piece{FORM_MFMA2, whole, n_d, false, false};
      covered = n_d;
    }
  }
  Piece sub[4];
  const int m = plan_pieces(P - covered, has_pair, has_wide, false, one_stream, slots, sub);
  for (int i = 0; i < m; ++i) out[n++] = Piece{sub[i].form, sub[i].p0 + covered, sub[i].p1 + covered, sub[i].side, sub[i].group};
  return n;
}

// ---- device-side plans (round 5) -----------------------------------------------------------------------------
// A list whose length n is known on the device only (a class of the partition, the problems a guard handed back) is
// solved by a STATIC sequence of candidate launches, each with a worst-case grid, each reading the slots [s0, s1) it
// solves from device memory (FistaArgs::range): the waves of an empty candidate leave at once.  `plan_to_candidates`
// lays the pieces of a plan over the candidates:
//   caller's stream, before the fork:  MFMA (whole rounds), PAIR0, FAST0 (the first piece of that form outside a group)
//   caller's stream, after the fork:   MFMA2, PAIR1, FAST1 (a group's piece, or a second piece of the form), WIDE
//   side stream (forked there):        SIDE_WIDE0, SIDE_WIDE1, SIDE_FAST            -- then the join
// which keeps the order the host-planned calls launch in (whole rounds, fork, the group with its side pieces).
enum Cand {
  CAND_MFMA = 0, CAND_PAIR0, CAND_FAST0, CAND_MFMA2, CAND_PAIR1, CAND_FAST1, CAND_WIDE,
  CAND_SIDE_WIDE0, CAND_SIDE_WIDE1, CAND_SIDE_FAST, CAND_COUNT
};
constexpr int CAND_FIRST_AFTER_FORK = CAND_MFMA2, CAND_FIRST_SIDE = CAND_SIDE_WIDE0;
PB_HD int cand_form(int c) {
  return c == CAND_MFMA ? FORM_MFMA : c == CAND_MFMA2 ? FORM_MFMA2 : (c == CAND_PAIR0 || c == CAND_PAIR1) ? FORM_PAIR
         : (c == CAND_FAST0 || c == CAND_FAST1 || c == CAND_SIDE_FAST) ? FORM_FAST1 : FORM_WIDE;
}
PB_HD bool cand_side(int c) { return c >= CAND_FIRST_SIDE; }

// the most slots candidate c can be given in a plan of at most n_max problems: what the host sizes its grid for
// (tests/test_host_logic.py checks every plan against these bounds)
PB_HD int cand_max_slots(int c, int n_max, double slots) {
  const int round = (int)slots * 8;                  // 16 384 problems on MI355X
  const int lim = c == CAND_MFMA || c == CAND_PAIR0 || c == CAND_FAST0 ? n_max     // whole rounds, or a whole list
                  : c == CAND_MFMA2 ? round / 2                                     // one pass of the split form
                  : c == CAND_PAIR1 || c == CAND_FAST1 || c == CAND_WIDE ? round    // what whole rounds leave
                  : c == CAND_SIDE_FAST ? round / 4                                 // one single-row wave per SIMD
                  : (int)slots;                                                     // side WIDE: two one-problem waves per SIMD
  return lim < n_max ? lim : n_max;
}

// ranges[2 c], ranges[2 c + 1] = [s0, s1) of candidate c (empty: 0, 0).  Returns 0, or -1 when a piece found its
// candidate taken (cannot happen for the plans above: tests/test_host_logic.py walks every list length through the
// host build of this function and checks that the candidates tile [0, n) exactly).
PB_HD int plan_to_candidates(const Piece* pc, int npc, int32_t* ranges) {
  for (int c = 0; c < 2 * CAND_COUNT; ++c) ranges[c] = 0;
  int rc = 0;
  for (int i = 0; i < npc; ++i) {
    if (pc[i].p1 <= pc[i].p0) continue;
    const auto taken = [&](int c) { return ranges[2 * c + 1] > ranges[2 * c]; };
    int c = -1;
    switch (pc[i].form) {
      case FORM_MFMA:  c = pc[i].side ? -1 : CAND_MFMA; break;
      case FORM_MFMA2: c = pc[i].side ? -1 : CAND_MFMA2; break;
      case FORM_PAIR:  c = pc[i].side ? -1 : ((pc[i].group || taken(CAND_PAIR0)) ? CAND_PAIR1 : CAND_PAIR0); break;
      case FORM_FAST1: c = pc[i].side ? CAND_SIDE_FAST : ((pc[i].group || taken(CAND_FAST0)) ? CAND_FAST1 : CAND_FAST0); break;
      case FORM_WIDE:  c = pc[i].side ? (taken(CAND_SIDE_WIDE0) ? CAND_SIDE_WIDE1 : CAND_SIDE_WIDE0) : CAND_WIDE; break;
      default: break;
    }
    if (c < 0) { rc = -1; continue; }
    // a second piece of a form that continues the first one on the same stream and outside a group: one launch
    if (!pc[i].side && !pc[i].group && (c == CAND_PAIR1 || c == CAND_FAST1)) {
      const int c0 = c == CAND_PAIR1 ? CAND_PAIR0 : CAND_FAST0;
      if (taken(c0) && ranges[2 * c0 + 1] == pc[i].p0) { ranges[2 * c0 + 1] = pc[i].p1; continue; }
    }
    if (taken(c)) { rc = -1; continue; }
    ranges[2 * c] = pc[i].p0;
    ranges[2 * c + 1] = pc[i].p1;
  }
  return rc;
}

}  // namespace pb
