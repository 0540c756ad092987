// tuning.h -- the kernels' tuning constants.
//
// Shipped build: every knob below is a COMPILE-TIME constant (the value its sweep settled on; the measurements sit next to
// the code that reads it) and dqq_set_option() knows none of them: a call's route is a function of its arguments.
// Developer build (-DDQQ_TUNING: `DQQ_EXTRA_FLAGS=-DDQQ_TUNING python -m diffqcqp_amd.build`): the same names are
// process-wide atomics behind dqq_set_option / dqq_get_option, for A/B sweeps (tools/) and for the tests that drive the
// alternative kernels (tests/: `needs_tuning`).  Nothing here changes a result: the knobs that DO change numerics --
// reference-order kernels instead of the matrix-core ones for 16 < N <= 64 -- are a per-call flag of the C ABI
// (DQQ_F_REFERENCE_ORDER, include/diffqcqp_hip.h), not process state.
#pragma once

#include <atomic>

namespace dqq {

#if defined(DQQ_TUNING)
constexpr bool kTuning = true;
#define DQQ_KNOB(name, dflt)              \
    extern std::atomic<int> g_##name;     \
    inline int knob_##name() { return g_##name.load(std::memory_order_relaxed); }
#else
constexpr bool kTuning = false;
#define DQQ_KNOB(name, dflt) \
    constexpr int knob_##name() { return (dflt); }
#endif

// name, shipped value                 what it selects (0 / -1 = the built-in choice where noted)
DQQ_KNOB(fwd_lpp, 0)                // lanes per problem of the diagonal forward (0 = from (N, B): fwd_diag_default_lpp)
DQQ_KNOB(wpb, 0)                    // waves per workgroup of the diagonal kernels (0 = 4)
DQQ_KNOB(fuse_fallback, -1)         // non-diagonal tiles inside the fast kernel (1), queued (0), by (N, B) (-1)
DQQ_KNOB(fwd_respread, 16)          // N = 8 forward on two lanes: tail of <= this many problems moves to four lanes
DQQ_KNOB(fwd_respread2, 8)          // ... and of <= this many to eight lanes
DQQ_KNOB(fwd_respread2_from, 48)    // ... but not before this iteration: a tile whose problems are done by then (the
                                    // well-conditioned bench shape: tile maxima 24-38) skips the second move, which costs it
                                    // more than its last few iterations on four lanes (QP forward 26.0 -> 25.4 us, round 6)
DQQ_KNOB(lane_dense, 1)             // general forward N <= 8: lane-per-problem kernel
DQQ_KNOB(lane_defer, 0)             // general forward N <= 16: refactorisation every k trips (0 = 4 QCQP / 6 others)
DQQ_KNOB(dense_teams, 1)            // general backward: 64/T problems per wave for small N
DQQ_KNOB(small_fwd, 1)              // general forward N = 10..16: team-per-problem kernel
DQQ_KNOB(small_bwd, 1)              // general backward even N <= 16: statically sized team kernel
DQQ_KNOB(lane_bwd, 1)               // general backward N <= 8, B >= 16384: lane-per-problem kernel
DQQ_KNOB(fwd_feedback, 1)           // 0: the forward ignores DQQ_F_EXPECT_DENSE
DQQ_KNOB(bwd_skip_classify, 1)      // 0: the backward ignores DQQ_F_EXPECT_DENSE (always classifies first)
#undef DQQ_KNOB

// Route counters (diagnostics, both builds): how often the feedback hint changed a route.  Read with dqq_get_option, reset by
// dqq_set_option(name, 0) -- the only names the shipped dqq_set_option accepts.
extern std::atomic<int> g_lane_list_drains;    // drain launches sent to the lane-per-problem backward
extern std::atomic<int> g_bwd_whole_batches;   // DQQ_P_AUTO backwards solved whole by the lane-per-problem kernel
extern std::atomic<int> g_fwd_feedback_routes; // N = 8 forwards moved to one lane per problem

} // namespace dqq
