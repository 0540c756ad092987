/*
 * diffqcqp_hip.h -- C ABI of libdiffqcqp_hip.so: batched ADMM QP / QCQP solve
 * and implicit-function backward on MI355X (gfx950), float64.
 *
 * This is the drop-in boundary for the batched path of quentinll/diffqcqp.
 * The reference crosses Python -> C++ once PER PROBLEM through the pybind11
 * module `diffqcqp` (reference pybindings.cpp:74-83) from the batch loops of
 * qcqp.py:29-31, 45-47, 149-151, 167-172.  Each entry point below replaces one
 * of those loops (loop + per-problem call + the torch.bmm gradient assembly
 * that follows it) with one asynchronous launch over the whole batch.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into memory the caller owns (borrowed
 *     for the duration of the stream-ordered work); tensors are contiguous and
 *     batch-major exactly as torch lays them out:
 *         P (B,N,N)   q, x, grad_x, grad_q (B,N,1)   l_n, mu, grad_l_n, grad_mu (B,N/2,1)
 *         l_min, l_max, v, grad_l_min, grad_l_max (B,N,1)
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream) and the call returns without synchronising;
 *   - return value: 0 on success, <0 for an invalid argument (DQQ_E_*), >0 a
 *     hipError_t from the launch (the launch's own status: the calling thread's
 *     sticky HIP error is neither consumed nor cleared).  Nothing throws.  The library
 *     keeps no state that a result or a route depends on: a call's results, bit for
 *     bit, and the kernels it launches are a function of its arguments alone (no
 *     history, thread-safe, any stream).  The one piece of process state is three
 *     diagnostic route counters (dqq_get_option: "lane_list_drains",
 *     "bwd_whole_batches", "fwd_feedback_routes"), bumped on the launch path.  What
 *     adapts to the data -- a steady stream of mostly non-diagonal N <= 8 batches
 *     through DQQ_P_AUTO is faster on other kernels of identical results -- is in the
 *     CALLER's hands: an optional report word the backward fills and hint flags the
 *     caller derives from it (dqq_hint_flags, below).  There are no tuning knobs
 *     (a developer build, -DDQQ_TUNING, has them: csrc/tuning.h);
 *   - like the reference (Solver.cpp:76, :100), numerical failure is not
 *     signalled: a non-PD P or L=0 yields NaNs in the output;
 *   - `warm_start` does not appear: the reference accepts it and overwrites it
 *     before reading it (Solver.cpp:70/80, 529/539).
 *
 * p_layout
 *   DQQ_P_AUTO  (0)  P is (B,N,N).  Tiles whose off-diagonals are all exactly
 *                    zero run the diagonal fast path; every other tile is
 *                    solved by the general dense kernel.  Never assumes.
 *   DQQ_P_DENSE (1)  P is (B,N,N); always the general dense kernel.
 *   DQQ_P_DIAG  (2)  extension: P is the compact diagonal (B,N); grad_P (if
 *                    requested) is the compact diagonal of -dl x^T, (B,N).
 */
#ifndef DIFFQCQP_HIP_H
#define DIFFQCQP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DQQ_P_AUTO 0
#define DQQ_P_DENSE 1
#define DQQ_P_DIAG 2
/* Per-call flag, ORed into p_layout: for 16 < N <= 64 the general (non-diagonal) path runs its kernels in the REFERENCE's
 * summation order (LDS wave kernels; QCQP backward beyond N = 42: a global-memory kernel that needs dqq_scratch_bytes of
 * scratch) instead of the register-resident kernels on the f64 matrix cores, which re-associate sums: forward x within
 * 1e-6 with the oracle's iteration counts on >= 99 % of the problems (all 1024 sampled of BASELINE configs[4]), gradients of the cond ~1e9 Tikhonov systems within 5e-7 (grad_P, grad_q) /
 * 8e-6 (grad_l_n, grad_mu) relative of the reference-order evaluation -- the evaluation-order noise of the reference's own
 * formulas.  10-30x slower; for parity studies.  No effect for N <= 16 (always reference order) or on diagonal tiles. */
#define DQQ_F_REFERENCE_ORDER 0x100
/* Per-call HINT flags, ORed into p_layout (DQQ_P_AUTO, QP / QCQP, N <= 8; ignored elsewhere).  They select between kernels of
 * IDENTICAL results -- bit for bit, on any input -- so a wrong hint costs time and nothing else.  Obtain them from
 * dqq_hint_flags(); never pass them on a stream that is being captured into a graph that will be replayed on other data.
 *   DQQ_F_EXPECT_DENSE      most of the batch is non-diagonal.  Forward (N = 8): one lane per problem, the general solve with a
 *                           problem's whole matrix in its lane's registers, the tile of P staged through LDS and read once.
 *                           Backward (B >= 16384; N = 8: 24576): no classifying launch, the lane-per-problem kernel takes the
 *                           whole batch (and recounts for `report`).
 *   DQQ_F_EXPECT_LONG_LIST  backward: the work-list of non-diagonal problems will be long enough to fill the chip: the drain
 *                           launch is the lane-per-problem kernel (2x faster there; its 512-register waves need an empty SIMD
 *                           each, so it is never launched "just in case": behind a diagonal batch it would stall). */
#define DQQ_F_EXPECT_DENSE 0x200
#define DQQ_F_EXPECT_LONG_LIST 0x400

#define DQQ_E_NULLPTR (-1)     /* a required pointer is NULL */
#define DQQ_E_BAD_SIZE (-2)    /* B < 0, N < 1, odd N for QCQP */
#define DQQ_E_UNSUPPORTED_N (-3) /* DQQ_P_DIAG only: N is not one of the fast-path sizes 2, 4, 8, 16, 32, 64 */
#define DQQ_E_BAD_LAYOUT (-4)
#define DQQ_E_WORKSPACE (-5)   /* workspace missing or too small */
#define DQQ_E_BAD_OPTION (-6)

/* Bytes of device workspace the calls below need for a batch of B problems
 * (fallback work-list of the AUTO layout).  The workspace must be zero-filled
 * ONCE when it is allocated; every call leaves its work-list empty again.  One
 * workspace must not be shared by calls that may run concurrently on different
 * streams. */
size_t dqq_workspace_bytes(int64_t B);

/* Work-list hygiene.  The kernels do not trust the work-list header (csrc/launch.h): a header that did not start at zero --
 * a workspace that was never initialised, memory written over by someone else, a launch chain cut short by an error -- is
 * REPAIRED by the next DQQ_P_AUTO call where that is possible (exit tickets and pick-up counters are re-zeroed by the fast
 * kernel; a stale or out-of-range count is clamped; an entry that is not a problem of the batch is replaced by problem 0,
 * which is then solved once more to the same values) and REPORTED where it is not (problems that cannot be queued get NaN
 * outputs, -1 in iters / ir_steps): nothing is read or written out of bounds and no problem is silently left unsolved.  A
 * sticky "dirty" word is set in the header whenever a COUNT or an ENTRY had to be clamped, replaced or dropped; stale exit
 * tickets and pick-up counters alone (idle words of the drain, re-zeroed before they are used) are repaired silently.
 *   dqq_workspace_reset   zero-fills the header on `stream` (asynchronous): the state a fresh workspace must be in.  The
 *                         recommended way to initialise one (a plain zero-fill of the first dqq_workspace_bytes(0) bytes is
 *                         equivalent).
 *   dqq_workspace_status  *dirty = 1 if any kernel has found the header inconsistent since the last reset.  SYNCHRONISES
 *                         `stream` (a 4-byte read-back): a diagnostic, never on a hot path, and not
 *                         allowed on a stream that is being captured (the synchronisation fails: a hipError_t is returned). */
int dqq_workspace_reset(void* workspace, size_t workspace_bytes, void* stream);
int dqq_workspace_status(const void* workspace, size_t workspace_bytes, void* stream, int* dirty);

/* The C ABI never allocates: every buffer, scratch included, is the caller's.  For sizes beyond the register / LDS
 * kernels of the general path (N > dqq_max_n(..), below) a workgroup-per-problem kernel works out of global memory and
 * needs this many bytes of scratch IN ADDITION to dqq_workspace_bytes(B), in the same `workspace` buffer (the work-list
 * first, the scratch behind it; no initialisation needed).  kind: 0 QP, 1 QCQP, 2 box QP, 3 signed box QP; pass: 0
 * forward, 1 backward; p_layout: as the call will pass it (only DQQ_F_REFERENCE_ORDER matters: QCQP backward 42 < N <= 64
 * needs scratch only with it).  A function of its arguments; 0 for every size the register / LDS kernels hold (all
 * BASELINE configs).  A call whose workspace is smaller than dqq_workspace_bytes(B) + dqq_scratch_bytes(..) returns
 * DQQ_E_WORKSPACE -- with DQQ_P_DENSE too, which otherwise needs no workspace at all.  Since nothing is allocated
 * or freed, a forward + backward pair can be captured into a HIP graph (tests/test_gpu_graph_capture.py). */
size_t dqq_scratch_bytes(int kind, int pass, int N, int64_t B, int p_layout);

/* There is no size limit (the reference has none, Solver.cpp:61): this returns the largest N the register / LDS
 * kernels of the general path hold -- kind 0 = QP forward/backward and the box forwards (64), 1 = QCQP forward (64),
 * 2 = QCQP backward (64; 42 with DQQ_F_REFERENCE_ORDER in p_layout), 3 = box QP backward (21).  Beyond it a
 * workgroup-per-problem kernel works out of global memory, in the reference's operation order, on the caller's scratch
 * (dqq_scratch_bytes). */
int dqq_max_n(int kind, int p_layout);

/* Replaces the loop qcqp.py:29-31 (QPFn2.forward -> diffqcqp.solveQP,
 * pybindings.cpp:17-22 -> Solver::solveQP, Solver.cpp:61-123).
 * iters (B ints, ADMM iterations executed per problem; -1 for a problem that could not be queued for the general kernel --
 * work-list hygiene above -- whose x is NaN) may be NULL.
 * pdiag_out (B,N doubles) / diag_flags_out (B bytes), both optional and DQQ_P_AUTO only: the forward leaves
 * the diagonal of every problem it verified to be diagonal (flag 1; flag 2: the problem sat in a tile with
 * non-zero off-diagonals; flag 0: not examined) for the backward of the SAME P, which then does not read P again
 * for tiles flagged 1 throughout (they take pdiag) or 2 throughout (they go to the general kernel); see
 * dqq_qp_bwd_f64. */
int dqq_qp_fwd_f64(const double* P, const double* q, double* x, int64_t B, int N, double eps, double mu_prox,
                   int max_iter, int adaptive_rho, int p_layout, int* iters, double* pdiag_out,
                   unsigned char* diag_flags_out, void* workspace, size_t workspace_bytes, void* stream);

/* Replaces the loop qcqp.py:45-47 + the assembly qcqp.py:48-51
 * (QPFn2.backward -> diffqcqp.solveDerivativesQP, pybindings.cpp:24-30 ->
 * Solver::dualFromPrimalQP / solveDerivativesQP, Solver.cpp:125-196):
 *   grad_P = -dl x^T (B,N,N), grad_q = -dl (B,N,1).  Either may be NULL
 * (ctx.needs_input_grad).  epsilon is the dual-recovery threshold of
 * solveDerivativesQP (pybindings.cpp:24, default 1e-10; qcqp.py never overrides
 * it).  ir_steps (B ints) may be NULL.  pdiag / diag_flags: optional, what the forward of the same
 * (unchanged) P stored; P itself must still be passed (non-flagged problems read it).
 * report: optional (NULL: none) -- the DEVICE address of one 8-byte word of pinned host memory the caller owns
 * (dqq_device_pointer); the launch that solves the batch's non-diagonal problems stores what it found there, for
 * dqq_hint_flags().  QP / QCQP, DQQ_P_AUTO, N <= 8; ignored otherwise. */
int dqq_qp_bwd_f64(const double* P, const double* q, const double* x, const double* grad_x, double* grad_P,
                   double* grad_q, int64_t B, int N, double epsilon, int p_layout, int* ir_steps,
                   const double* pdiag, const unsigned char* diag_flags, unsigned long long* report, void* workspace,
                   size_t workspace_bytes, void* stream);

/* Replaces the loop qcqp.py:149-151 (QCQPFn2.forward -> diffqcqp.solveQCQP,
 * pybindings.cpp:54-60 -> Solver::solveQCQP, Solver.cpp:521-582).  l_n and mu
 * are the raw inputs; the radius l_n*mu is formed inside (pybindings.cpp:57). */
int dqq_qcqp_fwd_f64(const double* P, const double* q, const double* l_n, const double* mu, double* x,
                     int64_t B, int N, double eps, double mu_prox, int max_iter, int adaptive_rho,
                     int p_layout, int* iters, double* pdiag_out, unsigned char* diag_flags_out, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Replaces the loop qcqp.py:167-172 + the assembly qcqp.py:173-180
 * (QCQPFn2.backward -> diffqcqp.solveDerivativesQCQP, pybindings.cpp:62-71 ->
 * Solver.cpp:584-691):  grad_P = -dl x^T, grad_q = -dl, grad_l_n = E2 dgamma,
 * grad_mu = E1 dgamma.  Any output may be NULL.  gamma / dgamma (B,N/2,1, may be
 * NULL) expose the contact duals and their derivative terms, from which the
 * reference's per-problem return value (E1, E2, blgamma) can be rebuilt:
 * E1 = diag(2 gamma l_n^2 mu), E2 = diag(2 gamma l_n mu^2), blgamma = [dgamma; -grad_q].
 * epsilon: dual-recovery threshold (pybindings.cpp:62, default 1e-10).  report: as dqq_qp_bwd_f64. */
int dqq_qcqp_bwd_f64(const double* P, const double* q, const double* l_n, const double* mu, const double* x,
                     const double* grad_x, double* grad_P, double* grad_q, double* grad_l_n, double* grad_mu,
                     double* gamma, double* dgamma, int64_t B, int N, double epsilon, int p_layout,
                     int* ir_steps, const double* pdiag, const unsigned char* diag_flags, unsigned long long* report,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- the box-constrained members of the same solver family ------------------
 *
 * Replaces the loop qcqp.py:60-62 (BoxQPFn2.forward -> diffqcqp.solveBoxQP, pybindings.cpp:32-37 ->
 * Solver::solveBoxQP, Solver.cpp:198-261): min 1/2 x'Px + q'x, l_min <= x <= l_max.
 * l_min, l_max: (B,N,1).  Same loop as the QP with the projection of Solver.cpp:219-220. */
int dqq_boxqp_fwd_f64(const double* P, const double* q, const double* l_min, const double* l_max, double* x,
                      int64_t B, int N, double eps, double mu_prox, int max_iter, int adaptive_rho, int p_layout,
                      int* iters, double* pdiag_out, unsigned char* diag_flags_out, void* workspace,
                      size_t workspace_bytes, void* stream);

/* Replaces the loop qcqp.py:103-105 (SignedBoxQPFn2.forward -> diffqcqp.solveSignedBoxQP,
 * pybindings.cpp:47-52 -> Solver::solveSignedBoxQP, Solver.cpp:374-439): the box QP with the extra
 * constraint sign(v_i) x_i <= 0 (projection of Solver.cpp:395-398).  v: (B,N,1), raw (its sign is taken
 * inside).  The reference has no backward for this problem (qcqp.py:111 "not implemented"). */
int dqq_signedboxqp_fwd_f64(const double* P, const double* q, const double* l_min, const double* l_max,
                            const double* v, double* x, int64_t B, int N, double eps, double mu_prox,
                            int max_iter, int adaptive_rho, int p_layout, int* iters, double* pdiag_out,
                            unsigned char* diag_flags_out, void* workspace, size_t workspace_bytes, void* stream);

/* Replaces the loop + assembly of BoxQPFn2.backward (qcqp.py:67-94 -> diffqcqp.solveDerivativesBoxQP,
 * pybindings.cpp:39-45 -> Solver::dualFromPrimalBoxQP / solveDerivativesBoxQP, Solver.cpp:263-371).  The
 * shipped Python of that method does not run; the semantics here are the ones it
 * spells out, with the signs finite differences confirm (tests/test_oracle.py):
 *   grad_P = -dl x^T, grad_q = -dl, grad_l_min = -dgamma_lo o gamma_lo, grad_l_max = +dgamma_hi o gamma_hi.
 * Any output may be NULL.  gamma / dgamma (B,2N: lower multipliers | upper multipliers, may be NULL) are the
 * reference's per-problem return values: gamma, and blgamma[0:2N]; blgamma[2N:3N] = -grad_q.
 * ir_steps (B,2 ints, may be NULL): refinement steps of the dual recovery and of the derivative system.
 * General (non-diagonal) P: wave kernel up to dqq_max_n(3) = 21, global-memory workgroup kernel beyond. */
int dqq_boxqp_bwd_f64(const double* P, const double* q, const double* l_min, const double* l_max, const double* x,
                      const double* grad_x, double* grad_P, double* grad_q, double* grad_l_min, double* grad_l_max,
                      double* gamma, double* dgamma, int64_t B, int N, double epsilon, int p_layout, int* ir_steps,
                      const double* pdiag, const unsigned char* diag_flags, void* workspace, size_t workspace_bytes,
                      void* stream);

/* ---- adapting to the data without state in the library -------------------------------------------------------------
 *
 * Behind the diagonal fast path's backward, a second launch solves the problems whose P is not diagonal.  Two kernels can do
 * it -- a team of lanes per problem (best for up to a few thousand such problems) and a lane per problem (2x faster once they
 * fill the chip: 65536 dense 8x8 QCQP problems 110 -> ~55 us) -- and how many there are is known on the device only.  A
 * caller that presents the same kind of batch step after step (a training loop) can close that loop itself:
 *   1. keep one 8-byte word of pinned host memory per (kind, N) -- zero-initialised; per device, per stream or per module as
 *      it sees fit -- and pass its device address (dqq_device_pointer) as `report` to the backward calls;
 *   2. before a call of (kind, pass, N, B), read the word (a plain host load: nothing waits for the device) and OR
 *      dqq_hint_flags(kind, pass, N, B, word) into p_layout.
 * dqq_hint_flags is a pure function; the flags only ever select between kernels of identical results.  First calls, other
 * batch sizes, stream capture (pass no hints there): the argument-determined routes.  Measured, dense 8x8 QCQP, B = 65536,
 * forward + backward: DQQ_P_DENSE 0.131 ms; DQQ_P_AUTO with hints 0.134 ms (from the third step on), without 0.231 ms.
 * diffqcqp_amd/_capi.py does exactly this for the Python layer (one word per (device, kind, N); DQQ_FEEDBACK=0 turns it off).
 *
 * Word format (for the curious; callers only pass it through): bits 0..30 problems found, bit 31 "counted singly",
 * 32..61 B mod 2^30, 62..63 consecutive earlier reports of >= 3/4 B. */
int dqq_hint_flags(int kind, int pass, int N, int64_t B, unsigned long long last_report);

/* The device-side address of 8-byte-aligned pinned (hipHostMalloc / hipHostRegister / torch pin_memory()) host memory:
 * 0, DQQ_E_NULLPTR, DQQ_E_BAD_SIZE (misaligned) or the hipError_t of hipHostGetDevicePointer (not pinned memory). */
int dqq_device_pointer(void* pinned_host, void** device);

/* Route counters (diagnostics): how many launches the hint flags have sent down their alternative routes since the counter
 * was last reset.  Names: "lane_list_drains" (drain launches on the lane-per-problem backward), "bwd_whole_batches"
 * (DQQ_P_AUTO backwards solved whole by that kernel), "fwd_feedback_routes" (N = 8 forwards on one lane per problem).
 * dqq_set_option(name, v) stores v (0 resets); any other name -> DQQ_E_BAD_OPTION.  These are the only names this library
 * knows and none of them changes what a call does.  (A developer build, -DDQQ_TUNING, also accepts the kernel-selection knobs
 * of csrc/tuning.h; they change time, never results.) */
int dqq_set_option(const char* name, int value);
int dqq_get_option(const char* name, int* value);

/* "diffqcqp_hip <version> gfx950" */
const char* dqq_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DIFFQCQP_HIP_H */
