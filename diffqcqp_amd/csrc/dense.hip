// dense.hip -- stand-alone kernels of the general (non-diagonal P) path: one wave64 per problem
// (dense_core.h), persistent over the batch or over the fallback work-list the diagonal fast paths
// fill for the tiles they cannot take.
#include <atomic>

#include "dense_core.h"
#include "launch.h"

namespace dqq {

// route counters (tuning.h)
std::atomic<int> g_bwd_whole_batches{0};
std::atomic<int> g_lane_list_drains{0};
std::atomic<int> g_fwd_feedback_routes{0};


// Workgroups hold `wpb` independent waves (wave-private LDS slices, no workgroup barrier): more waves
// per dispatched workgroup keeps the launch cheap when the work-list turns out to be empty.
template <int KIND>
__global__ __launch_bounds__(256) void fwd_dense_kernel(const double* __restrict__ P, const double* __restrict__ q,
                                                        const double* __restrict__ l_n,
                                                        const double* __restrict__ mu_c,
                                                        const double* __restrict__ v_sign, double* __restrict__ x, long B,
                                                        int n, double eps, double mu, int max_iter, int adaptive,
                                                        int* __restrict__ iters, int* __restrict__ ws, int use_worklist,
                                                        int lds_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    double* sw = smem + wave * lds_per_wave;
    const long count = use_worklist ? worklist_count(ws, n, B) : B;
    const long nwaves = (long)gridDim.x * wpb;
    for (long w = (long)blockIdx.x * wpb + wave; w < count; w += nwaves) {
        const long prob = use_worklist ? worklist_entry(ws, n, B, w) : w;
        dense_fwd_problem<KIND>(P, q, l_n, mu_c, v_sign, x, iters, prob, n, eps, mu, max_iter, adaptive, sw, lane);
    }
    if (use_worklist && lane == 0) worklist_release(ws, count, (int)nwaves);
}

// Backward: 64/T problems per wave (dense_core.h: team width T), each team in its own LDS slice.
template <int KIND, int T>
__global__ __launch_bounds__(256) void bwd_dense_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ l_n,
    const double* __restrict__ mu_c, const double* __restrict__ x, const double* __restrict__ grad_x,
    double* __restrict__ grad_P, double* __restrict__ grad_q, double* __restrict__ grad_l_n,
    double* __restrict__ grad_mu, double* __restrict__ gamma_out, double* __restrict__ dgamma_out, long B, int n,
    double dual_eps, int* __restrict__ ir_steps, int* __restrict__ ws, int use_worklist, int lds_per_team)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int TP = 64 / T; // teams per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int team = lane / T, tl = lane % T;
    double* sw = smem + (wave * TP + team) * lds_per_team;
    const long count = use_worklist ? worklist_count(ws, n, B) : B;
    const long nteams = (long)gridDim.x * wpb * TP;
    for (long w = ((long)blockIdx.x * wpb + wave) * TP + team; w < count; w += nteams) {
        const long prob = use_worklist ? worklist_entry(ws, n, B, w) : w;
        dense_bwd_problem<KIND, T>(P, q, l_n, mu_c, x, grad_x, grad_P, grad_q, grad_l_n, grad_mu, gamma_out, dgamma_out,
                                   ir_steps, prob, n, dual_eps, sw, tl);
    }
    if (use_worklist && lane == 0) worklist_release(ws, count, (int)(gridDim.x * wpb));
}

// ---------------------------------------------------------------- launchers
int dense_max_n(int kind)
{
    if (kind == 2) return (kDenseMaxRows * 2) / 3; // n + n/2 <= 64
    if (kind == 3) return kDenseMaxRows / 3;       // box QP backward: 3n <= 64
    return kDenseMaxRows;
}

// Does a backward of this size reach the global-memory kernels, as routed NOW?  (QCQP 42 < N <= 64 only with the
// register-resident kernels switched off.)  dqq_scratch_bytes / dqq_max_n report exactly this, so that the scratch a call
// demands is the scratch the kernels it launches use (the default route of QCQP 42 < N <= 64 demanded 46 MB it
// never touched).
bool bwd_uses_any(int kind, int N, bool ref_order)
{
    if (kind == kKindQCQP && N > 16 && N <= 64 && !ref_order) return false;
    return N > dense_max_n(kind == kKindQP ? 0 : (kind == kKindBox ? 3 : 2));
}
int public_max_n(int kind, bool ref_order)
{
    if (kind == 2 && !ref_order) return 64;
    return dense_max_n(kind);
}

// what the general path can take at all: everything (beyond dqq_max_n: the global-memory kernels of general_any.hip)
bool fwd_dense_supported(int kind, int N) { return N >= 1 && !(kind == kKindQCQP && (N & 1)); }
bool bwd_dense_supported(int kind, int N) { return N >= 1 && !(kind == kKindQCQP && (N & 1)); }

template <typename K>
static hipError_t set_lds(K kernel, size_t bytes)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

// waves per workgroup: as many (<= 4) as fit in 64 KiB of LDS; doubles per wave rounded to 16 B
struct DenseGeom {
    int wpb, lds_per_wave;
    size_t lds_bytes;
    unsigned grid;
};
static DenseGeom dense_geom(int lds_doubles, long B, bool use_worklist)
{
    DenseGeom g;
    g.lds_per_wave = (lds_doubles + 1) & ~1;
    const size_t per_wave = sizeof(double) * (size_t)g.lds_per_wave;
    g.wpb = (int)((64 * 1024) / per_wave);
    if (g.wpb > 4) g.wpb = 4;
    if (g.wpb < 1) g.wpb = 1;
    g.lds_bytes = per_wave * g.wpb;
    // persistent waves loop over the problems; in work-list mode the size of the list is only known on
    // the device, so a fixed 512 workgroups are dispatched (an empty list costs one short launch)
    const long cap = 256L * 16 / g.wpb * (g.wpb > 1 ? 2 : 1);
    const long need = (B + g.wpb - 1) / g.wpb;
    g.grid = (unsigned)(need < (use_worklist ? 512L : cap) ? (need > 0 ? need : 1) : (use_worklist ? 512L : cap));
    return g;
}

template <int KIND>
static hipError_t launch_fwd_wave(const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    const DenseGeom g = dense_geom(dense_fwd_lds_doubles(a.N), a.B, use_worklist);
    hipError_t e = set_lds(fwd_dense_kernel<KIND>, g.lds_bytes);
    if (e != hipSuccess) return e;
    return launch(fwd_dense_kernel<KIND>, dim3(g.grid), dim3(64 * g.wpb), g.lds_bytes, s, a.P, a.q, a.l_n, a.mu, a.v,
                       a.x, a.B, a.N, a.eps, a.mu_prox, a.max_iter, a.adaptive, a.iters, a.ws, use_worklist ? 1 : 0,
                       g.lds_per_wave);
}

hipError_t launch_fwd_dense(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    if (fwd_needs_any(kind, a.N)) return launch_fwd_any(kind, a, use_worklist, s);
    if (fwd_lane_dense_supported(a.N) && knob_lane_dense() != 0)
        return launch_fwd_lane_dense(kind, a, use_worklist, s);
    if (fwd_small_supported(a.N) && knob_small_fwd() != 0) return launch_fwd_small(kind, a, use_worklist, s);
    if (fwd_dense_wave64_supported(a.N) && !a.ref_order)
        return launch_fwd_dense_wave64(kind, a, use_worklist, s);
    switch (kind) {
    case 0: return launch_fwd_wave<0>(a, use_worklist, s);
    case 1: return launch_fwd_wave<1>(a, use_worklist, s);
    case 2: return launch_fwd_wave<2>(a, use_worklist, s);
    case 3: return launch_fwd_wave<3>(a, use_worklist, s);
    default: return hipErrorInvalidValue;
    }
}

template <int KIND, int T>
static hipError_t launch_bwd_team(const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    constexpr int TP = 64 / T;
    const int lds_per_team = (dense_bwd_lds_doubles(KIND, a.N) + 1) & ~1;
    const size_t per_wave = sizeof(double) * (size_t)lds_per_team * TP;
    int wpb = (int)((64 * 1024) / per_wave);
    wpb = wpb > 4 ? 4 : (wpb < 1 ? 1 : wpb);
    const size_t lds_bytes = per_wave * wpb;
    const long per_block = (long)wpb * TP;
    const long need = (a.B + per_block - 1) / per_block;
    const long cap = 256L * 8;
    const unsigned grid = (unsigned)(need < (use_worklist ? 512L : cap) ? (need > 0 ? need : 1) : (use_worklist ? 512L : cap));
    auto kernel = bwd_dense_kernel<KIND, T>;
    hipError_t e = set_lds(kernel, lds_bytes);
    if (e != hipSuccess) return e;
    return launch(kernel, dim3(grid), dim3(64 * wpb), lds_bytes, s, a.P, a.q, a.l_n, a.mu, a.x, a.grad_x, a.grad_P,
                       a.grad_q, a.grad_l_n, a.grad_mu, a.gamma, a.dgamma, a.B, a.N, a.epsilon, a.ir_steps, a.ws,
                       use_worklist ? 1 : 0, lds_per_team);
}

template <int KIND>
static hipError_t launch_bwd_kind(const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    const int rows = dense_bwd_rows(KIND, a.N); // lanes a problem needs
    if (knob_dense_teams() != 0) {
        if (rows <= 8) return launch_bwd_team<KIND, 8>(a, use_worklist, s);
        if (rows <= 16) return launch_bwd_team<KIND, 16>(a, use_worklist, s);
        if (rows <= 32) return launch_bwd_team<KIND, 32>(a, use_worklist, s);
    }
    return launch_bwd_team<KIND, 64>(a, use_worklist, s);
}

bool bwd_lane_takes_auto_batch(int kind, int N, long B, int hints)
{
    // DQQ_F_EXPECT_DENSE (dqq_hint_flags: three quarters of the batch or more queued, twice running): one launch of the lane
    // kernel over everything costs what its waves cost (B / 64 of them, whatever their problems are); classifying first costs
    // a launch that queues the entries through one atomic per workgroup (14 us per 65536) plus the same waves for the queued part
    return knob_lane_bwd() != 0 && knob_bwd_skip_classify() != 0 && bwd_lane_dense_supported(kind, N, B) &&
           (hints & DQQ_F_EXPECT_DENSE) != 0;
}

hipError_t launch_bwd_dense(int kind, const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    // a batch DECLARED dense that fills the chip: a lane per problem (bwd_lane_dense.hip; the same bits as the team kernel).
    // Not in work-list mode unless the list is known to be long: its 512-register waves need an empty SIMD each, and an empty
    // list must cost next to nothing.
    if (!use_worklist && knob_lane_bwd() != 0 && bwd_lane_dense_supported(kind, a.N, a.B))
        return launch_bwd_lane_dense(kind, a, 0, s);
    // ... and the drain launch of a work-list the caller expects to be that long (DQQ_F_EXPECT_LONG_LIST, launch.h)
    if (use_worklist && knob_lane_bwd() != 0 && bwd_lane_dense_supported(kind, a.N, a.B) &&
        (a.hints & DQQ_F_EXPECT_LONG_LIST) != 0) {
        g_lane_list_drains.fetch_add(1, std::memory_order_relaxed);
        return launch_bwd_lane_dense(kind, a, 1, s);
    }
    if (bwd_small_supported(kind, a.N) && knob_small_bwd() != 0) return launch_bwd_small(kind, a, use_worklist, s);
    if (bwd_dense_wave64_supported(kind, a.N) && !a.ref_order)
        return launch_bwd_dense_wave64(kind, a, use_worklist, s);
    // QCQP, 16 < N <= 64: the register-resident block-Cholesky kernels re-associate the sums of these Tikhonov systems
    // (cond(K) ~ 1e9: gradients within 5e-7 / 8e-6 of the reference-order evaluation, the evaluation-order noise of the
    // reference's own formulas); the per-call flag DQQ_F_REFERENCE_ORDER selects the reference-order kernels instead
    // (LDS wave kernel up to N = 42, global-memory kernel beyond: 1e-9, 10-30x slower).
    if (bwd_wave_qcqp_supported(kind, a.N) && !a.ref_order) return launch_bwd_wave_qcqp(a, use_worklist, s);
    if (bwd_wave_qcqp_big_supported(kind, a.N) && !a.ref_order)
        return launch_bwd_wave_qcqp_big(a, use_worklist, s);
    // Systems beyond the wave kernel's 64 rows (QP N > 64, QCQP N > 42, box N > 21): the global-memory kernel in the
    // reference's summation order.
    if (bwd_uses_any(kind, a.N, a.ref_order)) return launch_bwd_any(kind, a, use_worklist, s);
    if (kind == kKindBox) return launch_bwd_kind<2>(a, use_worklist, s);
    return kind == 0 ? launch_bwd_kind<0>(a, use_worklist, s) : launch_bwd_kind<1>(a, use_worklist, s);
}

} // namespace dqq
