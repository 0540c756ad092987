// bwd_small.hip -- stand-alone kernel of the general (dense P) backward for small N (even, <= 16): the team
// routine of small_bwd_core.h over a whole batch or over the fallback work-list (as in dense.hip).
#include "launch.h"
#include "small_bwd_core.h"

namespace dqq {

template <int KIND, int N>
__global__ __launch_bounds__(256, (SmallSys<KIND, N>::M > 16 ? 1 : 2)) void bwd_small_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ aux0,
    const double* __restrict__ aux1, const double* __restrict__ x, const double* __restrict__ grad_x,
    double* __restrict__ grad_P, double* __restrict__ grad_q, double* __restrict__ gout0, double* __restrict__ gout1,
    double* __restrict__ gamma_out, double* __restrict__ dgamma_out, long B, double dual_eps, int* __restrict__ ir_steps,
    int* __restrict__ ws, int use_worklist, unsigned long long* __restrict__ feedback)
{
    using S = SmallSys<KIND, N>;
    constexpr int T = S::T, TP = 64 / T;
    // An empty work-list -- what this launch finds behind every backward of a diagonal batch -- leaves on ONE scalar load, before
    // anything else: with the exit below the lane / team arithmetic the compiler had put a register spill (a scratch store by each
    // of the 4096 waves) in front of it, and the headline step paid 4.5 us for it (round 4, A/B of the builds).
    // (round 5: the hygiene checks of launch.h come BEHIND that exit too -- in front of it they cost every empty drain 0.4 us,
    // A/B of the builds: tools/ab_libs.py)
    if (use_worklist && ws[kWsCount] == 0) return;
    asm volatile("" ::: "memory");
    const long count = use_worklist ? worklist_checked_count(ws, ws + kWsCount, kWsEntryInts(B)) : B;
    if (count == 0) return;
    asm volatile("" ::: "memory");
    // launch.h: a hint for the next call -- how long the list was.  Only BEHIND the exit above: with the report in front of it
    // (an empty list reported too) the first workgroup read the workspace's shadow word behind the count, two dependent
    // round trips instead of one in a launch that does nothing else, and each empty drain took 0.6 us longer (headline step
    // 58.0 -> 59.5 us, A/B of the builds on one box).  Nothing is lost: a stale "long" in front of an empty list sends the
    // lane-per-problem kernel, and THAT reports the 0 (bwd_lane_dense.hip).
    if (use_worklist && blockIdx.x == 0 && threadIdx.x == 0) worklist_feedback(feedback, ws, B, count);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int team = lane / T, tl = lane % T;
    if (team >= TP && lane != 0) return;   // (T need not divide 64: the lanes left over idle)
    double* sw = smem + (wave * TP + team) * S::LDS_DOUBLES;
    const long nteams = (long)gridDim.x * wpb * TP;
    for (long w = ((long)blockIdx.x * wpb + wave) * TP + team; w < count; w += nteams) {
        const long prob = use_worklist ? worklist_checked_entry(ws, ws[kWsEntries + w], B) : w;
        small_bwd_problem<KIND, N>(P, q, aux0, aux1, x, grad_x, grad_P, grad_q, gout0, gout1, gamma_out, dgamma_out,
                                   ir_steps, prob, dual_eps, sw, tl);
    }
    if (use_worklist && lane == 0) worklist_release(ws, count, (int)(gridDim.x * wpb));
}

template <int KIND, int N>
static hipError_t launch_small(const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    using S = SmallSys<KIND, N>;
    constexpr int TP = 64 / S::T, WPB = S::WPB;
    const size_t lds_bytes = sizeof(double) * (size_t)S::LDS_DOUBLES * TP * WPB;
    const long per_block = (long)WPB * TP;
    const long need = (a.B + per_block - 1) / per_block;
    const long cap = 256L * 16;
    // work-list mode: the list holds at most B entries (a small batch does not pay for idle workgroups); persistent
    // workgroups, 65536 x 8 dense QCQP list, us: 128: 279, 256: 150, 512: 110, 1024: 114, 4096: 155 -- and an EMPTY list costs
    // the same launch latency whatever the grid (round 5, profiles/r06l_drain_grid.txt)
#if defined(DQQ_SMALL_LIST_GRID)   // developer A/B (tools/ab_libs.py)
    const long lim = use_worklist ? DQQ_SMALL_LIST_GRID : cap;
#else
    const long lim = use_worklist ? 512 : cap;
#endif
    const unsigned grid = (unsigned)(need < lim ? (need > 0 ? need : 1) : lim);
    auto kernel = bwd_small_kernel<KIND, N>;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    return launch(kernel, dim3(grid), dim3(64 * WPB), lds_bytes, s, a.P, a.q, a.l_n, a.mu, a.x, a.grad_x, a.grad_P,
                       a.grad_q, a.grad_l_n, a.grad_mu, a.gamma, a.dgamma, a.B, a.epsilon, a.ir_steps, a.ws,
                       use_worklist ? 1 : 0, (use_worklist && hint_applies(KIND, N)) ? a.report : nullptr);
}

// The box QP instantiations (M = 3N: 24 unknowns at N = 8) exist and are correct, but run out of registers
// (256 VGPRs + scratch at N >= 4) and lose to the run-time sized kernel: routed there only for N = 2.
bool bwd_small_supported(int kind, int N)
{
    if (kind == kKindBox) return N == 2;
    return (kind == kKindQP || kind == kKindQCQP) && N >= 2 && N <= 16 && N % 2 == 0;
}

hipError_t launch_bwd_small(int kind, const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
#define DQQ_CASE(NN)                                                      \
    if (a.N == NN) {                                                      \
        switch (kind) {                                                   \
        case 0: return launch_small<0, NN>(a, use_worklist, s);           \
        case 1: return launch_small<1, NN>(a, use_worklist, s);           \
        default: break;                                                   \
        }                                                                 \
    }
    DQQ_CASE(2) DQQ_CASE(4) DQQ_CASE(6) DQQ_CASE(8) DQQ_CASE(10) DQQ_CASE(12) DQQ_CASE(14) DQQ_CASE(16)
#undef DQQ_CASE
    if (kind == kKindBox && a.N == 2) return launch_small<2, 2>(a, use_worklist, s);
    return hipErrorInvalidValue;
}

} // namespace dqq
