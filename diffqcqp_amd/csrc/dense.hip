// dense.hip -- stand-alone kernels of the general (non-diagonal P) path: one wave64 per problem
// (dense_core.h), persistent over the batch or over the fallback work-list the diagonal fast paths
// fill for the tiles they cannot take.
#include "dense_core.h"
#include "launch.h"

namespace dqq {

// Last workgroup out re-zeroes the work-list header for the next call.  With an
// empty work-list (the common case: every tile was diagonal) there is nothing to
// reset and no workgroup touches the ticket -- 1024 same-address atomics would
// otherwise serialise into ~13 us of an otherwise empty launch.
static DQQ_D void worklist_release(int* ws, int lane, long count)
{
    if (count > 0 && lane == 0) {
        const int t = atomicAdd(&ws[kWsTicket], 1);
        if (t == (int)gridDim.x - 1) {
            ws[kWsCount] = 0;
            ws[kWsTicket] = 0;
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(64) void fwd_dense_kernel(const double* __restrict__ P, const double* __restrict__ q,
                                                       const double* __restrict__ l_n,
                                                       const double* __restrict__ mu_c, double* __restrict__ x, long B,
                                                       int n, double eps, double mu, int max_iter, int adaptive,
                                                       int* __restrict__ iters, int* __restrict__ ws, int use_worklist)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const long count = use_worklist ? (long)ws[kWsCount] : B;
    for (long w = blockIdx.x; w < count; w += gridDim.x) {
        const long prob = use_worklist ? (long)ws[kWsEntries + w] : w;
        dense_fwd_problem<KIND>(P, q, l_n, mu_c, x, iters, prob, n, eps, mu, max_iter, adaptive, smem, lane);
    }
    if (use_worklist) worklist_release(ws, lane, count);
}

template <int KIND>
__global__ __launch_bounds__(64) void bwd_dense_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ l_n,
    const double* __restrict__ mu_c, const double* __restrict__ x, const double* __restrict__ grad_x,
    double* __restrict__ grad_P, double* __restrict__ grad_q, double* __restrict__ grad_l_n,
    double* __restrict__ grad_mu, long B, int n, int* __restrict__ ir_steps, int* __restrict__ ws, int use_worklist)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x;
    const long count = use_worklist ? (long)ws[kWsCount] : B;
    for (long w = blockIdx.x; w < count; w += gridDim.x) {
        const long prob = use_worklist ? (long)ws[kWsEntries + w] : w;
        dense_bwd_problem<KIND>(P, q, l_n, mu_c, x, grad_x, grad_P, grad_q, grad_l_n, grad_mu, ir_steps, prob, n, smem,
                                lane);
    }
    if (use_worklist) worklist_release(ws, lane, count);
}

// ---------------------------------------------------------------- launchers
int dense_max_n(int kind)
{
    if (kind == 2) return (kDenseMaxRows * 2) / 3; // n + n/2 <= 64
    return kDenseMaxRows;
}

static size_t fwd_dense_lds(int n) { return sizeof(double) * (size_t)dense_fwd_lds_doubles(n); }
static size_t bwd_dense_lds(int kind, int n) { return sizeof(double) * (size_t)dense_bwd_lds_doubles(kind, n); }

template <typename K>
static hipError_t set_lds(K kernel, size_t bytes)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

static unsigned dense_grid(long B, bool use_worklist)
{
    // persistent: enough single-wave workgroups to fill the chip, each loops over problems
    const long cap = 256L * 16;
    // work-list mode: the size of the list is only known on the device; 512 single-wave workgroups loop
    // over it (an empty list -- every tile was diagonal -- costs one short launch)
    if (use_worklist) return 512;
    return (unsigned)(B < cap ? (B > 0 ? B : 1) : cap);
}

hipError_t launch_fwd_dense(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    const size_t lds = fwd_dense_lds(a.N);
    const unsigned grid = dense_grid(a.B, use_worklist);
    hipError_t e;
    if (kind == 0) {
        if ((e = set_lds(fwd_dense_kernel<0>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(fwd_dense_kernel<0>, dim3(grid), dim3(64), lds, s, a.P, a.q, a.l_n, a.mu, a.x, a.B, a.N,
                           a.eps, a.mu_prox, a.max_iter, a.adaptive, a.iters, a.ws, use_worklist ? 1 : 0);
    } else {
        if ((e = set_lds(fwd_dense_kernel<1>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(fwd_dense_kernel<1>, dim3(grid), dim3(64), lds, s, a.P, a.q, a.l_n, a.mu, a.x, a.B, a.N,
                           a.eps, a.mu_prox, a.max_iter, a.adaptive, a.iters, a.ws, use_worklist ? 1 : 0);
    }
    return hipGetLastError();
}

hipError_t launch_bwd_dense(int kind, const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    const size_t lds = bwd_dense_lds(kind, a.N);
    const unsigned grid = dense_grid(a.B, use_worklist);
    hipError_t e;
    if (kind == 0) {
        if ((e = set_lds(bwd_dense_kernel<0>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(bwd_dense_kernel<0>, dim3(grid), dim3(64), lds, s, a.P, a.q, a.l_n, a.mu, a.x, a.grad_x,
                           a.grad_P, a.grad_q, a.grad_l_n, a.grad_mu, a.B, a.N, a.ir_steps, a.ws,
                           use_worklist ? 1 : 0);
    } else {
        if ((e = set_lds(bwd_dense_kernel<1>, lds)) != hipSuccess) return e;
        hipLaunchKernelGGL(bwd_dense_kernel<1>, dim3(grid), dim3(64), lds, s, a.P, a.q, a.l_n, a.mu, a.x, a.grad_x,
                           a.grad_P, a.grad_q, a.grad_l_n, a.grad_mu, a.B, a.N, a.ir_steps, a.ws,
                           use_worklist ? 1 : 0);
    }
    return hipGetLastError();
}

} // namespace dqq
