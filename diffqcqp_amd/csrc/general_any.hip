// general_any.hip -- the general (dense P) kernels without a size limit: one 256-thread workgroup per problem,
// matrices in a per-workgroup slice of global memory (any_core.h).  Takes every N the register / LDS kernels do
// not hold (forward and QP backward N > 64, QCQP backward N > 42, box QP backward N > 21) in the reference's
// operation order.  The two matrices of the O(n^3) loops are staged in LDS while they fit (forward N <= 98, systems
// of up to 98 unknowns in the backward: QCQP N <= 64, box N <= 32).  The scratch slices come from the stream-ordered allocator (hipMallocAsync / hipFreeAsync on
// the caller's stream: no synchronisation, and the only entry points that allocate).
#include "any_core.h"
#include "launch.h"

namespace dqq {

// LDS = true: the two matrices of the O(n^3) loops live in dynamic LDS (one workgroup's worth: up to 152 KiB, one
// workgroup per CU at the largest sizes), everything else in the global slice.
template <int KIND, bool LDS>
__global__ __launch_bounds__(kAnyT) void fwd_any_kernel(const double* __restrict__ P, const double* __restrict__ q,
                                                        const double* __restrict__ l_n, const double* __restrict__ mu_c,
                                                        const double* __restrict__ v_sign, double* __restrict__ x, long B,
                                                        int n, double eps, double mu, int max_iter, int adaptive,
                                                        int* __restrict__ iters, int* __restrict__ ws, int use_worklist,
                                                        double* __restrict__ scratch, long scratch_stride)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[kAnyT];
    const int t = threadIdx.x;
    double* scr = scratch + (long)blockIdx.x * scratch_stride;
    const long mat = any_mat_doubles(n);
    double* A = LDS ? smem : scr;
    double* Ainv = LDS ? smem + mat : scr + mat;
    double* vec = LDS ? scr : scr + 2 * mat;
    const long count = use_worklist ? worklist_count(ws, n, B) : B;
    for (long w = blockIdx.x; w < count; w += gridDim.x) {
        const long prob = use_worklist ? worklist_entry(ws, n, B, w) : w;
        any_fwd_problem<KIND>(P, q, l_n, mu_c, v_sign, x, iters, prob, n, eps, mu, max_iter, adaptive, A, Ainv, vec, red,
                              t);
    }
    if (use_worklist && t == 0) worklist_release(ws, count, (int)gridDim.x);
}

template <int KIND, bool LDS>
__global__ __launch_bounds__(kAnyT) void bwd_any_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ l_n,
    const double* __restrict__ mu_c, const double* __restrict__ x, const double* __restrict__ grad_x,
    double* __restrict__ grad_P, double* __restrict__ grad_q, double* __restrict__ grad_l_n,
    double* __restrict__ grad_mu, double* __restrict__ gamma_out, double* __restrict__ dgamma_out, long B, int n,
    double dual_eps, int* __restrict__ ir_steps, int* __restrict__ ws, int use_worklist, double* __restrict__ scratch,
    long scratch_stride)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int t = threadIdx.x;
    double* scr = scratch + (long)blockIdx.x * scratch_stride;
    const long mat = any_mat_doubles(any_bwd_rows(KIND, n));
    double* K = scr;
    double* At = LDS ? smem : scr + mat;
    double* Kinv = LDS ? smem + mat : scr + 2 * mat;
    double* vec = LDS ? scr + mat : scr + 3 * mat;
    const long count = use_worklist ? worklist_count(ws, n, B) : B;
    for (long w = blockIdx.x; w < count; w += gridDim.x) {
        const long prob = use_worklist ? worklist_entry(ws, n, B, w) : w;
        any_bwd_problem<KIND>(P, q, l_n, mu_c, x, grad_x, grad_P, grad_q, grad_l_n, grad_mu, gamma_out, dgamma_out,
                              ir_steps, prob, n, dual_eps, At, K, Kinv, vec, t);
    }
    if (use_worklist && t == 0) worklist_release(ws, count, (int)gridDim.x);
}

// persistent grid: two workgroups per CU when the batch is that large -- fewer when their scratch slices
// (stride doubles each) would not fit a 4 GiB budget (N in the thousands).  A function of (stride, B) only, in
// work-list mode too (the list cannot hold more than B entries): the caller sizes the scratch from it.
static unsigned any_grid(long B, long stride)
{
    long cap = 512;
    const long budget = (4L << 30) / (8 * (stride > 0 ? stride : 1));
    if (cap > budget) cap = budget > 0 ? budget : 1;
    return (unsigned)(B < cap ? (B > 0 ? B : 1) : cap);
}

static long any_fwd_stride(int N)
{
    const long mat = any_mat_doubles(N), vec = any_fwd_vec_doubles(N);
    return (sizeof(double) * 2 * (size_t)mat <= kAnyLdsBytes) ? ((vec + 1) & ~1L) : ((2 * mat + vec + 1) & ~1L);
}

static long any_bwd_stride(int kind, int N)
{
    const int k = kind == kKindQP ? 0 : (kind == kKindQCQP ? 1 : 2);
    const long mat = any_mat_doubles(any_bwd_rows(k, N)), vec = any_bwd_vec_doubles(k, N);
    return (sizeof(double) * 2 * (size_t)mat <= kAnyLdsBytes) ? ((mat + vec + 1) & ~1L) : ((3 * mat + vec + 1) & ~1L);
}

size_t any_scratch_bytes(int kind, bool backward, int N, long B)
{
    if (B <= 0) return 0;
    const long stride = backward ? any_bwd_stride(kind, N) : any_fwd_stride(N);
    return sizeof(double) * (size_t)stride * any_grid(B, stride);
}

// The sizes beyond the register / LDS kernels of the general path (dqq_max_n).  The backward's answer follows the route
// (dense.hip: bwd_uses_any): the scratch a call demands is the scratch its kernels use.
bool fwd_needs_any(int kind, int N) { return N > dense_max_n(kind == kKindQCQP ? 1 : 0); }
bool bwd_needs_any(int kind, int N, bool ref_order) { return bwd_uses_any(kind, N, ref_order); }

template <typename Kern, typename... Args>
static hipError_t launch_any(Kern kernel, size_t lds_bytes, long stride, long B, double* scratch, hipStream_t s,
                             Args... args)
{
    if (scratch == nullptr) return hipErrorInvalidValue; // capi.hip checks the workspace before it gets here
    const unsigned grid = any_grid(B, stride);
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    return launch(kernel, dim3(grid), dim3(kAnyT), lds_bytes, s, args..., scratch, stride);
}

template <int KIND>
static hipError_t launch_fwd_any_kind(const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    const long mat = any_mat_doubles(a.N);
    const size_t lds = sizeof(double) * 2 * (size_t)mat;
    const int wl = use_worklist ? 1 : 0;
    const long stride = any_fwd_stride(a.N);
    if (lds <= kAnyLdsBytes)
        return launch_any(fwd_any_kernel<KIND, true>, lds, stride, a.B, a.scratch, s, a.P, a.q, a.l_n, a.mu,
                          a.v, a.x, a.B, a.N, a.eps, a.mu_prox, a.max_iter, a.adaptive, a.iters, a.ws, wl);
    return launch_any(fwd_any_kernel<KIND, false>, 0, stride, a.B, a.scratch, s, a.P, a.q, a.l_n,
                      a.mu, a.v, a.x, a.B, a.N, a.eps, a.mu_prox, a.max_iter, a.adaptive, a.iters, a.ws, wl);
}

hipError_t launch_fwd_any(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    switch (kind) {
    case 0: return launch_fwd_any_kind<0>(a, use_worklist, s);
    case 1: return launch_fwd_any_kind<1>(a, use_worklist, s);
    case 2: return launch_fwd_any_kind<2>(a, use_worklist, s);
    case 3: return launch_fwd_any_kind<3>(a, use_worklist, s);
    default: return hipErrorInvalidValue;
    }
}

template <int KIND>
static hipError_t launch_bwd_any_kind(const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    const long mat = any_mat_doubles(any_bwd_rows(KIND, a.N));
    const size_t lds = sizeof(double) * 2 * (size_t)mat;
    const int wl = use_worklist ? 1 : 0;
    const long stride = any_bwd_stride(KIND == 0 ? kKindQP : (KIND == 1 ? kKindQCQP : kKindBox), a.N);
    if (lds <= kAnyLdsBytes)
        return launch_any(bwd_any_kernel<KIND, true>, lds, stride, a.B, a.scratch, s, a.P, a.q, a.l_n,
                          a.mu, a.x, a.grad_x, a.grad_P, a.grad_q, a.grad_l_n, a.grad_mu, a.gamma, a.dgamma, a.B, a.N,
                          a.epsilon, a.ir_steps, a.ws, wl);
    return launch_any(bwd_any_kernel<KIND, false>, 0, stride, a.B, a.scratch, s, a.P, a.q, a.l_n,
                      a.mu, a.x, a.grad_x, a.grad_P, a.grad_q, a.grad_l_n, a.grad_mu, a.gamma, a.dgamma, a.B, a.N,
                      a.epsilon, a.ir_steps, a.ws, wl);
}

hipError_t launch_bwd_any(int kind, const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    switch (kind) {
    case kKindQP: return launch_bwd_any_kind<0>(a, use_worklist, s);
    case kKindQCQP: return launch_bwd_any_kind<1>(a, use_worklist, s);
    case kKindBox: return launch_bwd_any_kind<2>(a, use_worklist, s);
    default: return hipErrorInvalidValue;
    }
}

} // namespace dqq
