// dense_block.hip -- general (dense P) forward solve for N = 32 and N = 64: one 256-thread
// workgroup per problem, everything resident in LDS (BASELINE configs[4]: B=65536, N=64).
//
// Same algorithm as dense_core.h / the reference (Solver::solveQP / solveQCQP, Solver.cpp:61-123,
// 521-582): power iteration, adaptive-rho ADMM with the explicit inverse of P + (rho+mu) I rebuilt
// at every rho update (the reference's llt() + solveInPlace(Identity), Solver.cpp:76-77, 100-101,
// 114-115).  What changes is how the O(N^2) and O(N^3) pieces are spread over 4 waves:
//
//   R = 256/N threads share a row (thread t: row i = t/R, part r = t%R).
//   mat-vec       M^-1 lives in a "slab" layout, element (i,c) at ((c/R)*N + i)*R + c%R, so that step j
//                 of thread t reads word j*256 + t: every ds_read_b64 of a wave is 512 contiguous bytes.
//                 N/R multiply-adds per thread, then a DPP all-reduce over the R lanes of the row.
//   Cholesky      right-looking, in place, row-major with stride N+R (conflict-free for the (row,part)
//                 lane pattern); per column: scale by 1/sqrt(pivot), rank-1 update of the trailing lower
//                 triangle split over all 256 threads; two barriers per column.
//   inverse       L^-1 column by column (R threads per column, DPP reductions, no barriers), then
//                 M^-1 = L^-T L^-1 as N^2 independent dot products written straight into the slab.
// The factorisation therefore sums in a different order than the reference's left-looking LLT and
// column-wise substitutions (differences ~1e-16 * cond, far inside the 1e-6 parity tolerance; the
// iteration counts are still required to match the oracle in tests/).  FP contraction is off.
//
// LDS: two regions of N*(N+R) doubles (68 KiB at N=64) -> two workgroups per CU.
#include "common.h"
#include "launch.h"

namespace dqq {

template <int N>
struct BlockGeom {
    static constexpr int T = 256;
    static constexpr int R = T / N;        // threads per row
    static constexpr int LD = N + R;       // row stride of the row-major regions
    static constexpr int REGION = N * LD;  // doubles per region
    static constexpr int VEC = 4 * N;      // scratch vectors
    static constexpr size_t LDS_BYTES = sizeof(double) * (2 * REGION + VEC);
};

template <int R>
DQQ_D double row_sum(double v) // all-reduce over the R adjacent lanes of a row
{
#pragma clang fp contract(off)
    return LaneGroup<R>::sum(v);
}

// W (row-major, stride LD, lower triangle + diagonal valid) -> L in place (right-looking).
template <int N>
DQQ_D void block_cholesky(double* W, int t)
{
#pragma clang fp contract(off)
    using G = BlockGeom<N>;
    const int i = t / G::R, r = t % G::R;
    for (int k = 0; k < N; ++k) {
        const double d = sqrt(W[k * G::LD + k]);
        __syncthreads(); // everybody has read the pivot
        if (r == 0) {
            if (i == k) W[k * G::LD + k] = d;
            else if (i > k) W[i * G::LD + k] = W[i * G::LD + k] / d;
        }
        __syncthreads();
        if (i > k) {
            const double lik = W[i * G::LD + k];
            for (int j = k + 1 + r; j <= i; j += G::R) W[i * G::LD + j] -= lik * W[j * G::LD + k];
        }
        // the next pivot W[k+1][k+1] is written by row k+1's threads above; the barrier at the top of
        // the next iteration orders it
        __syncthreads();
    }
}

// L (row-major in W) -> LinvT (LinvT[c][j] = (L^-1)[j][c], row-major stride LD) ; R threads per column.
template <int N>
DQQ_D void block_tri_inverse(const double* W, double* LinvT, int t)
{
#pragma clang fp contract(off)
    using G = BlockGeom<N>;
    const int c = t / G::R, r = t % G::R;
    double* y = LinvT + c * G::LD;
    const double ycc = 1.0 / W[c * G::LD + c];
    if (r == 0) y[c] = ycc;
    wave_lds_fence();
    for (int i = c + 1; i < N; ++i) { // rows of a column live in one wave: no workgroup barrier needed
        double s = 0.0;
        for (int j = c + r; j < i; j += G::R) s += W[i * G::LD + j] * y[j];
        s = row_sum<G::R>(s);
        const double yi = -s / W[i * G::LD + i];
        if (r == 0) y[i] = yi;
        wave_lds_fence();
    }
}

// Minv = L^-T L^-1 into the slab layout: Minv[a][b] = sum_{k >= max(a,b)} LinvT[a][k] * LinvT[b][k].
template <int N>
DQQ_D void block_inverse_product(const double* LinvT, double* slab, int t)
{
#pragma clang fp contract(off)
    using G = BlockGeom<N>;
    const int a = t / G::R, r = t % G::R;
    for (int j = 0; j < N / G::R; ++j) {
        const int b = j * G::R + r;
        const int k0 = a > b ? a : b;
        double s = 0.0;
        for (int k = k0; k < N; ++k) s += LinvT[a * G::LD + k] * LinvT[b * G::LD + k];
        slab[j * G::T + t] = s;
    }
}

// out_i = sum_c M[i][c] v[c] with M in the slab layout; result valid in all R lanes of row i.
template <int N>
DQQ_D double block_matvec(const double* slab, const double* v, int t)
{
#pragma clang fp contract(off)
    using G = BlockGeom<N>;
    const int r = t % G::R;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < N / G::R; ++j) s += slab[j * G::T + t] * v[j * G::R + r];
    return row_sum<G::R>(s);
}

// max over the N entries of vec (LDS), identical in every thread; vec must be complete (barrier before).
template <int N>
DQQ_D double block_max(const double* vec, int t)
{
    double m = vec[t % N];
    return LaneGroup<(N < 64 ? N : 64)>::max(m);
}

template <int N>
DQQ_D double block_sum_seq(const double* vec)
{
#pragma clang fp contract(off)
    double s = 0.0;
    for (int i = 0; i < N; ++i) s += vec[i];
    return s;
}

template <int KIND, int N>
__global__ __launch_bounds__(256) void fwd_dense_block_kernel(const double* __restrict__ P,
                                                              const double* __restrict__ q,
                                                              const double* __restrict__ l_n,
                                                              const double* __restrict__ mu_c, double* __restrict__ x,
                                                              long B, double eps, double mu, int max_iter, int adaptive,
                                                              int* __restrict__ iters, int* __restrict__ ws,
                                                              int use_worklist)
{
#pragma clang fp contract(off)
    using G = BlockGeom<N>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* X = smem;                 // region 0: P (slab) -> W / L -> M^-1 (slab)
    double* Y = X + G::REGION;        // region 1: LinvT
    double* va = Y + G::REGION;       // N
    double* vb = va + N;              // N
    double* vc = vb + N;              // N
    double* vd = vc + N;              // N
    const int t = threadIdx.x, i = t / G::R, r = t % G::R;
    const long count = use_worklist ? (long)ws[kWsCount] : B;

    for (long w = blockIdx.x; w < count; w += gridDim.x) {
        const long prob = use_worklist ? (long)ws[kWsEntries + w] : w;
        const double* Pg = P + prob * (long)(N * N);
        __syncthreads();
        // ---- P into the slab layout (for the power iteration)
        for (int idx = t; idx < N * N; idx += G::T) {
            const int row = idx / N, c = idx % N;
            X[((c / G::R) * N + row) * G::R + c % G::R] = Pg[idx];
        }
        // ---- power_iteration, Solver.cpp:46-59
        double v = 1 / sqrt((double)N);
        if (r == 0) va[i] = v * v;
        __syncthreads();
        {
            const double s = block_sum_seq<N>(va);
            if (s > 0) v = v / sqrt(s);
        }
        const int pi_steps = (KIND == 0) ? 10 : 100;
        for (int k = 0; k < pi_steps; ++k) {
            __syncthreads();
            if (r == 0) vb[i] = v;
            __syncthreads();
            const double Av = block_matvec<N>(X, vb, t);
            if (r == 0) va[i] = Av * Av;
            __syncthreads();
            const double s = block_sum_seq<N>(va);
            v = (s > 0) ? Av / sqrt(s) : Av;
        }
        double Lmax;
        {
            __syncthreads();
            if (r == 0) vb[i] = v;
            __syncthreads();
            const double Av = block_matvec<N>(X, vb, t);
            if (r == 0) va[i] = v * Av;
            __syncthreads();
            Lmax = block_sum_seq<N>(va);
        }
        double rho = sqrt(mu * Lmax) * pow(Lmax / mu, .4);       // :72 / :531
        double tau_inc = pow(Lmax / mu, .15), tau_dec = tau_inc; // :73 / :532
        double mdiag = Pg[i * N + i] + (rho + mu);               // accumulated shifted diagonal, :75

        auto refactor = [&]() { // llt() + solveInPlace(Identity) of P + shift, Solver.cpp:76-77
            __syncthreads();
            for (int idx = t; idx < N * N; idx += G::T) {
                const int row = idx / N, c = idx % N;
                if (c <= row) X[row * G::LD + c] = (c == row) ? 0.0 : Pg[idx];
            }
            __syncthreads();
            if (r == 0) X[i * G::LD + i] = mdiag;
            __syncthreads();
            block_cholesky<N>(X, t);
            block_tri_inverse<N>(X, Y, t);
            __syncthreads();
            block_inverse_product<N>(Y, X, t);
            __syncthreads();
        };
        refactor();

        const double qi = q[prob * N + i];
        double rad = 0.0;
        if (KIND == 1) rad = l_n[prob * (N / 2) + i / 2] * mu_c[prob * (N / 2) + i / 2];
        double qp = qi, l2 = 0.0, l2p = 0.0, u = 0.0;
        int rho_up = 0, cpt = 0, it_done = 0;
        for (int it = 0; it < max_iter; ++it) {
            it_done = it + 1;
            if (r == 0) va[i] = rho * l2 - u - qp;
            __syncthreads();
            const double l = block_matvec<N>(X, va, t);                  // :80 / :539
            qp = qi - mu * l;                                            // :81 / :540
            double z = kAlpha * l + (1 - kAlpha) * l2 + u / rho;         // :82 / :541
            if (KIND == 0) {
                z = z < 0 ? 0 : z;
            } else {                                                     // prox_circle, :505-519
                const double other = __shfl_xor(z, G::R, 64);            // the row i^1 sits R lanes away
                const double a = (i & 1) ? other : z, b = (i & 1) ? z : other;
                const double nrm = sqrt(a * a + b * b);
                if (nrm > rad) z = z * rad / nrm;
            }
            l2 = z;
            u += rho * (kAlpha * l + (1 - kAlpha) * l2p - l2);           // :83 / :543
            const double rd_i = (KIND == 0) ? fabs(rho * (l2 - l2p)) : fabs(l2 - l2p);
            const double rp_i = fabs(l2 - (kAlpha * l + (1 - kAlpha) * l2p));
            l2p = l2;
            if (r == 0) { vb[i] = rd_i; vc[i] = rp_i; if (KIND == 1) vd[i] = l * l; }
            __syncthreads();
            const double rdm = block_max<N>(vb, t), res_prim = block_max<N>(vc, t);
            const double res_dual = (KIND == 0) ? rdm : rho * rdm;
            bool stop = res_dual < eps;                                  // :88
            if (KIND == 1) stop = (res_prim < eps + kEpsRel * sqrt(block_sum_seq<N>(vd))) && stop; // :548
            __syncthreads(); // va/vb/vc are rewritten next iteration
            if (stop) break;
            if (adaptive) {
                bool upd = false;
                if (res_prim > kMuThresh * res_dual) {                   // :92 / :552
                    if (cpt % 5 == 0) {
                        if (rho_up == -1) {
                            tau_inc = 1 + .8 * (tau_inc - 1);
                            if (KIND == 0) tau_dec = 1 + .8 * (tau_dec - 1);
                        }
                        mdiag += rho * (tau_inc - 1);
                        rho *= tau_inc;
                        rho_up = 1;
                        upd = true;
                    }
                    cpt++;
                } else if (res_dual > kMuThresh * res_prim) {            // :106 / :566
                    if (cpt % 5 == 0) {
                        if (rho_up == 1) {
                            if (KIND == 0) tau_inc = 1 + .8 * (tau_inc - 1);
                            tau_dec = 1 + .8 * (tau_dec - 1);
                        }
                        mdiag += rho * (1. / tau_dec - 1);
                        rho /= tau_dec;
                        rho_up = -1;
                        upd = true;
                    }
                    cpt++;
                }
                if (upd) refactor();
            }
        }
        if (r == 0) x[prob * N + i] = l2;
        if (iters != nullptr && t == 0) iters[prob] = it_done;
    }
    // last workgroup out re-zeroes the work-list header (nothing to do when the list was empty)
    if (use_worklist && count > 0 && t == 0) {
        const int tk = atomicAdd(&ws[kWsTicket], 1);
        if (tk == (int)gridDim.x - 1) {
            ws[kWsCount] = 0;
            ws[kWsTicket] = 0;
        }
    }
}

template <int KIND, int N>
static hipError_t launch_block(const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    using G = BlockGeom<N>;
    auto kernel = fwd_dense_block_kernel<KIND, N>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return e;
    const long cap = 256L * 2 * 2; // persistent: two workgroups per CU fit in LDS, x2 for load balance
    const unsigned grid = use_worklist ? 512u : (unsigned)(a.B < cap ? (a.B > 0 ? a.B : 1) : cap);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), G::LDS_BYTES, s, a.P, a.q, a.l_n, a.mu, a.x, a.B, a.eps, a.mu_prox,
                       a.max_iter, a.adaptive, a.iters, a.ws, use_worklist ? 1 : 0);
    return hipGetLastError();
}

bool fwd_dense_block_supported(int N) { return N == 32 || N == 64; }

hipError_t launch_fwd_dense_block(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    if (a.N == 64) return kind == 0 ? launch_block<0, 64>(a, use_worklist, s) : launch_block<1, 64>(a, use_worklist, s);
    if (a.N == 32) return kind == 0 ? launch_block<0, 32>(a, use_worklist, s) : launch_block<1, 32>(a, use_worklist, s);
    return hipErrorInvalidValue;
}

} // namespace dqq
