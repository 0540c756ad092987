// dense_block.hip -- general (dense P) forward solve for N = 32 and N = 64: one 256-thread
// workgroup per problem (BASELINE configs[4]: B=65536, N=64).
//
// Same algorithm as dense_core.h / the reference (Solver::solveQP / solveQCQP, Solver.cpp:61-123,
// 521-582): power iteration, adaptive-rho ADMM with the explicit inverse of P + (rho+mu) I rebuilt
// at every rho update (the reference's llt() + solveInPlace(Identity), Solver.cpp:76-77, 100-101,
// 114-115).  What changes is how the O(N^2) and O(N^3) pieces are spread over 4 waves:
//
//   refactor      blocked (16-column panels) right-looking Cholesky in LDS, L^-1 by blocked forward
//                 substitution and M^-1 = L^-T L^-1 as tile products -- all 16x16 tile products run on
//                 the f64 matrix cores (v_mfma_f64_16x16x4_f64); only the 16x16 diagonal blocks are
//                 factored / inverted in registers (v_readlane broadcasts), redundantly by every wave.
//                 Two workgroup barriers per panel, none in the inverse.
//   iteration     every wave keeps ALL rows (one per lane) and a quarter of the columns of M^-1 in
//                 registers; one barrier per ADMM iteration (see WaveRows below).
// The factorisation sums in a different order than the reference's left-looking LLT and column-wise
// substitutions, and FP contraction is on here (the matrix cores fuse anyway): differences are
// ~1e-16 * cond, far inside the 1e-6 parity tolerance; the iteration counts are still required to
// match the oracle in tests/.
//
// LDS: two regions of N*(N+R) doubles (68 KiB at N=64) + 4.5 KiB exchange area = 72.5 KiB -> two
// workgroups per CU (160 KiB); the scratch tiles of the blocked inverse live inside the second region.
#include "block_core.h"
#include "kkt_core.h"
#include "launch.h"

namespace dqq {

// ---- iteration layout: every wave holds ALL rows (lane l = row l & (N-1); the upper half of the wave
// duplicates the lower one when N = 32) and a block of CW = N/4 columns of the matrix in registers.
// A mat-vec is: the wave drops its CW right-hand-side entries into a wave-private LDS strip and reads
// them back as broadcasts (ds_read_b128 of a wave-uniform address), CW multiply-adds per lane, one
// ds_write of the partial row sums, ONE workgroup barrier, and four ds_reads summed in a fixed order --
// so the four waves end up with bit-identical vectors and run the element-wise ADMM update, the
// residual reductions and the stop / rho logic redundantly, with no further exchange.  The partial-sum
// buffer alternates with the parity of the mat-vec counter: a buffer is rewritten only after the next
// barrier, which every wave reaches after its reads.
template <int N>
struct WaveRows {
    static constexpr int CW = N / 4;
    static constexpr int LDS_DOUBLES = 8 * N + 4 * CW; // partial sums (2 x 4 x N) + one strip per wave
    int wave, lane, row, parity;
    double* part;
    double* strip;

    DQQ_D void init(double* base, int t)
    {
        wave = __builtin_amdgcn_readfirstlane(t >> 6);
        lane = t & 63;
        row = lane & (N - 1);
        parity = 0;
        part = base;
        strip = base + 8 * N + CW * wave;
    }
    // sum_c M[row][c] v[c]; m[k] = M[row][CW*wave + k], v = this lane's row entry of the vector
    DQQ_D double matvec(const double (&m)[CW], double v)
    {
        if ((unsigned)(lane - CW * wave) < (unsigned)CW) strip[lane - CW * wave] = v;
        wave_lds_fence();
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int k = 0; k < CW; k += 4) {
            a0 = fma(m[k + 0], strip[k + 0], a0);
            a1 = fma(m[k + 1], strip[k + 1], a1);
            a2 = fma(m[k + 2], strip[k + 2], a2);
            a3 = fma(m[k + 3], strip[k + 3], a3);
        }
        double* buf = part + parity * 4 * N;
        parity ^= 1;
        if (lane < N) buf[wave * N + row] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        return ((buf[row] + buf[N + row]) + buf[2 * N + row]) + buf[3 * N + row];
    }
    // max over the N rows of a and of b (both >= 0), identical in every lane of every wave.  One
    // butterfly serves both: the lower half-wave carries a, the upper one b (gfx950 v_permlane32_swap
    // exchanges the half-waves of two registers; at N = 32 the upper half already duplicates the rows).
    DQQ_D void max2_rows(double a, double b, double& ma, double& mb) const
    {
        double v;
        if (N == 64) {
            const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
            const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
            // [0] = {a[0..31], b[0..31]}, [1] = {a[32..63], b[32..63]}  (tools/ubench/permlane_probe.hip)
            v = fmax(__hiloint2double(hi[0], lo[0]), __hiloint2double(hi[1], lo[1]));
        } else {
            v = lane < 32 ? a : b;
        }
        const double m = LaneGroup<16>::max(v);
        ma = fmax(lane_bcast(m, 0), lane_bcast(m, 16));
        mb = fmax(lane_bcast(m, 32), lane_bcast(m, 48));
    }
    static DQQ_D double sum_rows(double v)
    {
        const double m = LaneGroup<16>::sum(v);
        double out = lane_bcast(m, 0) + lane_bcast(m, 16);
        if (N == 64) out = (out + lane_bcast(m, 32)) + lane_bcast(m, 48);
        return out;
    }
};

template <int KIND, int N>
__global__ __launch_bounds__(256, N == 32 ? 3 : 2) void fwd_dense_block_kernel(const double* __restrict__ P,
                                                              const double* __restrict__ q,
                                                              const double* __restrict__ l_n,
                                                              const double* __restrict__ mu_c,
                                                              const double* __restrict__ v_sign, double* __restrict__ x,
                                                              long B, double eps, double mu, int max_iter, int adaptive,
                                                              int* __restrict__ iters, int* __restrict__ ws,
                                                              int use_worklist)
{
    // KIND 2 / 3 (box / signed box QP, Solver.cpp:198-261 / 374-439): l_n = l_min, mu_c = l_max per coordinate
    using G = BlockGeom<N>;
    using WR = WaveRows<N>;
    constexpr int CW = WR::CW;
    constexpr bool QP_LIKE = (KIND != 1);
    static_assert(WR::LDS_DOUBLES == G::VEC, "exchange area size");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* X = smem;                 // region 0: W / L -> M^-1 (row-major, stride LD)
    double* Y = X + G::REGION;        // region 1: LinvT
    double* part = Y + G::REGION;     // WaveRows exchange area
    double* fail_flag = part + G::VEC;
    const int t = threadIdx.x;
    WR wr;
    wr.init(part, t);
    const int row = wr.row;
    const long count = use_worklist ? worklist_count(ws, N) : B;

    for (long w = blockIdx.x; w < count; w += gridDim.x) {
        const long prob = use_worklist ? worklist_entry(ws, N, B, w) : w;
        const double* Pg = P + prob * (long)(N * N);
        double m[CW]; // the wave's column block of P (power iteration), then of M^-1 (ADMM)
#pragma unroll
        for (int k = 0; k < CW; ++k) m[k] = Pg[row * N + CW * wr.wave + k];
        // ---- power_iteration, Solver.cpp:46-59
        double v = 1 / sqrt((double)N);
        {
            const double s = WR::sum_rows(v * v);
            if (s > 0) v = v / sqrt(s);
        }
        const int pi_steps = QP_LIKE ? 10 : 100;
        for (int k = 0; k < pi_steps; ++k) {
            const double Av = wr.matvec(m, v);
            const double s = WR::sum_rows(Av * Av); // normalised every step like the reference (:53)
            v = s > 0 ? Av / sqrt(s) : Av;
        }
        const double Lmax = WR::sum_rows(v * wr.matvec(m, v));
        bool bad = false;
        if (t == 0) *fail_flag = 0.0;
        RhoSchedule sched;
        sched.init(Lmax, mu);                                    // :72-73 / :531-532
        double rho = sched.rho;
        double mdiag = Pg[row * N + row] + (rho + mu);           // accumulated shifted diagonal, :75

        auto refactor = [&]() { // llt() + solveInPlace(Identity) of P + shift, Solver.cpp:76-77
            __syncthreads();
            for (int idx = t; idx < N * N; idx += G::T) {
                const int rr = idx / N, c = idx % N;
                if (c < rr) X[rr * G::LD + c] = Pg[idx];
            }
            if (t < N) X[row * G::LD + row] = mdiag;
            __syncthreads();
            block_cholesky_and_inverse<N>(X, Y, fail_flag, t, bad);
            block_inverse_product<N>(Y, X, t);
            __syncthreads();
            // M^-1 is symmetric: read the column block entries as rows (conflict-free)
#pragma unroll
            for (int k = 0; k < CW; ++k) m[k] = X[(CW * wr.wave + k) * G::LD + row];
        };

        const double qi = q[prob * N + row];
        double rad = 0.0;
        if (KIND == 1) rad = l_n[prob * (N / 2) + row / 2] * mu_c[prob * (N / 2) + row / 2];
        double blo = 0.0, bhi = 0.0, bsg = 0.0;
        if (KIND >= 2) {
            blo = l_n[prob * N + row];
            bhi = mu_c[prob * N + row];
            if (KIND == 3) { const double vv = v_sign[prob * N + row]; bsg = (double)((vv > 0) - (vv < 0)); } // :395
        }
        double qp = qi, l2 = 0.0, l2p = 0.0, u = 0.0;
        int it_done = 0;
        bool need_refactor = true;
        double inv_rho = 1.0 / rho;
        for (int it = 0; it < max_iter; ++it) {
            if (need_refactor) { refactor(); need_refactor = false; inv_rho = 1.0 / rho; } // the ONLY call site
            it_done = it + 1;
            const double l = wr.matvec(m, rho * l2 - u - qp);            // :80 / :539
            qp = qi - mu * l;                                            // :81 / :540
            double z = kAlpha * l + (1 - kAlpha) * l2 + u * inv_rho;     // :82 / :541 (inv_rho = 1/rho)
            if (KIND == 0) {
                z = z < 0 ? 0 : z;
            } else if (KIND >= 2) {
                z = z < blo ? blo : z;                                   // cwiseMax(l_min), :219 / :396
                z = bhi < z ? bhi : z;                                   // cwiseMin(l_max), :220 / :397
                if (KIND == 3) {                                         // v o min(v o l_2, 0), :398
                    double mm = bsg * z;
                    mm = 0 < mm ? 0 : mm;
                    z = bsg * mm;
                }
            } else {                                                     // prox_circle, :505-519
                const double other = partner<1>(z);                      // row ^ 1
                const double a = (row & 1) ? other : z, b = (row & 1) ? z : other;
                const double nrm = sqrt(a * a + b * b);
                if (nrm > rad) z = z * rad / nrm;
            }
            l2 = z;
            u += rho * (kAlpha * l + (1 - kAlpha) * l2p - l2);           // :83 / :543
            const double rd_i = QP_LIKE ? fabs(rho * (l2 - l2p)) : fabs(l2 - l2p);
            const double rp_i = fabs(l2 - (kAlpha * l + (1 - kAlpha) * l2p));
            l2p = l2;
            double rdm, res_prim;
            wr.max2_rows(rd_i, rp_i, rdm, res_prim);
            const double res_dual = QP_LIKE ? rdm : rho * rdm;
            bool stop = res_dual < eps;                                  // :88
            if (KIND == 1) stop = (res_prim < eps + kEpsRel * sqrt(WR::sum_rows(l * l))) && stop; // :548
            if (stop) break;
            if (adaptive) {
                double delta = 0.0;
                if (sched.template update<QP_LIKE, false>(res_prim, res_dual, delta)) { // :90-120 / :550-580
                    mdiag += delta;
                    rho = sched.rho;
                    need_refactor = true;
                }
            }
        }
        __syncthreads(); // the failure flag of the last refactor is visible; X/Y free for the next problem
        const bool failed = *fail_flag != 0.0 || !(rho > 0.0) || !(rho < 1.79e308);
        if (t < N) x[prob * N + row] = failed ? NAN : l2;
        if (iters != nullptr && t == 0) iters[prob] = it_done;
    }
    // last workgroup out re-zeroes the work-list header (nothing to do when the list was empty)
    if (use_worklist && t == 0) worklist_release(ws, count, (int)gridDim.x);
}

template <int KIND, int N>
static hipError_t launch_block(const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    using G = BlockGeom<N>;
    auto kernel = fwd_dense_block_kernel<KIND, N>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return e;
    const long cap = 256L * (N == 32 ? 3 : 2) * 2; // persistent: 2 (N=64, LDS) or 3 (N=32, VGPRs) workgroups per CU, x2 for balance
    const unsigned grid = (unsigned)(a.B < (use_worklist ? 512L : cap) ? (a.B > 0 ? a.B : 1) : (use_worklist ? 512L : cap));
    return launch(kernel, dim3(grid), dim3(256), G::LDS_BYTES, s, a.P, a.q, a.l_n, a.mu, a.v, a.x, a.B, a.eps, a.mu_prox,
                       a.max_iter, a.adaptive, a.iters, a.ws, use_worklist ? 1 : 0);
}

bool fwd_dense_block_supported(int N) { return N == 32 || N == 64; }

hipError_t launch_fwd_dense_block(int kind, const FwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
#define DQQ_CASE(NN)                                                            \
    if (a.N == NN) {                                                            \
        switch (kind) {                                                         \
        case 0: return launch_block<0, NN>(a, use_worklist, s);                 \
        case 1: return launch_block<1, NN>(a, use_worklist, s);                 \
        case 2: return launch_block<2, NN>(a, use_worklist, s);                 \
        case 3: return launch_block<3, NN>(a, use_worklist, s);                 \
        default: return hipErrorInvalidValue;                                   \
        }                                                                       \
    }
    DQQ_CASE(64) DQQ_CASE(32)
#undef DQQ_CASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------- backward (QP), N = 32 / 64
// One workgroup per problem; the composition of pybindings.cpp:24-30 -> Solver::dualFromPrimalQP
// (Solver.cpp:125-134), Solver::solveDerivativesQP (:136-196), Solver::iterative_refinement (:15-44) and
// the gradient assembly of qcqp.py:48-51.
//
// The reference orders the unknowns (active..., inactive...) and solves A^T b = [0; grad_I] with
// A = blkdiag(diag(l_A), P_II) through the normal equations.  Here the system keeps the ORIGINAL index
// order with the active rows / columns of P masked out (A~[r][c] = P[r][c] if r and c are inactive,
// l_r if r = c is active, 0 otherwise): a symmetric permutation of the same block-diagonal matrix.  The
// 1x1 active blocks never mix with the rest in K = A~ A~^T + mu I, its Cholesky factor or its inverse
// (all cross terms are exact zeros), and the inactive indices keep their relative order, so the
// arithmetic on the inactive block is the reference's up to the summation order inside the tile
// products.  K, the factorisation and K^-1 run on the matrix cores as in the forward kernel; the
// refinement loop uses the WaveRows layout (K and K^-1 column blocks in registers).
template <int N>
__global__ __launch_bounds__(256, 2) void bwd_dense_block_qp_kernel(const double* __restrict__ P,
                                                                 const double* __restrict__ q,
                                                                 const double* __restrict__ x,
                                                                 const double* __restrict__ grad_x,
                                                                 double* __restrict__ grad_P, double* __restrict__ grad_q,
                                                                 long B, double dual_eps, int* __restrict__ ir_steps,
                                                                 int* __restrict__ ws, int use_worklist)
{
    using G = BlockGeom<N>;
    using WR = WaveRows<N>;
    constexpr int CW = WR::CW;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* X = smem;                 // K -> L -> K^-1
    double* Y = X + G::REGION;        // masked P -> LinvT
    double* part = Y + G::REGION;
    double* fail_flag = part + G::VEC;
    const int t = threadIdx.x;
    WR wr;
    wr.init(part, t);
    const int row = wr.row;
    const long count = use_worklist ? worklist_count(ws, N) : B;

    for (long w = blockIdx.x; w < count; w += gridDim.x) {
        const long prob = use_worklist ? worklist_entry(ws, N, B, w) : w;
        const double* Pg = P + prob * (long)(N * N);
        double m[CW]; // P[row][CW*wave ..]
#pragma unroll
        for (int k = 0; k < CW; ++k) m[k] = Pg[row * N + CW * wr.wave + k];
        const double xi = x[prob * N + row], gi = grad_x[prob * N + row], qi = q[prob * N + row];
        if (t == 0) *fail_flag = 0.0;
        // dualFromPrimalQP, Solver.cpp:125-134, and the active set of solveDerivativesQP, :139-147
        double gamma = -(wr.matvec(m, xi) + qi);
        if (xi > dual_eps) gamma = 0;
        const bool is_act = gamma < -kActiveEps;
        unsigned long long am = __ballot(is_act);
        if (N == 32) am &= 0xffffffffull;
        // A~ (see above) into Y, row-major
        for (int idx = t; idx < N * N; idx += G::T) {
            const int r = idx / N, c = idx % N;
            const bool masked = ((am >> r) | (am >> c)) & 1ull;
            Y[r * G::LD + c] = masked ? 0.0 : Pg[idx];
        }
        __syncthreads();
        if (t < N && is_act) Y[row * G::LD + row] = xi;                       // diag(l_A), :148-158
        __syncthreads();
        // A^T b (:19) with b = [0; grad_I]: sum over the inactive k of P[i][k] g[k]; zero on active rows
        double Ab = wr.matvec(m, is_act ? 0.0 : gi);
        if (is_act) Ab = 0.0;
        // K = A~ A~^T + mu_ir I (:20-21) -> X
        block_inverse_product<N, true>(Y, X, t);
        __syncthreads();
        if (t < N) X[row * G::LD + row] += kMuIr;
        __syncthreads();
        double kk[CW]; // K[row][CW*wave ..] (bitwise symmetric: read as columns, conflict-free)
#pragma unroll
        for (int k = 0; k < CW; ++k) kk[k] = X[(CW * wr.wave + k) * G::LD + row];
        __syncthreads();
        bool bad = false;
        block_cholesky_and_inverse<N>(X, Y, fail_flag, t, bad);              // :22-23
        block_inverse_product<N>(Y, X, t);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CW; ++k) m[k] = X[(CW * wr.wave + k) * G::LD + row]; // K^-1
        const double KinvAb = wr.matvec(m, Ab);                               // :27
        double xs = 0.0;
        IrControl ctl;
        ctl.init();
        int steps = 0;
        for (int it = 0; it < kIrMaxIter; ++it) {
            steps = it + 1;
            xs = kMuIr * wr.matvec(m, xs) + KinvAb;                           // :29
            const double d = wr.matvec(kk, xs) - Ab;                          // :30
            const double res = sqrt(WR::sum_rows(d * d));                     // :31
            if (ctl.update(res)) break;                                       // :32-41
        }
        const bool failed = *fail_flag != 0.0;
        const double dl = failed ? NAN : (is_act ? 0.0 : xs);                 // :187-191
        if (t < N && grad_q != nullptr) grad_q[prob * N + row] = -dl;        // qcqp.py:49
        if (grad_P != nullptr) {                                              // qcqp.py:48: -(dl l^T)
            double* Gp = grad_P + prob * (long)(N * N);
            const int lane = t & 63;
#pragma unroll
            for (int k = 0; k < N * N / 256; ++k) {
                double dr;
                if (N == 64) {
                    dr = lane_bcast(dl, wr.wave + 4 * k);
                } else {
                    const double d0 = lane_bcast(dl, 2 * (wr.wave + 4 * k)), d1 = lane_bcast(dl, 2 * (wr.wave + 4 * k) + 1);
                    dr = (lane >> 5) ? d1 : d0;
                }
                __builtin_nontemporal_store(-(dr * xi), Gp + (wr.wave + 4 * k) * 64 + lane);
            }
        }
        if (ir_steps != nullptr && t == 0) ir_steps[prob] = steps;
        __syncthreads(); // X / Y / the failure flag are free for the next problem
    }
    if (use_worklist && t == 0) worklist_release(ws, count, (int)gridDim.x);
}

template <int N>
static hipError_t launch_block_bwd(const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    using G = BlockGeom<N>;
    auto kernel = bwd_dense_block_qp_kernel<N>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES);
    if (e != hipSuccess) return e;
    const long cap = 256L * 2 * 4;
    const unsigned grid = (unsigned)(a.B < (use_worklist ? 512L : cap) ? (a.B > 0 ? a.B : 1) : (use_worklist ? 512L : cap));
    return launch(kernel, dim3(grid), dim3(256), G::LDS_BYTES, s, a.P, a.q, a.x, a.grad_x, a.grad_P, a.grad_q, a.B,
                       a.epsilon, a.ir_steps, a.ws, use_worklist ? 1 : 0);
}

bool bwd_dense_block_supported(int kind, int N) { return kind == 0 && (N == 32 || N == 64); }

hipError_t launch_bwd_dense_block(int kind, const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    if (kind != 0) return hipErrorInvalidValue;
    if (a.N == 64) return launch_block_bwd<64>(a, use_worklist, s);
    if (a.N == 32) return launch_block_bwd<32>(a, use_worklist, s);
    return hipErrorInvalidValue;
}

} // namespace dqq
