// bwd_block.hip -- general (dense P) backward for the systems too large for one wave (more than 64 unknowns:
// QCQP 42 < N <= 64, box QP 21 < N <= 32) and, on request, for QCQP N = 32 / box QP N = 16.  The system is
// padded with decoupled zero slots to MP = 48, 80 or 96 unknowns.  One 256-thread workgroup per problem,
// the normal-equations solve of Solver::iterative_refinement (Solver.cpp:15-44) on the f64 matrix cores
// (block_core.h), as in the QP backward of dense_block.hip.
//
// The derivative system keeps a FIXED number of unknowns M in the reference's order (bwd_small.hip): QCQP
// [gamma_0..gamma_{N/2-1}, dl_0..dl_{N-1}], box QP [lower_0, upper_0, lower_1, ..., dl_0..dl_{N-1}]; a
// multiplier the reference leaves out (inactive contact / bound) is a zero row and column, i.e. a decoupled
// K = mu_ir, x = 0.  A (M x M, row-major) is assembled in LDS, K = A A^T + mu I, blocked Cholesky + explicit
// inverse, then the refinement loop with one thread per unknown (K rows in registers, K^-1 rows read from
// LDS).  Dual recovery: per contact in closed form (QCQP, Solver.cpp:584-617); per coordinate 2x2 blocks of
// the Id2 system (box QP, :263-308, kkt_core.h BoxCoord) with the problem-wide residual norm.
// Tile products sum in a different order than the reference's loops (1e-16 relative): like every
// refinement loop of this path the 1-vs-3 step exit can differ from the oracle's on the problems where
// it is rounding noise; tests compare where the exits agree.
#include "block_core.h"
#include "kkt_core.h"
#include "launch.h"

namespace dqq {

template <int MP>
struct BlockSys {
    using G = BlockGeom<MP>;
    static constexpr int VEC = 8 * MP;
    static constexpr size_t LDS_BYTES = sizeof(double) * (2 * G::REGION + VEC + 2);
    static_assert(MP % 16 == 0 && MP <= 96, "at most 96 unknowns");
};

// MP: padded number of unknowns (multiple of 16); N (run time): N + N/2 <= MP (QCQP) or 3N <= MP (box QP).
// Slots beyond the problem's own unknowns are zero rows / columns of A (K = mu_ir there).
template <int KIND, int MP>
__global__ __launch_bounds__(256, 1) void bwd_block_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ aux0,
    const double* __restrict__ aux1, const double* __restrict__ x, const double* __restrict__ grad_x,
    double* __restrict__ grad_P, double* __restrict__ grad_q, double* __restrict__ gout0, double* __restrict__ gout1,
    double* __restrict__ gamma_out, double* __restrict__ dgamma_out, long B, int N, double dual_eps,
    int* __restrict__ ir_steps, int* __restrict__ ws, int use_worklist)
{
    using S = BlockSys<MP>;
    using G = typename S::G;
    constexpr int M = MP, LD = G::LD;
    const int NC = N / 2;
    const int MU = (KIND == 1) ? N + NC : 3 * N; // unknowns in use
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* X = smem;                 // K -> L -> K^-1
    double* Y = X + G::REGION;        // A -> LinvT
    double* vxs = Y + G::REGION;      // M
    double* vAb = vxs + M;            // M
    double* vb = vAb + M;             // M
    double* vd = vb + M;              // M
    double* vx = vd + M;              // N
    double* vw = vx + M;              // 2M: QCQP gamma (NC), S (NC), act (NC), P l + q (N); box gamma (2N), act (2N)
    double* fail_flag = vxs + S::VEC;
    const int t = threadIdx.x;
    const bool has = t < M;
    const int me = has ? t : 0;
    const long count = use_worklist ? worklist_count(ws, N) : B;

    for (long w = blockIdx.x; w < count; w += gridDim.x) {
        const long prob = use_worklist ? worklist_entry(ws, N, B, w) : w;
        const double* Pg = P + prob * (long)(N * N);
        __syncthreads();
        if (t < N) vx[t] = x[prob * N + t];
        if (t == 0) *fail_flag = 0.0;
        __syncthreads();
        int steps_dual = 0;
        if constexpr (KIND == 1) {
            double* vgam = vw;
            double* vS = vw + NC;
            double* vact = vw + 2 * NC;
            double* vplq = vw + 3 * NC;
            if (t < N) {
                double s = 0.0;
                for (int j = 0; j < N; ++j) s += Pg[t * N + j] * vx[j];
                vplq[t] = s + q[prob * N + t];                               // (P l + q)_i, :606
            }
            __syncthreads();
            if (t < NC) { // dualFromPrimalQCQP, :584-617, and the active set of solveDerivativesQCQP, :622-641
                const double ln = aux0[prob * NC + t], mc = aux1[prob * NC + t];
                const double r = ln * mc;                                    // pybindings.cpp:65
                const double xa = vx[2 * t], xb = vx[2 * t + 1];
                double gamma = 0.0;
                const double slack = r + -sqrt(xa * xa + xb * xb);
                if (!(slack > dual_eps || r < dual_eps)) {
                    const double ca = 2 * xa, cb = 2 * xb;
                    const double Gd = ca * ca + cb * cb;
                    const double rhs = ca * vplq[2 * t] + cb * vplq[2 * t + 1];
                    const double L = sqrt(Gd);
                    gamma = -((rhs / L) / L);
                }
                double Sc = -(r * r);
                Sc = Sc + (xa * xa + xb * xb);
                vgam[t] = gamma;
                vS[t] = Sc;
                vact[t] = (Sc > -kActiveEps && r > kActiveEps) ? 1.0 : 0.0;
            }
            __syncthreads();
            // A = [[diag(S), diag(gamma) C^T],[C, P + blkdiag(2 gamma_i I2)]] (:643-657), inactive contacts zeroed
            for (int idx = t; idx < M * M; idx += 256) {
                const int r = idx / M, c = idx % M;
                double val = 0.0;
                if (r >= MU || c >= MU) {
                    // padding slot
                } else if (r < NC) {
                    if (vact[r] != 0.0) {
                        if (c == r) val = vS[r];
                        else if (c == NC + 2 * r || c == NC + 2 * r + 1) val = vgam[r] * (2 * vx[c - NC]);
                    }
                } else {
                    const int i = r - NC;
                    if (c < NC) val = (c == i / 2 && vact[c] != 0.0) ? 2 * vx[i] : 0.0;
                    else val = ((c - NC == i) ? 2 * vgam[i / 2] : 0.0) + Pg[i * N + (c - NC)];
                }
                Y[r * LD + c] = val;
            }
            if (has) vb[t] = (t < NC || t >= MU) ? 0.0 : grad_x[prob * N + (t - NC)];    // :659-667
        } else {
            double* vgam = vw;          // 2N, slots (2i lower, 2i+1 upper)
            double* vact = vw + 2 * N;  // 2N
            // ---- dualFromPrimalBoxQP (:263-308): one coordinate per thread, 2x2 blocks of the Id2 system
            BoxCoord bc;
            {
                double rhs = 0.0, xi = 0.0, lo = 0.0, hi = 0.0;
                if (t < N) {
                    for (int j = 0; j < N; ++j) rhs += (-Pg[t * N + j]) * vx[j];
                    rhs = rhs - q[prob * N + t];
                    xi = vx[t];
                    lo = aux0[prob * N + t];
                    hi = aux1[prob * N + t];
                }
                bc.setup_dual_rhs(rhs, xi, lo, hi, dual_eps);
            }
            IrControl dctl;
            dctl.init();
            for (int it = 0; it < kIrMaxIter; ++it) {
                double dsq[3];
                bc.step_dual(dsq);
                if (t < N) { vd[2 * t] = dsq[0]; vd[2 * t + 1] = dsq[1]; }
                __syncthreads();
                double ss = 0.0;
                for (int i = 0; i < 2 * N; ++i) ss += vd[i];
                __syncthreads();
                steps_dual = it + 1;
                if (dctl.update(sqrt(ss))) break;
            }
            bc.finish_dual();
            if (t < N) {
                vgam[2 * t] = bc.gamma_lo; vgam[2 * t + 1] = bc.gamma_hi;
                vact[2 * t] = bc.aL ? 1.0 : 0.0; vact[2 * t + 1] = bc.aU ? 1.0 : 0.0;
            }
            __syncthreads();
            // A = [[0, B],[Id2, P]] (:341-350), B.row(j) = gamma_j * Id2.col(j)^T; missing multipliers zeroed
            for (int idx = t; idx < M * M; idx += 256) {
                const int r = idx / M, c = idx % M;
                double val = 0.0;
                if (r >= MU || c >= MU) {
                    // padding slot
                } else if (r < 2 * N) {
                    if (c == 2 * N + r / 2 && vact[r] != 0.0) val = vgam[r] * ((r & 1) ? 1.0 : -1.0);
                } else {
                    const int i = r - 2 * N;
                    if (c < 2 * N) val = (c / 2 == i && vact[c] != 0.0) ? ((c & 1) ? 1.0 : -1.0) : 0.0;
                    else val = Pg[i * N + (c - 2 * N)];
                }
                Y[r * LD + c] = val;
            }
            if (has) vb[t] = (t < 2 * N || t >= MU) ? 0.0 : grad_x[prob * N + (t - 2 * N)]; // :352-360
        }
        __syncthreads();
        // A^T_t b = A b (:19); iterative_refinement is handed A^T (transposeInPlace, :351 / :658)
        double Ab = 0.0;
        for (int k = 0; k < M; ++k) Ab += Y[me * LD + k] * vb[k];
        if (has) vAb[t] = Ab;
        // K = A A^T + mu_ir I (:20-21) -> X
        block_inverse_product<M, true>(Y, X, t);
        __syncthreads();
        if (has) X[t * LD + t] += kMuIr;
        __syncthreads();
        double Kr[M];
#pragma unroll
        for (int j = 0; j < M; ++j) Kr[j] = X[me * LD + j];
        __syncthreads();
        bool bad = false;
        block_cholesky_and_inverse<M>(X, Y, fail_flag, t, bad);              // :22-23
        block_inverse_product<M>(Y, X, t);
        __syncthreads();
        double KinvAb = 0.0;                                                 // :27
        for (int j = 0; j < M; ++j) KinvAb += X[me * LD + j] * vAb[j];
        double xs = 0.0;
        IrControl ctl;
        ctl.init();
        int steps = 0;
        for (int it = 0; it < kIrMaxIter; ++it) {
            steps = it + 1;
            if (has) vxs[t] = xs;
            __syncthreads();
            double tmp = 0.0;                                                // :29
            for (int j = 0; j < M; ++j) tmp += X[me * LD + j] * vxs[j];
            xs = kMuIr * tmp + KinvAb;
            __syncthreads();
            if (has) vxs[t] = xs;
            __syncthreads();
            double d = 0.0;                                                  // :30
#pragma unroll
            for (int j = 0; j < M; ++j) d += Kr[j] * vxs[j];
            d = d - Ab;
            if (has) vd[t] = d;
            __syncthreads();
            double ss = 0.0;                                                 // :31
            for (int i = 0; i < M; ++i) ss += vd[i] * vd[i];
            __syncthreads();
            if (ctl.update(sqrt(ss))) break;                                 // :32-41
        }
        // vxs holds the solution: multipliers' slots first, then dl
        const bool failed = *fail_flag != 0.0;
        const int L0 = (KIND == 1) ? NC : 2 * N; // first dl slot
        if constexpr (KIND == 1) {
            if (t < NC) {
                const double ln = aux0[prob * NC + t], mc = aux1[prob * NC + t];
                const double gamma = vw[t];
                const double dg = failed ? NAN : ((vw[2 * NC + t] != 0.0) ? vxs[t] : 0.0);  // :671-674
                if (gout0 != nullptr) gout0[prob * NC + t] = QcqpContact::e2(gamma, ln, mc) * dg;  // grad_l_n
                if (gout1 != nullptr) gout1[prob * NC + t] = QcqpContact::e1(gamma, ln, mc) * dg;  // grad_mu
                if (gamma_out != nullptr) gamma_out[prob * NC + t] = gamma;
                if (dgamma_out != nullptr) dgamma_out[prob * NC + t] = dg;
            }
            if (ir_steps != nullptr && t == 0) ir_steps[prob] = steps;
        } else {
            if (t < N) {
                const double glo = vw[2 * t], ghi = vw[2 * t + 1];
                const double dlo = failed ? NAN : ((vw[2 * N + 2 * t] != 0.0) ? vxs[2 * t] : 0.0);       // :363-366
                const double dhi = failed ? NAN : ((vw[2 * N + 2 * t + 1] != 0.0) ? vxs[2 * t + 1] : 0.0);
                if (gout0 != nullptr) gout0[prob * N + t] = -(dlo * glo);   // grad_l_min
                if (gout1 != nullptr) gout1[prob * N + t] = dhi * ghi;      // grad_l_max
                if (gamma_out != nullptr) { gamma_out[prob * 2 * N + t] = glo; gamma_out[prob * 2 * N + N + t] = ghi; }
                if (dgamma_out != nullptr) { dgamma_out[prob * 2 * N + t] = dlo; dgamma_out[prob * 2 * N + N + t] = dhi; }
            }
            if (ir_steps != nullptr && t == 0) { ir_steps[2 * prob] = steps_dual; ir_steps[2 * prob + 1] = steps; }
        }
        if (t < N && grad_q != nullptr) grad_q[prob * N + t] = failed ? NAN : -vxs[L0 + t];
        if (grad_P != nullptr) {
            double* Gp = grad_P + prob * (long)(N * N);
            for (int idx = t; idx < N * N; idx += 256) {
                const double v = failed ? NAN : -(vxs[L0 + idx / N] * vx[idx % N]);
                __builtin_nontemporal_store(v, Gp + idx);
            }
        }
    }
    if (use_worklist && t == 0) worklist_release(ws, count, (int)gridDim.x);
}

template <int KIND, int MP>
static hipError_t launch_bwd_block_sys(const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    using S = BlockSys<MP>;
    auto kernel = bwd_block_kernel<KIND, MP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES);
    if (e != hipSuccess) return e;
    const long cap = 256L * 8;
    const unsigned grid = (unsigned)(a.B < (use_worklist ? 512L : cap) ? (a.B > 0 ? a.B : 1) : (use_worklist ? 512L : cap));
    return launch(kernel, dim3(grid), dim3(256), S::LDS_BYTES, s, a.P, a.q, a.l_n, a.mu, a.x, a.grad_x, a.grad_P,
                       a.grad_q, a.grad_l_n, a.grad_mu, a.gamma, a.dgamma, a.B, a.N, a.epsilon, a.ir_steps, a.ws,
                       use_worklist ? 1 : 0);
}

// unknowns of the derivative system
static int block_sys_unknowns(int kind, int N) { return kind == kKindQCQP ? N + N / 2 : 3 * N; }

bool bwd_block_sys_supported(int kind, int N)
{
    if (kind != kKindQCQP && kind != kKindBox) return false;
    if (kind == kKindQCQP && (N % 2) != 0) return false;
    const int m = block_sys_unknowns(kind, N);
    return N >= 1 && m <= 96 && m >= 33; // below that the wave kernel is the faster faithful choice anyway
}

hipError_t launch_bwd_block_sys(int kind, const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    if (!bwd_block_sys_supported(kind, a.N)) return hipErrorInvalidValue;
    const int m = block_sys_unknowns(kind, a.N);
    if (kind == kKindQCQP) {
        if (m <= 48) return launch_bwd_block_sys<1, 48>(a, use_worklist, s);
        if (m <= 80) return launch_bwd_block_sys<1, 80>(a, use_worklist, s);
        return launch_bwd_block_sys<1, 96>(a, use_worklist, s);
    }
    if (m <= 48) return launch_bwd_block_sys<2, 48>(a, use_worklist, s);
    if (m <= 80) return launch_bwd_block_sys<2, 80>(a, use_worklist, s);
    return launch_bwd_block_sys<2, 96>(a, use_worklist, s);
}

} // namespace dqq
