// bwd_wave_qcqp.hip -- QCQP backward for a general (dense) P, 16 < N <= 32: ONE WAVE per problem, the whole
// (N/2 + N)-unknown system in registers (wave_tile.h / wave_chol.h), like the QP backward of dense_wave64.hip.
//
// The composition of pybindings.cpp:62-71 -> Solver::dualFromPrimalQCQP (Solver.cpp:584-617), getE12QCQP (:683-691),
// solveDerivativesQCQP (:619-681), iterative_refinement (:15-44) and the gradient assembly of qcqp.py:173-180.
//
// The reference orders the unknowns (active contacts..., coordinates...).  Here the layout is FIXED: 48 slots =
// 3 x 3 tiles of 16; slots 0..15 are the contacts (contact c in slot c; slots of inactive contacts and beyond N/2 are
// empty: zero row and column, so K has mu on their diagonal, a zero right-hand side, and they never mix with the rest),
// slots 16..47 the coordinates (padded with the identity beyond N).  That is a symmetric permutation of the
// reference's matrix plus decoupled slots; the arithmetic on the live block differs from the reference's by the
// summation order of the tile products only.
//     A[c][c] = S_c,  A[c][16+i] = gamma_c 2 l_i (i in contact c)        S_c = |l_(c)|^2 - r_c^2      (:643-650)
//     A[16+i][c] = 2 l_i (i in contact c),  A[16+i][16+j] = P[i][j] + [i == j] 2 gamma_(i/2)            (:651-657)
//     solve  A^T b = [0; grad_l]  in the Tikhonov sense of iterative_refinement(A^T, .):  K = A A^T + mu I, rhs A dd.
// K (upper tiles) is factored and inverted in place by block Cholesky (wave_chol.h), a second copy serves the
// refinement residual.  The alternative is the LDS wave kernel in the reference's summation order (0.9 ms per 4096
// problems at N = 32), selected per call with DQQ_F_REFERENCE_ORDER.
#include "kkt_core.h"
#include "launch.h"
#include "wave_chol.h"

namespace dqq {

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void bwd_wave_qcqp_kernel(
    const double* __restrict__ P, const double* __restrict__ q, const double* __restrict__ l_n,
    const double* __restrict__ mu_c, const double* __restrict__ x, const double* __restrict__ grad_x,
    double* __restrict__ grad_P, double* __restrict__ grad_q, double* __restrict__ grad_l_n,
    double* __restrict__ grad_mu, double* __restrict__ gamma_out, double* __restrict__ dgamma_out, long B, int N,
    double dual_eps, int* __restrict__ ir_steps, int* __restrict__ ws, int use_worklist)
{
    constexpr int NT = 3;
    __shared__ __attribute__((aligned(16))) double s_trb[16 * kTrLd]; // tile transposes (one wave per workgroup)
    WorkClaim claim; // (launch.h: direct mode, or dynamic pick-up from the work-list)
    claim.open(ws, use_worklist, N, B);
    const int nc = N / 2;
    for (long w = blockIdx.x;; w += gridDim.x) {
        const long prob = claim.next(ws, B, w); // wave-uniform (SGPRs: P's addressing uses a scalar base)
        if (prob < 0) break;
        claim.ahead_issue(ws); // (work-list: the ticket for this wave's next problem travels while this one is solved)
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane)); // nothing lane-derived is hoisted out of the problem loop (dense_wave64.hip)
        const int g = lane >> 4, n = lane & 15;
        const int xsrc = 4 * n + g;
        const double* Pg = P + prob * (long)(N * N);
        // vectors: one entry per lane, lane = slot: 0..15 contacts, 16..47 coordinates
        const int ci = lane - 16;                        // coordinate of this lane
        const bool is_coord = lane >= 16 && ci < N, is_contact = lane < nc;
        const double xi = is_coord ? x[prob * N + ci] : 0.0, gi = is_coord ? grad_x[prob * N + ci] : 0.0;
        const double qi = is_coord ? q[prob * N + ci] : 0.0;
        const double ln = is_contact ? l_n[prob * nc + lane] : 1.0, mc = is_contact ? mu_c[prob * nc + lane] : 1.0;

        WaveChol<NT> C;
        v4d Kc[NT][NT];
        double Ab, gamma;
        bool is_act;
        {
            WaveTile<NT> A; // A.G[tk][ta][r] of lane (g,n) = A[16 ta + n][16 tk + 4 r + g]   (tile layout of A^T)
            const v4d zero = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < NT; ++t) { A.G[0][t] = zero; A.G[t][0] = zero; }
            // the coordinate block: P[i][j], i = 16 (ta-1) + n (row), j = 16 (tk-1) + 4 r + g (column); identity beyond N
            const unsigned lo = n * N + g;
#pragma unroll
            for (int tk = 1; tk < NT; ++tk)
#pragma unroll
                for (int ta = 1; ta < NT; ++ta)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * (ta - 1) + n, j = 16 * (tk - 1) + 4 * r + g;
                        A.G[tk][ta][r] = (i < N && j < N) ? (Pg + ((16 * (ta - 1)) * N + 16 * (tk - 1) + 4 * r))[lo]
                                                          : ((i == j) ? 1.0 : 0.0);
                    }
            // dualFromPrimalQCQP, Solver.cpp:584-617: P l + q per coordinate, then one contact per lane
            const double plq = A.matvec(xi, xsrc) + qi;              // contact tiles are still zero
            const int c2 = 16 + 2 * (lane & 15);                     // slots of this contact's two coordinates
            const double xa = lane_gather(xi, c2), xb = lane_gather(xi, c2 + 1);
            const double pa = lane_gather(plq, c2), pb = lane_gather(plq, c2 + 1);
            const double rr = ln * mc;                               // pybindings.cpp:65
            gamma = 0.0;
            {
                const double slack = rr - sqrt(xa * xa + xb * xb);
                if (is_contact && !(slack > dual_eps || rr < dual_eps)) {
                    const double ca = 2 * xa, cb = 2 * xb;
                    const double G2 = ca * ca + cb * cb;
                    const double rhs = ca * pa + cb * pb;
                    const double L = sqrt(G2);
                    gamma = -((rhs / L) / L);                        // (A~^T A~)^-1 A~^T (P l + q), :605-615
                }
            }
            const double S = (xa * xa + xb * xb) - rr * rr;          // :622-629
            is_act = is_contact && S > -kActiveEps && rr > kActiveEps; // :637-641
            const unsigned am = (unsigned)__ballot(is_act) & 0xffffu;
            // ---- the contact rows / columns (:643-657)
            const double S_n = lane_gather(S, n), gam_n = lane_gather(gamma, n);
            const bool act_n = (am >> n) & 1u;
#pragma unroll
            for (int r = 0; r < 4; ++r) A.G[0][0][r] = (act_n && 4 * r + g == n) ? S_n : 0.0;   // A[c][c] = S_c
#pragma unroll
            for (int t = 1; t < NT; ++t) {
                const double x_row = lane_gather(xi, 16 * t + n);   // l_i of row i = 16 (t-1) + n
                const double gam_row = lane_gather(gamma, (16 * (t - 1) + n) >> 1); // gamma of that row's contact
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * (t - 1) + 4 * r + g;          // column coordinate of G[t][.][r]
                    const double x_col = lane_gather(xi, 16 + j);
                    // A[contact n][16 + j] = gamma_n 2 l_j, j in contact n
                    A.G[t][0][r] = (act_n && (j >> 1) == n) ? gam_n * (2 * x_col) : 0.0;
                    // A[16 + i][contact 4r+g] = 2 l_i, i = 16 (t-1) + n in contact 4r+g
                    const int cc = 4 * r + g, i = 16 * (t - 1) + n;
                    A.G[0][t][r] = (((am >> cc) & 1u) && (i >> 1) == cc) ? 2 * x_row : 0.0;
                    // the diagonal of the coordinate block: + 2 gamma_(i/2)
                    if (4 * r + g == n && i < N) A.G[t][t][r] += 2 * gam_row;
                }
            }
            Ab = A.matvec(is_coord ? gi : 0.0, xsrc);                // rhs of the normal equations: A [0; grad_l] (:19)
            const bool on_diag = (n & 3) == g;                       // K = A A^T + mu_ir I (:20-21), upper tiles
#pragma unroll
            for (int ta = 0; ta < NT; ++ta)
#pragma unroll
                for (int tb = ta; tb < NT; ++tb) {
                    v4d acc = zero;
#pragma unroll
                    for (int tk = 0; tk < NT; ++tk) acc = tile_xty(acc, A.G[tk][ta], A.G[tk][tb]);
                    if (ta == tb) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] += (on_diag && (n >> 2) == r) ? kMuIr : 0.0;
                    }
                    C.U[ta][tb] = acc;
                    Kc[ta][tb] = acc;
                }
        }
        bool bad = false;
        claim.ahead_entry(ws, B);
        C.factor(lane, bad);                                                  // :22
        C.invert_in_place(lane, s_trb);                                              // :23
        const double KinvAb = sym_upper_matvec<NT>(C.U, Ab, xsrc, lane, s_trb);      // :27
        double xs = 0.0;
        IrControl ctl;
        ctl.init();
        int steps = 0;
        for (int it = 0; it < kIrMaxIter; ++it) {
            steps = it + 1;
            xs = it == 0 ? KinvAb : KinvAb + kMuIr * sym_upper_matvec<NT>(C.U, xs, xsrc, lane, s_trb); // :29 (first body: x = 0)
            const double d = sym_upper_matvec<NT>(Kc, xs, xsrc, lane, s_trb) - Ab;   // :30
            const double res = sqrt(wave_sum64(d * d));                       // :31
            if (ctl.update(res)) break;                                       // :32-41
        }
        const double b = bad ? NAN : xs;                                      // blgamma, :670-679
        if (is_contact) {
            const double dg = is_act ? b : 0.0;
            if (grad_l_n != nullptr) grad_l_n[prob * nc + lane] = QcqpContact::e2(gamma, ln, mc) * dg;   // qcqp.py:178
            if (grad_mu != nullptr) grad_mu[prob * nc + lane] = QcqpContact::e1(gamma, ln, mc) * dg;     // qcqp.py:180
            if (gamma_out != nullptr) gamma_out[prob * nc + lane] = gamma;
            if (dgamma_out != nullptr) dgamma_out[prob * nc + lane] = dg;
        }
        claim.ahead_done(ws, B);
        if (is_coord && grad_q != nullptr) grad_q[prob * N + ci] = -b;       // qcqp.py:176
        if (grad_P != nullptr) {                                              // qcqp.py:174: -(dl l^T)
            double* Gp = grad_P + prob * (long)(N * N);
            for (int k = 0; k < N; ++k) {
                const double v = -(lane_bcast(b, 16 + k) * xi);
                if (is_coord) __builtin_nontemporal_store(v, Gp + k * N + ci);
            }
        }
        if (ir_steps != nullptr && lane == 0) ir_steps[prob] = steps;
    }
    if (use_worklist && threadIdx.x == 0) worklist_release(ws, claim.count, (int)gridDim.x);
}

bool bwd_wave_qcqp_supported(int kind, int N) { return kind == kKindQCQP && N > 16 && N <= 32 && (N & 1) == 0; }

hipError_t launch_bwd_wave_qcqp(const BwdArgs& a, bool use_worklist, hipStream_t s)
{
    if (a.B == 0) return hipSuccess;
    if (!bwd_wave_qcqp_supported(kKindQCQP, a.N)) return hipErrorInvalidValue;
    const long cap = 1L << 22;
    const unsigned grid = (unsigned)(a.B < (use_worklist ? 2048L : cap) ? (a.B > 0 ? a.B : 1) : (use_worklist ? 2048L : cap));
    return launch(bwd_wave_qcqp_kernel, dim3(grid), dim3(64), 0, s, a.P, a.q, a.l_n, a.mu, a.x, a.grad_x, a.grad_P, a.grad_q,
                  a.grad_l_n, a.grad_mu, a.gamma, a.dgamma, a.B, a.N, a.epsilon, a.ir_steps, a.ws, use_worklist ? 1 : 0);
}

} // namespace dqq
