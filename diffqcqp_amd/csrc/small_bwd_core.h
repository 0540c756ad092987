// small_bwd_core.h -- general (dense P) backward for small N (even, <= 16: up to 8 contacts): a dense Delassus
// matrix is what a real contact problem presents, and its backward is three to five times the work of its
// forward.
//
// Same composition as dense_core.h / the reference (pybindings.cpp:24-30, 39-45, 62-71 ->
// Solver::dualFromPrimal*, solveDerivatives*, iterative_refinement, Solver.cpp:15-44, 125-196, 263-371,
// 584-691, and the gradient assembly of qcqp.py), but with STATICALLY SIZED systems so that every loop
// unrolls and every operand of the O(M^3) parts is a register or a broadcast LDS read:
//
//   * the derivative system always has M = N (QP), N + N/2 (QCQP) or 3N (box QP) unknowns, in the
//     reference's order -- multipliers first (contact by contact / coordinate by coordinate, lower before
//     upper), then dl.  A multiplier the reference would leave out (inactive contact / bound) keeps its
//     slot with a zero row and column: it decouples into K = mu_ir, A^T b = 0, x = 0, residual 0 and
//     contributes exact zeros to every sum of the live unknowns, whose relative order is the reference's,
//     so the arithmetic on them is the reference's operation for operation.  The QP keeps the ORIGINAL
//     coordinate order with active rows / columns masked (a symmetric permutation of the reference's
//     [active, inactive] order with the same property).
//   * a team of T = M lanes owns a problem (64/T problems per wave); lane i owns unknown i: its
//     column of the matrix handed to iterative_refinement, row i of K = A^T A + mu I, row i of the Cholesky
//     factor, column i (then row i) of K^-1 -- all in registers.  What other lanes need is published to
//     the team's LDS slice once and read back as broadcasts.
//
// Sums run sequentially in index order and FP contraction is off (build.py): like dense_core.h this is
// the reference's dense arithmetic.  Work-list mode as in dense.hip.
//
// This header holds the per-problem routine (a team of lanes): bwd_small.hip wraps it into the stand-alone kernel
// of the general path, bwd_diag.hip calls it for the non-diagonal tiles its fused form meets.
#pragma once

#include "dense_core.h"

namespace dqq {

template <int KIND, int N>
struct SmallSys {
    static constexpr int NC = N / 2;
    static constexpr int M = (KIND == 0) ? N : (KIND == 1 ? N + NC : 3 * N); // unknowns of the derivative system
    // team width: exactly the number of unknowns (round 4; it was the next power of two -- 12 of 16 lanes worked on the QCQP at
    // N = 8, 9 of 16 at N = 6).  Nothing needs a power of two: what a lane needs from its team goes through the team's LDS slice
    // and team_ballot shifts by team * T.  64 / T teams per wave, the 64 mod T lanes left over idle.  Dense P, B = 65536, backward, us:
    // QCQP N = 8 124 -> 113, N = 6 79 -> 54, N = 4 26 -> 24; QP N = 8 36.4 -> 35.6, N = 6 20 -> 18.5, N = 4 14.7 -> 11.5.
#if defined(DQQ_SMALL_T_POW2)
    static constexpr int T = (M <= 8) ? 8 : (M <= 16 ? 16 : 32);              // developer A/B
#else
    static constexpr int T = (M == 3) ? 4 : M;                                 // (three-lane teams measured slower than four)
#endif
    static constexpr int LDA = (M + 1) & ~1;                                   // even: rows start 16-byte aligned
    static constexpr int LDS_DOUBLES = 2 * M * LDA + 8 * M;                    // [A_t, then K^-1], [L], vectors
#if defined(DQQ_SMALL_WPB)
    static constexpr int WPB = DQQ_SMALL_WPB;   // developer A/B (tools/ab_build.sh)
#else
    static constexpr int WPB = (LDS_DOUBLES * (64 / T) * 8 * 4 <= 64 * 1024) ? 4 : 2;   // two workgroups per CU by registers
#endif
};

// Solver::iterative_refinement (Solver.cpp:15-44) for a ROWS x M matrix A_t, by a team.  a[k] = A_t[k][lane]
// (this lane's column), vb = right-hand side (ROWS, LDS).  Returns x[lane].  Scratch (team-private LDS):
// AtL (M x LDA, A_t rows then K^-1), Lm (M x LDA), v0..v2 (M each).
template <int M, int ROWS, int LDA>
static DQQ_D double team_ir(const double (&a)[ROWS], const double* vb, double* AtL, double* Lm, double* v0, double* v1,
                            double* v2, int lane, int& steps)
{
#pragma clang fp contract(off)
    const bool act = lane < M;
    const int me = act ? lane : 0;
    if (act) {
#pragma unroll
        for (int k = 0; k < ROWS; ++k) AtL[k * LDA + lane] = a[k];
    }
    DQQ_SYNC();
    double Ab = 0.0;                                                        // A^T b, :19
#pragma unroll
    for (int k = 0; k < ROWS; ++k) Ab += a[k] * vb[k];
    double Kr[M];                                                           // row `lane` of A^T A, :20
#pragma unroll
    for (int j = 0; j < M; ++j) Kr[j] = 0.0;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
#pragma unroll
        for (int j = 0; j < M; ++j) Kr[j] += a[k] * AtL[k * LDA + j];
    }
    double kd = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        if (j == lane) { Kr[j] += kMuIr; kd = Kr[j]; }                      // :21
    }
    double* vKd = v0;
    double* vAb = v1;
    if (act) { vKd[lane] = kd; vAb[lane] = Ab; }
    DQQ_SYNC();
    // llt(), :23 -- left-looking; row k of L is complete (and published) before step k reads it
    double Lr[M];
#pragma unroll
    for (int k = 0; k < M; ++k) {
        double s = 0.0, t = 0.0;
#pragma unroll
        for (int j = 0; j < k; ++j) {
            const double lkj = Lm[k * LDA + j];
            s += lkj * lkj;
            t += Lr[j] * lkj;
        }
        double xk = vKd[k] - s;
        xk = sqrt(xk);
        const double lik = (Kr[k] - t) / xk;
        Lr[k] = (lane == k) ? xk : lik;
        if (act && lane >= k) Lm[lane * LDA + k] = Lr[k];
        DQQ_SYNC();
    }
    // solveInPlace(Identity), :22-23: lane = column c of the inverse
    double y[M];
#pragma unroll
    for (int i = 0; i < M; ++i) {
        double t = (i == lane) ? 1.0 : 0.0;
#pragma unroll
        for (int j = 0; j < i; ++j) t -= Lm[i * LDA + j] * y[j];
        y[i] = t / Lm[i * LDA + i];
    }
#pragma unroll
    for (int i = M - 1; i >= 0; --i) {
        double t = y[i];
#pragma unroll
        for (int j = i + 1; j < M; ++j) t -= Lm[j * LDA + i] * y[j];
        y[i] = t / Lm[i * LDA + i];
    }
    if (act) {
#pragma unroll
        for (int i = 0; i < M; ++i) AtL[i * LDA + lane] = y[i];               // K^-1[i][c]
    }
    DQQ_SYNC();
    double Ki[M];                                                           // row `lane` of K^-1
#pragma unroll
    for (int j = 0; j < M; ++j) Ki[j] = AtL[me * LDA + j];
    double KinvAb = 0.0;                                                    // :27
#pragma unroll
    for (int j = 0; j < M; ++j) KinvAb += Ki[j] * vAb[j];
    double* vxs = v0;
    double* vd = v2;
    double xs = 0.0;
    IrControl ctl;
    ctl.init();
    steps = 0;
    DQQ_SYNC();
    for (int it = 0; it < kIrMaxIter; ++it) {
        steps = it + 1;
        if (act) vxs[lane] = xs;
        DQQ_SYNC();
        double tmp = 0.0;                                                   // :29
#pragma unroll
        for (int j = 0; j < M; ++j) tmp += Ki[j] * vxs[j];
        xs = kMuIr * tmp + KinvAb;
        DQQ_SYNC();
        if (act) vxs[lane] = xs;
        DQQ_SYNC();
        double d = 0.0;                                                     // :30
#pragma unroll
        for (int j = 0; j < M; ++j) d += Kr[j] * vxs[j];
        d = d - Ab;
        if (act) vd[lane] = d;
        DQQ_SYNC();
        double ss = 0.0;                                                    // :31
#pragma unroll
        for (int i = 0; i < M; ++i) ss += vd[i] * vd[i];
        const double res = sqrt(ss);
        DQQ_SYNC();
        if (ctl.update(res)) break;                                         // :32-41
    }
    return xs;
}

template <int KIND, int N>
static DQQ_D void small_bwd_problem(const double* __restrict__ P, const double* __restrict__ q,
                                    const double* __restrict__ aux0, const double* __restrict__ aux1,
                                    const double* __restrict__ x, const double* __restrict__ grad_x,
                                    double* __restrict__ grad_P, double* __restrict__ grad_q,
                                    double* __restrict__ gout0, double* __restrict__ gout1,
                                    double* __restrict__ gamma_out, double* __restrict__ dgamma_out,
                                    int* __restrict__ ir_steps, long prob, double dual_eps, double* smem, int lane)
{
#pragma clang fp contract(off)
    using S = SmallSys<KIND, N>;
    constexpr int M = S::M, T = S::T, LDA = S::LDA, NC = S::NC;
    double* AtL = smem;
    double* Lm = AtL + M * LDA;
    double* v0 = Lm + M * LDA;      // M
    double* v1 = v0 + M;            // M
    double* v2 = v1 + M;            // M
    double* vb = v2 + M;            // M: right-hand side
    double* vx = vb + M;            // N
    double* vw = vx + M;            // up to 2N: gamma / (P x + q)
    const double* Pg = P + prob * (long)(N * N);
    int steps = 0;

    if constexpr (KIND == 0) {
        // lane i (< N) = coordinate i, original order
        const bool actn = lane < N;
        const int i = actn ? lane : 0;
        double Prow[N];
#pragma unroll
        for (int j = 0; j < N; ++j) Prow[j] = Pg[i * N + j];
        const double xi = x[prob * N + i], gi = grad_x[prob * N + i], qi = q[prob * N + i];
        if (actn) vx[lane] = xi;
        DQQ_SYNC();
        double gamma = 0.0;                                                 // dualFromPrimalQP, :125-134
#pragma unroll
        for (int j = 0; j < N; ++j) gamma += Prow[j] * vx[j];
        gamma = -(gamma + qi);
        if (xi > dual_eps) gamma = 0;
        const bool is_act = actn && gamma < -kActiveEps;                    // :139-147
        const unsigned long long am = team_ballot<T>(is_act);
        // row i of A = [[diag(l_A), 0],[0, P_II]] in the original order = column i of the matrix handed to
        // iterative_refinement (A.transposeInPlace(), :174)
        double a[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const bool ka = (am >> k) & 1ull;
            a[k] = is_act ? ((k == i) ? xi : 0.0) : (ka ? 0.0 : Prow[k]);
        }
        if (actn) vb[lane] = is_act ? 0.0 : gi;                             // :175-184
        DQQ_SYNC();
        const double xs = team_ir<M, M, LDA>(a, vb, AtL, Lm, v0, v1, v2, lane, steps);
        const double dl = is_act ? 0.0 : xs;                                // :187-191
        if (actn) {
            if (grad_q != nullptr) grad_q[prob * N + lane] = -dl;
            if (grad_P != nullptr) {
                double* Gp = grad_P + prob * (long)(N * N) + lane * N;
#pragma unroll
                for (int j = 0; j < N; ++j) Gp[j] = -(dl * vx[j]);
            }
        }
        if (ir_steps != nullptr && lane == 0) ir_steps[prob] = steps;
    } else if constexpr (KIND == 1) {
        // slots: c < NC = gamma of contact c, NC + i = dl_i
        const bool is_c = lane < NC, is_l = lane >= NC && lane < M;
        const int c = is_c ? lane : 0, i = is_l ? lane - NC : 0;
        double Prow[N];
#pragma unroll
        for (int j = 0; j < N; ++j) Prow[j] = Pg[i * N + j];
        const double xi = x[prob * N + i], gi = grad_x[prob * N + i], qi = q[prob * N + i];
        if (is_l) vx[i] = xi;
        DQQ_SYNC();
        double plq = 0.0;                                                   // (P l + q)_i, :606
#pragma unroll
        for (int j = 0; j < N; ++j) plq += Prow[j] * vx[j];
        plq = plq + qi;
        if (is_l) vw[N + i] = plq;
        DQQ_SYNC();
        // dualFromPrimalQCQP, :584-617, and the active set of solveDerivativesQCQP, :622-641 (contact lanes)
        const double ln = aux0[prob * NC + c], mc = aux1[prob * NC + c];
        const double r = ln * mc;                                           // pybindings.cpp:65
        const double xa = vx[2 * c], xb = vx[2 * c + 1];
        double gamma = 0.0;
        {
            const double slack = r + -sqrt(xa * xa + xb * xb);
            if (!(slack > dual_eps || r < dual_eps)) {
                const double ca = 2 * xa, cb = 2 * xb;
                const double G = ca * ca + cb * cb;
                const double rhs = ca * vw[N + 2 * c] + cb * vw[N + 2 * c + 1];
                const double L = sqrt(G);
                gamma = -((rhs / L) / L);
            }
        }
        double Sc = -(r * r);
        Sc = Sc + (xa * xa + xb * xb);
        const bool c_act = is_c && Sc > -kActiveEps && r > kActiveEps;
        if (is_c) vw[c] = gamma;
        const unsigned long long am = team_ballot<T>(c_act);
        DQQ_SYNC();
        // row `slot` of A = [[diag(S), diag(gamma) C^T],[C, P + blkdiag(2 gamma_i I2)]] (:643-657), inactive
        // contacts zeroed
        double a[M];
#pragma unroll
        for (int k = 0; k < M; ++k) a[k] = 0.0;
        if (is_c) {
            if (c_act) {
#pragma unroll
                for (int k = 0; k < M; ++k) {
                    if (k == c) a[k] = Sc;
                    else if (k == NC + 2 * c) a[k] = gamma * (2 * xa);
                    else if (k == NC + 2 * c + 1) a[k] = gamma * (2 * xb);
                }
            }
        } else if (is_l) {
            const bool my_act = (am >> (i / 2)) & 1ull;
#pragma unroll
            for (int k = 0; k < NC; ++k) a[k] = (k == i / 2 && my_act) ? 2 * xi : 0.0;
            const double g2 = 2 * vw[i / 2];
#pragma unroll
            for (int j = 0; j < N; ++j) a[NC + j] = ((j == i) ? g2 : 0.0) + Prow[j];
        }
        if (lane < M) vb[lane] = is_c ? 0.0 : gi;                           // :659-667
        DQQ_SYNC();
        const double xs = team_ir<M, M, LDA>(a, vb, AtL, Lm, v0, v1, v2, lane, steps);
        if (is_c) {
            const double dg = c_act ? xs : 0.0;                             // :671-674
            if (gout0 != nullptr) gout0[prob * NC + c] = QcqpContact::e2(gamma, ln, mc) * dg;   // grad_l_n
            if (gout1 != nullptr) gout1[prob * NC + c] = QcqpContact::e1(gamma, ln, mc) * dg;   // grad_mu
            if (gamma_out != nullptr) gamma_out[prob * NC + c] = gamma;
            if (dgamma_out != nullptr) dgamma_out[prob * NC + c] = dg;
        } else if (is_l) {
            if (grad_q != nullptr) grad_q[prob * N + i] = -xs;
            if (grad_P != nullptr) {
                double* Gp = grad_P + prob * (long)(N * N) + i * N;
#pragma unroll
                for (int j = 0; j < N; ++j) Gp[j] = -(xs * vx[j]);
            }
        }
        if (ir_steps != nullptr && lane == 0) ir_steps[prob] = steps;
    } else {
        // box QP.  slots: 2i = lower multiplier of coordinate i, 2i+1 = upper one, 2N + i = dl_i
        const bool is_m = lane < 2 * N, is_l = lane >= 2 * N && lane < M;
        const int i = is_m ? lane / 2 : (is_l ? lane - 2 * N : 0);
        const bool upper = is_m && (lane & 1);
        double Prow[N];
#pragma unroll
        for (int j = 0; j < N; ++j) Prow[j] = Pg[i * N + j];
        const double xi = x[prob * N + i], gi = grad_x[prob * N + i], qi = q[prob * N + i];
        const double lo = aux0[prob * N + i], hi = aux1[prob * N + i];
        if (is_l) vx[i] = xi;
        DQQ_SYNC();
        const bool aL = !(xi - lo > dual_eps), aU = !(xi - hi < -dual_eps);   // :268-283 / :315-327
        const bool slot_act = is_m && (upper ? aU : aL);
        const double sgn = upper ? 1.0 : -1.0;                              // Id2, :291-300
        // ---- dualFromPrimalBoxQP: gamma_not_null = iterative_refinement(Id2, -P*l - q), :301
        double rhs = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) rhs += (-Prow[j]) * vx[j];
        rhs = rhs - qi;
        if (is_l) vb[i] = rhs;
        double ad[N];
#pragma unroll
        for (int k = 0; k < N; ++k) ad[k] = (slot_act && k == i) ? sgn : 0.0;
        DQQ_SYNC();
        int steps_dual = 0;
        const double gsol = team_ir<2 * N, N, LDA>(ad, vb, AtL, Lm, v0, v1, v2, lane, steps_dual);
        const double gamma = slot_act ? gsol : 0.0;                         // :302-304
        DQQ_SYNC();
        if (is_m) vw[lane] = gamma;
        DQQ_SYNC();
        // ---- solveDerivativesBoxQP, :341-369: row `slot` of A = [[0, B],[Id2, P]]
        double a[M];
#pragma unroll
        for (int k = 0; k < M; ++k) a[k] = 0.0;
        if (is_m) {
#pragma unroll
            for (int k = 0; k < N; ++k)
                if (k == i) a[2 * N + k] = slot_act ? gamma * sgn : 0.0;       // B.row(j) = gamma_j * Id2.col(j)^T
        } else if (is_l) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                if (k == i) {
                    a[2 * k] = aL ? -1.0 : 0.0;
                    a[2 * k + 1] = aU ? 1.0 : 0.0;
                }
                a[2 * N + k] = Prow[k];
            }
        }
        if (lane < M) vb[lane] = is_m ? 0.0 : gi;                           // :352-360
        DQQ_SYNC();
        const double xs = team_ir<M, M, LDA>(a, vb, AtL, Lm, v0, v1, v2, lane, steps);
        DQQ_SYNC();
        // dgamma, scattered like the reference's blgamma[0:2N] (lower | upper), next to gamma
        if (is_m) v0[lane] = slot_act ? xs : 0.0;                           // :363-366
        DQQ_SYNC();
        if (is_l) {
            const double glo = vw[2 * i], ghi = vw[2 * i + 1], dlo = v0[2 * i], dhi = v0[2 * i + 1];
            if (grad_q != nullptr) grad_q[prob * N + i] = -xs;
            if (gout0 != nullptr) gout0[prob * N + i] = -(dlo * glo);       // grad_l_min
            if (gout1 != nullptr) gout1[prob * N + i] = dhi * ghi;          // grad_l_max
            if (gamma_out != nullptr) { gamma_out[prob * 2 * N + i] = glo; gamma_out[prob * 2 * N + N + i] = ghi; }
            if (dgamma_out != nullptr) { dgamma_out[prob * 2 * N + i] = dlo; dgamma_out[prob * 2 * N + N + i] = dhi; }
            if (grad_P != nullptr) {
                double* Gp = grad_P + prob * (long)(N * N) + i * N;
#pragma unroll
                for (int j = 0; j < N; ++j) Gp[j] = -(xs * vx[j]);
            }
        }
        if (ir_steps != nullptr && lane == 0) { ir_steps[2 * prob] = steps_dual; ir_steps[2 * prob + 1] = steps; }
    }
    DQQ_SYNC();
}


} // namespace dqq
