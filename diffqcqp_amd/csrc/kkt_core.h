// kkt_core.h -- implicit-function backward for a DIAGONAL P, per coordinate
// (QP) / per contact (QCQP).
//
// Restates, for diagonal P, the composition the reference runs per problem:
//   QP   pybindings.cpp:24-30 -> Solver::dualFromPrimalQP (Solver.cpp:125-134),
//        Solver::solveDerivativesQP (:136-196), Solver::iterative_refinement (:15-44)
//   QCQP pybindings.cpp:62-71 -> Solver::dualFromPrimalQCQP (:584-617),
//        Solver::getE12QCQP (:683-691), Solver::solveDerivativesQCQP (:619-681),
//        Solver::iterative_refinement (:15-44)
//
// With P diagonal the reference's (N + n_active)-dimensional system is, up to
// the permutation it applies, block diagonal: one 3x3 block (dgamma_c, dl_2c,
// dl_2c+1) per active contact, 1x1 blocks otherwise.  A dense Cholesky /
// triangular solve of a permuted block-diagonal matrix never mixes blocks and
// only adds exact zeros across them, so evaluating block by block IN THE
// REFERENCE'S OPERATION ORDER (this file; compile with -ffp-contract=off)
// reproduces the dense arithmetic bit for bit.  The one problem-wide quantity
// is the refinement residual ||K x - A^T b||_2 that drives the loop exit
// (:31-41); callers sum the squared residual entries returned here in the
// reference's order (gamma entries in contact order, then l entries in index
// order) and run the loop control in ir_control().
#pragma once

#include "common.h"
#include <float.h>

namespace dqq {

// Loop control of Solver::iterative_refinement, Solver.cpp:26-41.
struct IrControl {
    double res_pred;
    int not_improved;
    DQQ_HD void init() { res_pred = DBL_MAX; not_improved = 0; }
    // feed the residual of the step just taken; returns true when the loop exits
    DQQ_HD bool update(double res)
    {
        if (res_pred - res < kIrEps) {
            not_improved++;
        } else {
            res_pred = res;
            not_improved = 0;
        }
        return res < kIrEps || not_improved == 2;
    }
};

// ------------------------------------------------------------------ QP
// One coordinate of a diagonal-P QP.  Active coordinates (gamma < -1e-10)
// contribute the 1x1 block diag(l_A) with a zero right-hand side: everything
// about them is exactly zero, so only the inactive case carries state.
struct QpCoord {
    bool act;
    double K, Kinv, Ab, KinvAb, xs;

    DQQ_HD void setup(double p, double q, double x, double g, double eps = kActiveEps)
    {
        double gamma = -(p * x + q);               // dualFromPrimalQP, :127
        if (x > eps) gamma = 0;                    // :128-131 (epsilon: pybindings.cpp:24, default 1e-10)
        act = gamma < -kActiveEps;                 // :139-141
        xs = 0.0;
        if (act) { K = Kinv = Ab = KinvAb = 0.0; return; }
        Ab = p * g;                                // A^T b, :19  (A = P_II, diagonal)
        K = p * p + kMuIr;                         // A^T A + mu_ir I, :20-21
        const double L = sqrt(K);                  // llt, :23
        Kinv = (1.0 / L) / L;                      // solveInPlace(Identity), :22-23
        KinvAb = Kinv * Ab;                        // :27
    }
    // one refinement step (:29-31); returns this coordinate's squared residual entry
    DQQ_HD double step()
    {
        if (act) return 0.0;
        const double t = Kinv * xs;
        xs = kMuIr * t + KinvAb;
        const double d = K * xs - Ab;
        return d * d;
    }
    DQQ_HD double dl() const { return act ? 0.0 : xs; } // :187-191
};

// ------------------------------------------------------------------ QCQP
// One contact (coordinates a = 2c, b = 2c+1) of a diagonal-P QCQP.
struct QcqpContact {
    bool act;          // in the derivative system's active set (Solver.cpp:639)
    double gamma;      // dual of the contact (0 when inactive), dualFromPrimalQCQP
    double K[3][3], Kinv[3][3], Ab[3], KinvAb[3], xs[3]; // order: gamma, a, b

    DQQ_HD void setup(double pa, double pb, double qa, double qb, double xa, double xb, double ga, double gb,
                      double l_n, double mu, double eps = kActiveEps)
    {
        const double r = l_n * mu;                             // pybindings.cpp:65
        // ---- dualFromPrimalQCQP, Solver.cpp:584-617 (epsilon: pybindings.cpp:62, default 1e-10)
        {
            const double slack = r + -sqrt(xa * xa + xb * xb); // :594-597
            gamma = 0.0;
            if (!(slack > eps || r < eps)) {                   // :598-604
                const double ca = 2 * xa, cb = 2 * xb;         // A(2i,i), A(2i+1,i), :589-592
                const double G = ca * ca + cb * cb;            // A~^T A~ (diagonal)
                const double pla = pa * xa + qa, plb = pb * xb + qb; // P l + q
                const double rhs = ca * pla + cb * plb;        // A~^T (P l + q)
                const double L = sqrt(G);                      // llt().solve, :611
                gamma = -((rhs / L) / L);
            }
        }
        // ---- solveDerivativesQCQP, Solver.cpp:619-681
        double S = -(r * r);                                   // :622
        S = S + (xa * xa + xb * xb);                           // :628-629
        act = S > -kActiveEps && r > kActiveEps;               // :637-641
        const double ca = 2 * xa, cb = 2 * xb;                 // C, :630-631
        const double Da = 2 * gamma + pa, Db = 2 * gamma + pb; // D_tild + P, :632-633, :651
        xs[0] = xs[1] = xs[2] = 0.0;
        if (act) {
            const double Ba = gamma * ca, Bb = gamma * cb;     // diag(gamma) C^T, :644
            // A (before transposeInPlace) = [[S,Ba,Bb],[ca,Da,0],[cb,0,Db]]; b = [0,ga,gb]
            Ab[0] = Ba * ga + Bb * gb;                         // A^T_t b = A b, :19
            Ab[1] = Da * ga;
            Ab[2] = Db * gb;
            K[0][0] = (S * S + Ba * Ba) + Bb * Bb + kMuIr;     // A_t^T A_t + mu_ir I, :20-21
            K[0][1] = K[1][0] = S * ca + Ba * Da;
            K[0][2] = K[2][0] = S * cb + Bb * Db;
            K[1][1] = ca * ca + Da * Da + kMuIr;
            K[1][2] = K[2][1] = ca * cb;
            K[2][2] = cb * cb + Db * Db + kMuIr;
            // lower Cholesky, :23
            const double L00 = sqrt(K[0][0]);
            const double L10 = K[1][0] / L00, L20 = K[2][0] / L00;
            const double L11 = sqrt(K[1][1] - L10 * L10);
            const double L21 = (K[2][1] - L20 * L10) / L11;
            const double L22 = sqrt(K[2][2] - (L20 * L20 + L21 * L21));
            // solveInPlace(Identity): L y = e_c then L^T x = y, column by column
            {   // column 0
                const double y0 = 1.0 / L00, y1 = (0.0 - L10 * y0) / L11, y2 = ((0.0 - L20 * y0) - L21 * y1) / L22;
                const double x2 = y2 / L22, x1 = (y1 - L21 * x2) / L11, x0 = ((y0 - L10 * x1) - L20 * x2) / L00;
                Kinv[0][0] = x0; Kinv[1][0] = x1; Kinv[2][0] = x2;
            }
            {   // column 1
                const double y1 = 1.0 / L11, y2 = (0.0 - L21 * y1) / L22;
                const double x2 = y2 / L22, x1 = (y1 - L21 * x2) / L11, x0 = ((0.0 - L10 * x1) - L20 * x2) / L00;
                Kinv[0][1] = x0; Kinv[1][1] = x1; Kinv[2][1] = x2;
            }
            {   // column 2
                const double y2 = 1.0 / L22;
                const double x2 = y2 / L22, x1 = (0.0 - L21 * x2) / L11, x0 = ((0.0 - L10 * x1) - L20 * x2) / L00;
                Kinv[0][2] = x0; Kinv[1][2] = x1; Kinv[2][2] = x2;
            }
            for (int i = 0; i < 3; ++i)
                KinvAb[i] = (Kinv[i][0] * Ab[0] + Kinv[i][1] * Ab[1]) + Kinv[i][2] * Ab[2]; // :27
        } else {
            // the contact's two coordinates are 1x1 blocks of D_tild + P
            const double D[2] = {Da, Db}, g[2] = {ga, gb};
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) K[i][j] = Kinv[i][j] = 0.0;
            Ab[0] = KinvAb[0] = 0.0;
            for (int i = 0; i < 2; ++i) {
                Ab[i + 1] = D[i] * g[i];
                K[i + 1][i + 1] = D[i] * D[i] + kMuIr;
                const double L = sqrt(K[i + 1][i + 1]);
                Kinv[i + 1][i + 1] = (1.0 / L) / L;
                KinvAb[i + 1] = Kinv[i + 1][i + 1] * Ab[i + 1];
            }
        }
    }

    // one refinement step (:29-31); dsq = squared residual entries (gamma, a, b)
    DQQ_HD void step(double (&dsq)[3])
    {
        if (act) {
            double t[3];
            for (int i = 0; i < 3; ++i) t[i] = (Kinv[i][0] * xs[0] + Kinv[i][1] * xs[1]) + Kinv[i][2] * xs[2];
            for (int i = 0; i < 3; ++i) xs[i] = kMuIr * t[i] + KinvAb[i];
            for (int i = 0; i < 3; ++i) {
                const double d = ((K[i][0] * xs[0] + K[i][1] * xs[1]) + K[i][2] * xs[2]) - Ab[i];
                dsq[i] = d * d;
            }
        } else {
            dsq[0] = 0.0;
            for (int i = 1; i < 3; ++i) {
                const double t = Kinv[i][i] * xs[i];
                xs[i] = kMuIr * t + KinvAb[i];
                const double d = K[i][i] * xs[i] - Ab[i];
                dsq[i] = d * d;
            }
        }
    }
    DQQ_HD double dgamma() const { return act ? xs[0] : 0.0; }  // :671-674
    DQQ_HD double dla() const { return xs[1]; }
    DQQ_HD double dlb() const { return xs[2]; }
    // getE12QCQP, Solver.cpp:683-691 (raw l_n, mu)
    static DQQ_HD double e1(double gamma, double l_n, double mu) { return 2 * gamma * l_n * l_n * mu; }
    static DQQ_HD double e2(double gamma, double l_n, double mu) { return 2 * gamma * l_n * mu * mu; }
};

// ------------------------------------------------------------------ box QP
// Solver::iterative_refinement (Solver.cpp:15-44) restricted to one diagonal block of a permuted
// block-diagonal system, in the reference's operation order (left-looking LLT, column-wise
// solveInPlace(Identity), row-wise products accumulated from 0) -- the loops of the oracle's chol_inverse /
// matvec with the exact zeros of the other blocks left out.  The block always has three slots; a slot whose
// unknown does not exist carries a zero column of A_t and a zero right-hand side: it decouples into
// K = mu_ir, A^T b = 0, x = 0, residual 0, and contributes only exact zeros to the sums of the live slots,
// so no runtime indexing (scratch memory) is needed.  At is 3 x 3 (rows beyond the block's are zero).
struct SmallIr {
    double K[3][3], Kinv[3][3], Ab[3], KinvAb[3], xs[3];

    DQQ_HD void setup(const double (&At)[3][3], const double (&b)[3])
    {
        double L[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            xs[i] = 0.0;
            double s = 0.0;                                            // A^T b, :19
#pragma unroll
            for (int k = 0; k < 3; ++k) s += At[k][i] * b[k];
            Ab[i] = s;
#pragma unroll
            for (int j = 0; j < 3; ++j) {                              // A^T A, :20
                double t = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) t += At[k][i] * At[k][j];
                K[i][j] = t;
                L[i][j] = 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) K[i][i] += kMuIr;                  // :21
#pragma unroll
        for (int k = 0; k < 3; ++k) {                                  // llt(), :23
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < k; ++j) s += L[k][j] * L[k][j];
            double xk = K[k][k] - s;
            xk = sqrt(xk);
            L[k][k] = xk;
#pragma unroll
            for (int i = k + 1; i < 3; ++i) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < k; ++j) t += L[i][j] * L[k][j];
                L[i][k] = (K[i][k] - t) / xk;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {                                  // solveInPlace(Identity), :22-23
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double t = (i == c) ? 1.0 : 0.0;
#pragma unroll
                for (int j = 0; j < i; ++j) t -= L[i][j] * Kinv[j][c];
                Kinv[i][c] = t / L[i][i];
            }
#pragma unroll
            for (int i = 2; i >= 0; --i) {
                double t = Kinv[i][c];
#pragma unroll
                for (int j = i + 1; j < 3; ++j) t -= L[j][i] * Kinv[j][c];
                Kinv[i][c] = t / L[i][i];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {                                  // :27
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) s += Kinv[i][j] * Ab[j];
            KinvAb[i] = s;
        }
    }
    // one refinement step (:29-31); dsq[i] = squared residual entry of slot i
    DQQ_HD void step(double (&dsq)[3])
    {
        double t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) s += Kinv[i][j] * xs[j];
            t[i] = s;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) xs[i] = kMuIr * t[i] + KinvAb[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) s += K[i][j] * xs[j];
            const double d = s - Ab[i];
            dsq[i] = d * d;
        }
    }
};

// One coordinate of a diagonal-P box QP: Solver::dualFromPrimalBoxQP (Solver.cpp:263-308) then
// Solver::solveDerivativesBoxQP (:310-371).  Slots: 0 = lower multiplier (exists when l - l_min <= eps),
// 1 = upper multiplier (exists when l - l_max >= -eps), 2 = dl -- the reference's order.  The two refinement
// loops exit on problem-wide residual norms; the callers sum the squared entries returned here in the
// reference's entry order (multipliers coordinate by coordinate, then the l entries).
struct BoxCoord {
    bool aL, aU;
    double gamma_lo, gamma_hi;
    SmallIr ir;

    DQQ_HD void setup_dual(double p, double q, double x, double lo, double hi, double eps)
    {
        setup_dual_rhs((-p) * x - q, x, lo, hi, eps);                  // -P*l - q, :301 (diagonal P)
    }
    // general P: rhs = (-P*l - q)_i computed by the caller; the dual system itself never involves P
    DQQ_HD void setup_dual_rhs(double rhs, double x, double lo, double hi, double eps)
    {
        aL = !(x - lo > eps);                                          // :268-274
        aU = !(x - hi < -eps);                                         // :275-282
        // the coordinate's row of Id2 (:291-300): -1 in the lower multiplier's column, +1 in the upper one's
        const double At[3][3] = {{aL ? -1.0 : 0.0, aU ? 1.0 : 0.0, 0.0}, {0, 0, 0}, {0, 0, 0}};
        const double b[3] = {rhs, 0.0, 0.0};
        ir.setup(At, b);
    }
    // multipliers once the dual refinement loop has ended (:302-304)
    DQQ_HD void finish_dual()
    {
        gamma_lo = aL ? ir.xs[0] : 0.0;
        gamma_hi = aU ? ir.xs[1] : 0.0;
    }
    DQQ_HD void step_dual(double (&dsq)[3]) { ir.step(dsq); }           // dsq[2] is always 0
    DQQ_HD void setup_derivative(double p, double g)
    {
        gamma_lo = aL ? ir.xs[0] : 0.0;                                // :302-304
        gamma_hi = aU ? ir.xs[1] : 0.0;
        // A = [[0, B],[Id2, P]] (:341-350) restricted to (lower, upper, l); iterative_refinement gets A^T (:351)
        const double sL = aL ? -1.0 : 0.0, sU = aU ? 1.0 : 0.0;
        const double bL = gamma_lo * sL, bU = gamma_hi * sU;           // B.row(j) = gamma_j * Id2.col(j)^T
        const double At[3][3] = {{0.0, 0.0, sL}, {0.0, 0.0, sU}, {bL, bU, p}};
        const double b[3] = {0.0, 0.0, g};                             // :352-360
        ir.setup(At, b);
    }
    DQQ_HD void step_derivative(double (&dsq)[3]) { ir.step(dsq); }
    DQQ_HD double dl() const { return ir.xs[2]; }                       // :367-369
    DQQ_HD double dgamma_lo() const { return aL ? ir.xs[0] : 0.0; }     // :363-366
    DQQ_HD double dgamma_hi() const { return aU ? ir.xs[1] : 0.0; }
};

} // namespace dqq
