// dense_core.h -- per-problem device routines of the general (non-diagonal P) path: one wave64 per
// problem, matrices resident in wave-private LDS.  Used by the stand-alone dense kernels (dense.hip)
// and, for small N, as the in-kernel fallback of the diagonal fast paths (fwd_diag.hip, bwd_diag.hip).
//
// Direct parallel restatement of the reference's dense algebra:
//   forward   Solver::solveQP / solveQCQP (Solver.cpp:61-123, 521-582) with power_iteration (:46-59),
//             LLT + explicit inverse at every rho update (:76-77, 100-101, 114-115);
//   backward  pybindings.cpp:24-30 / 62-71 -> Solver.cpp:125-196, 584-691, iterative_refinement (:15-44).
// Lane i owns coordinate i (row i of a mat-vec, column i of the explicit inverse, row i of A^T A); every
// inner sum runs sequentially in index order and the 2-norms are summed sequentially from LDS, i.e. in
// the order a scalar CPU loop uses.  FP contraction is switched off inside every routine, so the
// arithmetic is the dense reference arithmetic operation for operation (device pow() aside).
//
// Row stride in LDS is odd (ld = n | 1): "lane i reads row i" is bank-conflict free and "all lanes
// read the same element" is a broadcast.  One lane per row limits N to 64 (QCQP backward: N + N/2 <= 64).
// The backward routine is templated on a TEAM width T (8, 16, 32 or 64 lanes): 64/T problems share a
// wave, each team working in its own LDS slice with team-local lane indices; teams never interact, so
// their loops may have different trip counts.  T = 64 is the one-problem-per-wave case.
#pragma once

#include "kkt_core.h"

namespace dqq {

// ballot restricted to the T-lane team of the calling lane (bit i = team-local lane i)
template <int T>
DQQ_D unsigned long long team_ballot(bool pred)
{
    const unsigned long long b = __ballot(pred);
    if constexpr (T == 64) return b;
    else return (b >> (((threadIdx.x & 63) / T) * T)) & ((1ull << T) - 1ull);
}

constexpr int kDenseMaxRows = 64;

#define DQQ_SYNC() wave_lds_fence()

// In-place lower Cholesky of A (n x n, row stride ld; only the lower triangle
// is read) followed by the explicit inverse into Ainv: Eigen's
// llt() + solveInPlace(Identity) as the reference uses it (Solver.cpp:76-77).
static DQQ_D void chol_inverse_wave(double* A, double* Ainv, int n, int ld, int lane)
{
#pragma clang fp contract(off)
    for (int k = 0; k < n; ++k) {
        double s = 0.0;
        for (int j = 0; j < k; ++j) { const double t = A[k * ld + j]; s += t * t; }
        const double xk = sqrt(A[k * ld + k] - s);
        double lik = 0.0;
        const bool below = lane > k && lane < n;
        if (below) {
            double t = 0.0;
            for (int j = 0; j < k; ++j) t += A[lane * ld + j] * A[k * ld + j];
            lik = (A[lane * ld + k] - t) / xk;
        }
        DQQ_SYNC();
        if (lane == k) A[k * ld + k] = xk;
        else if (below) A[lane * ld + k] = lik;
        DQQ_SYNC();
    }
    if (lane < n) { // lane = column of the inverse: L y = e_c, then L^T x = y
        for (int i = 0; i < n; ++i) {
            double t = (i == lane) ? 1.0 : 0.0;
            for (int j = 0; j < i; ++j) t -= A[i * ld + j] * Ainv[j * ld + lane];
            Ainv[i * ld + lane] = t / A[i * ld + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            double t = Ainv[i * ld + lane];
            for (int j = i + 1; j < n; ++j) t -= A[j * ld + i] * Ainv[j * ld + lane];
            Ainv[i * ld + lane] = t / A[i * ld + i];
        }
    }
    DQQ_SYNC();
}

// out_i = sum_j Mat[i][j] * vec[j], j in index order (lane = row i)
static DQQ_D double row_dot(const double* Mat, int ld, int row, const double* vec, int n)
{
#pragma clang fp contract(off)
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += Mat[row * ld + j] * vec[j];
    return s;
}

// sum_i vec[i]^2 in index order, computed redundantly by every lane
static DQQ_D double seq_sumsq(const double* vec, int n)
{
#pragma clang fp contract(off)
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += vec[i] * vec[i];
    return s;
}

template <int T = 64>
static DQQ_D void load_matrix(double* dst, int ld, const double* __restrict__ src, int n, int lane)
{
#pragma clang fp contract(off)
    for (int idx = lane; idx < n * n; idx += T) dst[(idx / n) * ld + idx % n] = src[idx];
}


constexpr int dense_fwd_lds_doubles(int n) { return 2 * n * (n | 1) + 2 * n + 2; }
constexpr int dense_bwd_rows(int kind, int n) { return kind == 0 ? n : (kind == 2 ? 3 * n : n + n / 2); }
constexpr int dense_bwd_lds_doubles(int kind, int n)
{
    return 3 * dense_bwd_rows(kind, n) * (dense_bwd_rows(kind, n) | 1) + 5 * n + 4 * dense_bwd_rows(kind, n) + 2 +
           (dense_bwd_rows(kind, n) + 3) / 2 + 1; // + perm ints
}

// One problem, forward: Solver::solveQP (KIND 0) / solveQCQP (KIND 1) / solveBoxQP (KIND 2, Solver.cpp:198-261;
// l_n = l_min, mu_c = l_max per coordinate) / solveSignedBoxQP (KIND 3, :374-439, + v) on the dense P of
// problem `prob`, executed by one wave.  smem: dense_fwd_lds_doubles(n) doubles of wave-private LDS.
template <int KIND>
static DQQ_D void dense_fwd_problem(const double* __restrict__ P, const double* __restrict__ q,
                                    const double* __restrict__ l_n, const double* __restrict__ mu_c,
                                    const double* __restrict__ v_sign, double* __restrict__ x, int* __restrict__ iters, long prob, int n, double eps,
                                    double mu, int max_iter, int adaptive, double* smem, int lane)
{
#pragma clang fp contract(off)
    const int ld = n | 1;
    double* A = smem;              // P, then P + shift (lower) -> its Cholesky factor, in place
    double* Ainv = A + n * ld;     // (P + shift)^-1
    double* va = Ainv + n * ld;    // n: vector broadcast buffer
    double* vb = va + n;           // n
    const bool act = lane < n;
    const double* Pg = P + prob * (long)n * n;
    load_matrix(A, ld, Pg, n, lane);
    DQQ_SYNC();

    // ---- power_iteration, Solver.cpp:46-59
    constexpr bool QP_LIKE = (KIND != 1);
    const int pi_steps = QP_LIKE ? 10 : 100;
    double v = 1 / sqrt((double)n);
    if (act) va[lane] = v;
    DQQ_SYNC();
    {
        const double s = seq_sumsq(va, n);
        if (s > 0) v = v / sqrt(s);
    }
    DQQ_SYNC();
    for (int k = 0; k < pi_steps; ++k) {
        if (act) va[lane] = v;
        DQQ_SYNC();
        const double Av = act ? row_dot(A, ld, lane, va, n) : 0.0;
        if (act) vb[lane] = Av;
        DQQ_SYNC();
        const double s = seq_sumsq(vb, n);
        v = (s > 0) ? Av / sqrt(s) : Av;
    }
    double Lmax;
    {
        if (act) va[lane] = v;
        DQQ_SYNC();
        const double Av = act ? row_dot(A, ld, lane, va, n) : 0.0;
        if (act) vb[lane] = v * Av;
        DQQ_SYNC();
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += vb[i];
        Lmax = s;
        DQQ_SYNC();
    }
    RhoSchedule sched;
    sched.init(Lmax, mu);                                           // :72-73 / :531-532
    double rho = sched.rho;
    double mdiag = act ? A[lane * ld + lane] + (rho + mu) : 0.0;    // :75 / :534 (accumulated diagonal)
    if (act) A[lane * ld + lane] = mdiag;
    DQQ_SYNC();
    chol_inverse_wave(A, Ainv, n, ld, lane);                        // :76-77

    const double qi = act ? q[prob * n + lane] : 0.0;
    double rad = 0.0;
    if (KIND == 1) rad = act ? l_n[prob * (n / 2) + lane / 2] * mu_c[prob * (n / 2) + lane / 2] : 0.0;
    double blo = 0.0, bhi = 0.0, bsg = 0.0;
    if (KIND >= 2 && act) {
        blo = l_n[prob * n + lane];
        bhi = mu_c[prob * n + lane];
        if (KIND == 3) { const double vv = v_sign[prob * n + lane]; bsg = (double)((vv > 0) - (vv < 0)); } // :395
    }
    double qp = qi, l2 = 0.0, l2p = 0.0, u = 0.0;
    int it_done = 0;
    for (int it = 0; it < max_iter; ++it) {
        it_done = it + 1;
        if (act) va[lane] = rho * l2 - u - qp;
        DQQ_SYNC();
        const double l = act ? row_dot(Ainv, ld, lane, va, n) : 0.0; // :80 / :539
        qp = qi - mu * l;                                            // :81 / :540
        double z = kAlpha * l + (1 - kAlpha) * l2 + u / rho;         // :82 / :541
        if (KIND == 0) {
            z = z < 0 ? 0 : z;
        } else if (KIND >= 2) {
            z = z < blo ? blo : z;                                   // cwiseMax(l_min), :219 / :396
            z = bhi < z ? bhi : z;                                   // cwiseMin(l_max), :220 / :397
            if (KIND == 3) {                                         // v o min(v o l_2, 0), :398
                double m = bsg * z;
                m = 0 < m ? 0 : m;
                z = bsg * m;
            }
        } else {                                                     // prox_circle, :505-519
            const double other = partner<1>(z);
            const double a = (lane & 1) ? other : z, b = (lane & 1) ? z : other;
            const double nrm = sqrt(a * a + b * b);
            if (nrm > rad) z = z * rad / nrm;
        }
        l2 = z;
        u += rho * (kAlpha * l + (1 - kAlpha) * l2p - l2);           // :83 / :543
        double rd, rp;
        if (QP_LIKE) rd = fabs(rho * (l2 - l2p)); else rd = fabs(l2 - l2p); // :84-85 / :544-545
        rp = fabs(l2 - (kAlpha * l + (1 - kAlpha) * l2p));           // :86 / :546
        rd = LaneGroup<64>::max(act ? rd : 0.0);
        rp = LaneGroup<64>::max(act ? rp : 0.0);
        const double res_dual = QP_LIKE ? rd : rho * rd;
        const double res_prim = rp;
        l2p = l2;                                                    // :87 / :547
        bool stop = res_dual < eps;
        if (KIND == 1) {                                             // :548
            DQQ_SYNC();
            if (act) vb[lane] = l;
            DQQ_SYNC();
            const double nl = sqrt(seq_sumsq(vb, n));
            stop = (res_prim < eps + kEpsRel * nl) && stop;
        }
        DQQ_SYNC();
        if (stop) break;
        if (adaptive) {
            double delta = 0.0;
            const bool upd = sched.template update<QP_LIKE, false>(res_prim, res_dual, delta); // :90-120 / :550-580
            if (upd) {
                mdiag += delta;
                rho = sched.rho;
            }
            if (upd) { // llt() of the shifted matrix + explicit inverse
                load_matrix(A, ld, Pg, n, lane);
                DQQ_SYNC();
                if (act) A[lane * ld + lane] = mdiag;
                DQQ_SYNC();
                chol_inverse_wave(A, Ainv, n, ld, lane);
            }
        }
    }
    if (act) x[prob * n + lane] = l2;
    if (iters != nullptr && lane == 0) iters[prob] = it_done;
    DQQ_SYNC();
}

// ----------------------------------------------------------------- backward
// Solver::iterative_refinement (Solver.cpp:15-44) for the m x m system whose
// TRANSPOSED matrix A_t sits in At and right-hand side in dd (LDS).  Lane i
// returns entry i of the solution.  At is overwritten (Cholesky workspace).
template <int T = 64>
static DQQ_D double ir_wave(double* At, double* K, double* Kinv, const double* dd, double* va, double* vb, int m,
                            int ld, int lane, int& steps, int rows = -1)
{
#pragma clang fp contract(off)
    // rows x m system matrix (rows = m unless given: the box QP's dual recovery is rectangular)
    if (rows < 0) rows = m;
    const bool act = lane < m;
    double Ab = 0.0;
    if (act) {
        for (int k = 0; k < rows; ++k) Ab += At[k * ld + lane] * dd[k];       // A^T b, :19
        for (int j = 0; j < m; ++j) {                                         // A^T A, :20
            double s = 0.0;
            for (int k = 0; k < rows; ++k) s += At[k * ld + lane] * At[k * ld + j];
            K[lane * ld + j] = s;
        }
        K[lane * ld + lane] += kMuIr;                                         // :21
    }
    DQQ_SYNC();
    for (int idx = lane; idx < m * m; idx += T) At[(idx / m) * ld + idx % m] = K[(idx / m) * ld + idx % m];
    if (act) va[lane] = Ab;
    DQQ_SYNC();
    chol_inverse_wave(At, Kinv, m, ld, lane);                                 // :22-23
    const double KinvAb = act ? row_dot(Kinv, ld, lane, va, m) : 0.0;         // :27
    double xs = 0.0;
    IrControl ctl;
    ctl.init();
    steps = 0;
    for (int it = 0; it < kIrMaxIter; ++it) {
        steps = it + 1;
        if (act) vb[lane] = xs;
        DQQ_SYNC();
        const double t = act ? row_dot(Kinv, ld, lane, vb, m) : 0.0;          // :29
        xs = kMuIr * t + KinvAb;
        DQQ_SYNC();
        if (act) vb[lane] = xs;
        DQQ_SYNC();
        const double d = act ? row_dot(K, ld, lane, vb, m) - Ab : 0.0;        // :30
        DQQ_SYNC();
        if (act) vb[lane] = d;
        DQQ_SYNC();
        const double res = sqrt(seq_sumsq(vb, m));                            // :31
        DQQ_SYNC();
        if (ctl.update(res)) break;                                           // :32-41
    }
    return xs;
}

// Box QP backward, one problem per team: pybindings.cpp:39-45 -> Solver::dualFromPrimalBoxQP
// (Solver.cpp:263-308), Solver::solveDerivativesBoxQP (:310-371), two runs of iterative_refinement (:15-44),
// and the gradient assembly BoxQPFn2.backward intends (qcqp.py:87-93; signs as settled by finite
// differences, tests/test_oracle.py): grad_P = -dl x^T, grad_q = -dl, grad_l_min = -dgamma_lo o gamma_lo,
// grad_l_max = +dgamma_hi o gamma_hi.  Multipliers are ordered as the reference orders them: coordinate by
// coordinate, the lower one (if l_i - l_min_i <= eps) before the upper one (if l_i - l_max_i >= -eps).
// smem: dense_bwd_lds_doubles(2, n).  ir_steps: two ints per problem (dual recovery, derivative system).
template <int T>
static DQQ_D void dense_bwd_box_problem(const double* __restrict__ P, const double* __restrict__ q,
                                        const double* __restrict__ l_min, const double* __restrict__ l_max,
                                        const double* __restrict__ x, const double* __restrict__ grad_x,
                                        double* __restrict__ grad_P, double* __restrict__ grad_q,
                                        double* __restrict__ grad_l_min, double* __restrict__ grad_l_max,
                                        double* __restrict__ gamma_out, double* __restrict__ dgamma_out,
                                        int* __restrict__ ir_steps, long prob, int n, double dual_eps, double* smem,
                                        int lane)
{
#pragma clang fp contract(off)
    const int mmax = 3 * n;
    const int ld = mmax | 1;
    double* At = smem;               // mmax*ld
    double* K = At + mmax * ld;      // mmax*ld (holds P while a system is assembled)
    double* Kinv = K + mmax * ld;    // mmax*ld
    double* vx = Kinv + mmax * ld;   // n
    double* vg = vx + n;             // n
    double* vdd = vg + n;            // mmax: right-hand side
    double* va = vdd + mmax;         // mmax
    double* vb = va + mmax;          // mmax
    double* vgam = vb + mmax;        // 2n: gamma, scattered (lower | upper)
    double* vdg = vgam + 2 * n;      // 2n: dgamma, scattered
    double* vdl = vdg + 2 * n;       // n
    int* nnl = reinterpret_cast<int*>(vdl + n); // 2n ints: not_null list
    const unsigned long long below = (1ull << lane) - 1ull;
    const double* Pg = P + prob * (long)n * n;
    double* Pl = K;
    load_matrix<T>(Pl, ld, Pg, n, lane);
    const bool actn = lane < n;
    const double xi = actn ? x[prob * n + lane] : 0.0;
    const double gi = actn ? grad_x[prob * n + lane] : 0.0;
    const double qi = actn ? q[prob * n + lane] : 0.0;
    const double lo = actn ? l_min[prob * n + lane] : 0.0, hi = actn ? l_max[prob * n + lane] : 0.0;
    if (actn) { vx[lane] = xi; vg[lane] = gi; vgam[lane] = 0.0; vgam[n + lane] = 0.0; vdg[lane] = 0.0; vdg[n + lane] = 0.0; }
    // not_null bookkeeping, :268-283 / :315-327
    const bool aL = actn && !(xi - lo > dual_eps);
    const bool aU = actn && !(xi - hi < -dual_eps);
    const unsigned long long mL = team_ballot<T>(aL), mU = team_ballot<T>(aU);
    const int nn = __popcll(mL) + __popcll(mU);
    const int posL = __popcll(mL & below) + __popcll(mU & below);
    const int posU = posL + (aL ? 1 : 0);
    if (aL) nnl[posL] = lane;
    if (aU) nnl[posU] = n + lane;
    DQQ_SYNC();
    // ---- dualFromPrimalBoxQP: gamma_not_null = iterative_refinement(Id2, -P*l - q), :291-304
    double rhs = 0.0;
    if (actn) {
        for (int j = 0; j < n; ++j) rhs += (-Pl[lane * ld + j]) * vx[j];
        rhs = rhs - qi;
    }
    for (int idx = lane; idx < n * nn; idx += T) At[(idx / nn) * ld + idx % nn] = 0.0;   // Id2: n x nn
    if (actn) vdd[lane] = rhs;
    DQQ_SYNC();
    if (aL) At[lane * ld + posL] = -1.0;
    if (aU) At[lane * ld + posU] = 1.0;
    DQQ_SYNC();
    int steps_dual = 1, steps = 0; // nn == 0: the reference runs one loop body on empty vectors and leaves
    if (nn > 0) {
        const double gnn = ir_wave<T>(At, K, Kinv, vdd, va, vb, nn, ld, lane, steps_dual, n);
        if (lane < nn) vgam[nnl[lane]] = gnn;
    }
    DQQ_SYNC();
    // ---- solveDerivativesBoxQP, :341-369: A = [[0, B],[Id2, P]], B.row(j) = gamma_j * Id2.col(j)^T
    const int m = nn + n;
    load_matrix<T>(Pl, ld, Pg, n, lane); // the refinement above used K as workspace
    DQQ_SYNC();
    for (int idx = lane; idx < m * m; idx += T) {
        const int rr = idx / m, cc = idx % m; // At[rr][cc] = A[cc][rr]: row = cc, col = rr
        const int row = cc, col = rr;
        double val = 0.0;
        if (row < nn) {
            if (col >= nn) {
                const int id = nnl[row], i = col - nn;
                const double s = (id < n) ? ((id == i) ? -1.0 : 0.0) : ((id - n == i) ? 1.0 : 0.0); // Id2(i, row)
                val = vgam[id] * s;
            }
        } else {
            const int i = row - nn;
            if (col < nn) {
                const int id = nnl[col];
                val = (id < n) ? ((id == i) ? -1.0 : 0.0) : ((id - n == i) ? 1.0 : 0.0);
            } else {
                val = Pl[i * ld + (col - nn)];
            }
        }
        At[rr * ld + cc] = val;
    }
    if (lane < m) vdd[lane] = (lane < nn) ? 0.0 : vg[lane - nn];            // :352-360
    DQQ_SYNC();
    const double bsol = ir_wave<T>(At, K, Kinv, vdd, va, vb, m, ld, lane, steps); // :362
    if (lane < nn) vdg[nnl[lane]] = bsol;                                   // :363-366
    else if (lane < m) vdl[lane - nn] = bsol;                               // :367-369
    DQQ_SYNC();
    if (actn) {
        const double glo = vgam[lane], ghi = vgam[n + lane], dlo = vdg[lane], dhi = vdg[n + lane];
        if (grad_q != nullptr) grad_q[prob * n + lane] = -vdl[lane];
        if (grad_l_min != nullptr) grad_l_min[prob * n + lane] = -(dlo * glo);
        if (grad_l_max != nullptr) grad_l_max[prob * n + lane] = dhi * ghi;
        if (gamma_out != nullptr) { gamma_out[prob * 2 * n + lane] = glo; gamma_out[prob * 2 * n + n + lane] = ghi; }
        if (dgamma_out != nullptr) { dgamma_out[prob * 2 * n + lane] = dlo; dgamma_out[prob * 2 * n + n + lane] = dhi; }
    }
    if (grad_P != nullptr) {
        double* Gp = grad_P + prob * (long)n * n;
        for (int idx = lane; idx < n * n; idx += T) Gp[idx] = -(vdl[idx / n] * vx[idx % n]);
    }
    if (ir_steps != nullptr && lane == 0) { ir_steps[2 * prob] = steps_dual; ir_steps[2 * prob + 1] = steps; }
    DQQ_SYNC();
}

// One problem, backward: the composition of pybindings.cpp:24-30 (KIND 0) / :62-71 (KIND 1) plus the
// gradient assembly of qcqp.py:48-51 / :173-180, executed by one wave.  smem: dense_bwd_lds_doubles(KIND,n).
template <int KIND, int T = 64>
static DQQ_D void dense_bwd_problem(const double* __restrict__ P, const double* __restrict__ q,
                                    const double* __restrict__ l_n, const double* __restrict__ mu_c,
                                    const double* __restrict__ x, const double* __restrict__ grad_x,
                                    double* __restrict__ grad_P, double* __restrict__ grad_q,
                                    double* __restrict__ grad_l_n, double* __restrict__ grad_mu,
                                    double* __restrict__ gamma_out, double* __restrict__ dgamma_out,
                                    int* __restrict__ ir_steps, long prob, int n, double dual_eps, double* smem,
                                    int lane)
{
#pragma clang fp contract(off)
    if constexpr (KIND == 2) { // box QP: l_n = l_min, mu_c = l_max, grad_l_n = grad_l_min, grad_mu = grad_l_max
        dense_bwd_box_problem<T>(P, q, l_n, mu_c, x, grad_x, grad_P, grad_q, grad_l_n, grad_mu, gamma_out, dgamma_out,
                                 ir_steps, prob, n, dual_eps, smem, lane);
        return;
    }
    const int nc = n / 2;
    const int mmax = (KIND == 0) ? n : n + nc;
    const int ld = mmax | 1;
    double* At = smem;               // mmax*ld
    double* K = At + mmax * ld;      // mmax*ld (holds P while the system is assembled)
    double* Kinv = K + mmax * ld;    // mmax*ld
    double* vx = Kinv + mmax * ld;   // n   : x
    double* vg = vx + n;             // n   : grad_x
    double* vdd = vg + n;            // mmax: right-hand side
    double* va = vdd + mmax;         // mmax
    double* vb = va + mmax;          // mmax
    double* vgam = vb + mmax;        // n   : gamma (QP: per coordinate, QCQP: per contact)
    double* vS = vgam + n;           // nc
    double* vdl = vS + n;            // n
    int* perm = reinterpret_cast<int*>(vdl + n); // mmax ints: QP position -> coordinate, QCQP active slot -> contact
    const unsigned long long below = (1ull << lane) - 1ull; // lane is team-local: 0 <= lane < T
    const double* Pg = P + prob * (long)n * n;
    double* Pl = K; // P in LDS, row stride ld
    load_matrix<T>(Pl, ld, Pg, n, lane);
    const bool actn = lane < n;
    const double xi = actn ? x[prob * n + lane] : 0.0;
    const double gi = actn ? grad_x[prob * n + lane] : 0.0;
    const double qi = actn ? q[prob * n + lane] : 0.0;
    if (actn) { vx[lane] = xi; vg[lane] = gi; }
    DQQ_SYNC();
    int steps = 0, m, na;
    double bsol;

    if (KIND == 0) {
        // dualFromPrimalQP, Solver.cpp:125-134
        double gamma = actn ? -(row_dot(Pl, ld, lane, vx, n) + qi) : 0.0;
        if (xi > dual_eps) gamma = 0;
        const bool is_act = actn && gamma < -kActiveEps;                  // :139-147
        const unsigned long long am = team_ballot<T>(is_act);
        const unsigned long long im = team_ballot<T>(actn && !is_act);
        na = __popcll(am);
        m = n;
        const int pos = is_act ? __popcll(am & below) : na + __popcll(im & below);
        if (actn) perm[pos] = lane;
        DQQ_SYNC();
        // A = [[diag(l_A), 0],[0, P_II]] in the order (active..., inactive...); At = A^T, :148-174
        for (int idx = lane; idx < m * m; idx += T) {
            const int r = idx / m, c = idx % m; // At[r][c] = A[c][r]
            double val;
            if (c < na || r < na) val = (c == r) ? vx[perm[c]] : 0.0;
            else val = Pl[perm[c] * ld + perm[r]];
            At[r * ld + c] = val;
        }
        if (actn) vdd[pos] = (pos < na) ? 0.0 : gi;                       // :175-184
        DQQ_SYNC();
        bsol = ir_wave<T>(At, K, Kinv, vdd, va, vb, m, ld, lane, steps);     // :186
        if (actn) vdl[perm[lane]] = (lane < na) ? 0.0 : bsol;             // :187-191
        DQQ_SYNC();
    } else {
        // dualFromPrimalQCQP, Solver.cpp:584-617: lane c <-> contact c
        const double plq = actn ? row_dot(Pl, ld, lane, vx, n) + qi : 0.0;
        if (actn) va[lane] = plq;
        DQQ_SYNC();
        const bool actc = lane < nc;
        const double ln = actc ? l_n[prob * nc + lane] : 1.0, mc = actc ? mu_c[prob * nc + lane] : 1.0;
        const double r = ln * mc;                                         // pybindings.cpp:65
        const double xa = actc ? vx[2 * lane] : 0.0, xb = actc ? vx[2 * lane + 1] : 0.0;
        double gamma = 0.0;
        {
            const double slack = r + -sqrt(xa * xa + xb * xb);
            if (actc && !(slack > dual_eps || r < dual_eps)) {
                const double ca = 2 * xa, cb = 2 * xb;
                const double G = ca * ca + cb * cb;
                const double rhs = ca * va[2 * lane] + cb * va[2 * lane + 1];
                const double L = sqrt(G);
                gamma = -((rhs / L) / L);
            }
        }
        double S = -(r * r);                                              // Solver.cpp:622-629
        S = S + (xa * xa + xb * xb);
        const bool is_act = actc && S > -kActiveEps && r > kActiveEps;    // :637-641
        const unsigned long long am = team_ballot<T>(is_act);
        na = __popcll(am);
        m = n + na;
        const int slot = __popcll(am & below);
        DQQ_SYNC();
        if (actc) { vgam[lane] = gamma; vS[lane] = S; }
        if (is_act) perm[slot] = lane;
        DQQ_SYNC();
        // A = [[diag(S_act), (diag(gamma) C^T)_act],[C_act, P + blkdiag(2 gamma_i I2)]]; At = A^T, :643-657
        for (int idx = lane; idx < m * m; idx += T) {
            const int rr = idx / m, cc = idx % m; // At[rr][cc] = A[cc][rr]: row = cc, col = rr
            const int row = cc, col = rr;
            double val;
            if (row < na) {
                const int cid = perm[row];
                if (col < na) val = (col == row) ? vS[cid] : 0.0;
                else { const int i = col - na; val = (i / 2 == cid) ? vgam[cid] * (2 * vx[i]) : 0.0; }
            } else {
                const int i = row - na;
                if (col < na) { const int cid = perm[col]; val = (i / 2 == cid) ? 2 * vx[i] : 0.0; }
                else { const int jj = col - na; const double d = (i == jj) ? 2 * vgam[i / 2] : 0.0; val = d + Pl[i * ld + jj]; }
            }
            At[rr * ld + cc] = val;
        }
        if (lane < m) vdd[lane] = (lane < na) ? 0.0 : vg[lane - na];      // :659-667
        DQQ_SYNC();
        bsol = ir_wave<T>(At, K, Kinv, vdd, va, vb, m, ld, lane, steps);     // :669
        // blgamma scatter, :670-679
        if (lane < na) vb[perm[lane]] = bsol;        // dgamma of active contacts
        else if (lane < m) vdl[lane - na] = bsol;    // dl
        DQQ_SYNC();
        if (actc) {
            const double dg = is_act ? vb[lane] : 0.0;
            if (grad_l_n != nullptr) grad_l_n[prob * nc + lane] = QcqpContact::e2(gamma, ln, mc) * dg;
            if (grad_mu != nullptr) grad_mu[prob * nc + lane] = QcqpContact::e1(gamma, ln, mc) * dg;
            if (gamma_out != nullptr) gamma_out[prob * nc + lane] = gamma;
            if (dgamma_out != nullptr) dgamma_out[prob * nc + lane] = dg;
        }
    }
    if (actn && grad_q != nullptr) grad_q[prob * n + lane] = -vdl[lane];
    if (grad_P != nullptr) {
        double* Gp = grad_P + prob * (long)n * n;
        for (int idx = lane; idx < n * n; idx += T) Gp[idx] = -(vdl[idx / n] * vx[idx % n]);
    }
    if (ir_steps != nullptr && lane == 0) ir_steps[prob] = steps;
    DQQ_SYNC();
}

} // namespace dqq
