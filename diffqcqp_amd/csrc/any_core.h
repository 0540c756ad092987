// any_core.h -- per-problem device routines of the general path WITHOUT a size limit: one 256-thread workgroup
// per problem, every matrix and vector in a per-workgroup slice of GLOBAL memory (L2-resident for moderate N),
// coordinates strided over the threads.  The reference has no size limit either (Solver.cpp:61: any n); this is
// the kernel behind every N the register / LDS kernels do not hold, and -- with DQQ_F_REFERENCE_ORDER -- the
// reference-ORDER QCQP backward for 42 < N <= 64, where the matrix-core kernels (bwd_wave_qcqp_big.hip) are the default.
//
// It is dense_core.h with "lane i owns coordinate i" replaced by "thread t owns coordinates t, t+256, ...":
// the same restatement of Solver::solveQP / solveQCQP / solveBoxQP / solveSignedBoxQP (Solver.cpp:61-123,
// 521-582, 198-261, 374-439), of pybindings.cpp:24-30 / 39-45 / 62-71 -> Solver.cpp:125-196, 263-371, 584-691 and
// of Solver::iterative_refinement (:15-44), every inner sum sequential in index order, FP contraction off:
// operation for operation the arithmetic of the wave kernels and of the oracle.
#pragma once

#include "kkt_core.h"

namespace dqq {

constexpr int kAnyT = 256;

#define DQQ_WG_SYNC() __syncthreads()

// max over the workgroup of a non-negative value (exact in any order); red: kAnyT doubles of LDS
static DQQ_D double wg_max(double v, double* red, int t)
{
    DQQ_WG_SYNC();
    red[t] = v;
    DQQ_WG_SYNC();
    for (int s = kAnyT / 2; s > 0; s >>= 1) {
        if (t < s) red[t] = fmax(red[t], red[t + s]);
        DQQ_WG_SYNC();
    }
    return red[0];
}

// In-place lower Cholesky of A (n x n, row stride ld; only the lower triangle is read) followed by the explicit
// inverse into Ainv: Eigen's llt() + solveInPlace(Identity) as the reference uses it (Solver.cpp:76-77), in the
// operation order of dense_core.h: chol_inverse_wave.
static DQQ_D void chol_inverse_wg(double* A, double* Ainv, int n, int ld, int t)
{
#pragma clang fp contract(off)
    for (int k = 0; k < n; ++k) {
        double s = 0.0;
#pragma unroll 8
        for (int j = 0; j < k; ++j) { const double v = A[k * ld + j]; s += v * v; }
        const double xk = sqrt(A[k * ld + k] - s);
        for (int i = k + 1 + t; i < n; i += kAnyT) {
            double v = 0.0;
#pragma unroll 8
            for (int j = 0; j < k; ++j) v += A[i * ld + j] * A[k * ld + j];
            A[i * ld + k] = (A[i * ld + k] - v) / xk;
        }
        DQQ_WG_SYNC(); // everybody has read A[k][k]
        if (t == 0) A[k * ld + k] = xk;
        DQQ_WG_SYNC();
    }
    for (int c = t; c < n; c += kAnyT) { // thread = column of the inverse: L y = e_c, then L^T x = y
        for (int i = 0; i < c; ++i) Ainv[i * ld + c] = 0.0; // L^-1 is lower triangular: exact zeros (0 - sum of 0 products)
        for (int i = c; i < n; ++i) {
            double v = (i == c) ? 1.0 : 0.0;
#pragma unroll 8 // loads ahead of the (sequential, reference-order) chain of subtractions
            for (int j = c; j < i; ++j) v -= A[i * ld + j] * Ainv[j * ld + c];
            Ainv[i * ld + c] = v / A[i * ld + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            double v = Ainv[i * ld + c];
#pragma unroll 8
            for (int j = i + 1; j < n; ++j) v -= A[j * ld + i] * Ainv[j * ld + c];
            Ainv[i * ld + c] = v / A[i * ld + i];
        }
    }
    DQQ_WG_SYNC();
}

static DQQ_D double any_row_dot(const double* Mat, int ld, int row, const double* vec, int n)
{
#pragma clang fp contract(off)
    double s = 0.0;
#pragma unroll 8
    for (int j = 0; j < n; ++j) s += Mat[row * ld + j] * vec[j];
    return s;
}

static DQQ_D double any_sumsq(const double* vec, int n)
{
#pragma clang fp contract(off)
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += vec[i] * vec[i];
    return s;
}

static DQQ_D void any_load_matrix(double* dst, int ld, const double* __restrict__ src, int n, int t)
{
    for (long idx = t; idx < (long)n * n; idx += kAnyT) dst[(idx / n) * ld + idx % n] = src[idx];
}

// Scratch of one workgroup, in doubles: `mat` matrices of rows x (rows|1) + a vector block.  The two matrices every
// O(n^3) loop runs on (forward: A, Ainv; backward: A^T, K^-1) go to LDS when they fit (kAnyLdsBytes); the rest
// (backward: K; all vectors) is always a slice of global memory.
constexpr size_t kAnyLdsBytes = 152 * 1024;
static DQQ_HD int any_bwd_rows(int kind, int n) { return kind == 0 ? n : (kind == 2 ? 3 * n : n + n / 2); }
static DQQ_HD long any_mat_doubles(int rows) { return (long)rows * (rows | 1); }
static DQQ_HD long any_fwd_vec_doubles(int n) { return 8L * n + 8; }
static DQQ_HD long any_bwd_vec_doubles(int kind, int n) { return 16L * any_bwd_rows(kind, n) + 8L * n + 16; }

// One problem, forward, by one workgroup.  red: kAnyT doubles of LDS.
template <int KIND>
static DQQ_D void any_fwd_problem(const double* __restrict__ P, const double* __restrict__ q,
                                  const double* __restrict__ l_n, const double* __restrict__ mu_c,
                                  const double* __restrict__ v_sign, double* __restrict__ x, int* __restrict__ iters,
                                  long prob, int n, double eps, double mu, int max_iter, int adaptive, double* A,
                                  double* Ainv, double* vec, double* red, int t)
{
    // A: P + shift (lower) -> its Cholesky factor, Ainv: the explicit inverse (n x (n|1) doubles each: LDS when they
    // fit, global memory otherwise); vec: 8n doubles of global memory
#pragma clang fp contract(off)
    constexpr bool QP_LIKE = (KIND != 1);
    const int ld = n | 1;
    double* va = vec;               // broadcast buffers
    double* vb = va + n;
    double* vqp = vb + n;           // ADMM state, one entry per coordinate
    double* vl2 = vqp + n;
    double* vl2p = vl2 + n;
    double* vu = vl2p + n;
    double* vl = vu + n;
    double* vmd = vl + n;           // accumulated shifted diagonal
    const double* Pg = P + prob * (long)n * n;
    const double* qg = q + prob * (long)n;

    // ---- power_iteration, Solver.cpp:46-59, on the full P
    any_load_matrix(A, ld, Pg, n, t);
    const double v0 = 1 / sqrt((double)n);
    for (int i = t; i < n; i += kAnyT) va[i] = v0;
    DQQ_WG_SYNC();
    {
        const double s = any_sumsq(va, n);
        DQQ_WG_SYNC();
        if (s > 0) for (int i = t; i < n; i += kAnyT) va[i] = v0 / sqrt(s);
        DQQ_WG_SYNC();
    }
    const int pi_steps = QP_LIKE ? 10 : 100;
    for (int k = 0; k < pi_steps; ++k) {
        for (int i = t; i < n; i += kAnyT) vb[i] = any_row_dot(A, ld, i, va, n);
        DQQ_WG_SYNC();
        const double s = any_sumsq(vb, n);
        for (int i = t; i < n; i += kAnyT) va[i] = (s > 0) ? vb[i] / sqrt(s) : vb[i];
        DQQ_WG_SYNC();
    }
    for (int i = t; i < n; i += kAnyT) vb[i] = va[i] * any_row_dot(A, ld, i, va, n);
    DQQ_WG_SYNC();
    double Lmax = 0.0;
    for (int i = 0; i < n; ++i) Lmax += vb[i];
    DQQ_WG_SYNC();

    RhoSchedule sched;
    sched.init(Lmax, mu);                                           // :72-73 / :531-532
    double rho = sched.rho;
    for (int i = t; i < n; i += kAnyT) {
        vmd[i] = A[i * ld + i] + (rho + mu);                        // :75 / :534 (accumulated diagonal)
        A[i * ld + i] = vmd[i];
        vqp[i] = qg[i];
        vl2[i] = 0.0;
        vl2p[i] = 0.0;
        vu[i] = 0.0;
    }
    DQQ_WG_SYNC();
    chol_inverse_wg(A, Ainv, n, ld, t);                             // :76-77

    int it_done = 0;
    for (int it = 0; it < max_iter; ++it) {
        it_done = it + 1;
        for (int i = t; i < n; i += kAnyT) va[i] = rho * vl2[i] - vu[i] - vqp[i];
        DQQ_WG_SYNC();
        double rd = 0.0, rp = 0.0;
        // a thread owns whole projection units: coordinates, or contact pairs for the QCQP
        constexpr int U = (KIND == 1) ? 2 : 1;
        for (int i0 = U * t; i0 < n; i0 += U * kAnyT) {
            double l[U], w[U], z[U];
#pragma unroll
            for (int e = 0; e < U; ++e) {
                const int i = i0 + e;
                l[e] = any_row_dot(Ainv, ld, i, va, n);                          // :80 / :539
                vl[i] = l[e];
                vqp[i] = qg[i] - mu * l[e];                                      // :81 / :540
                w[e] = kAlpha * l[e] + (1 - kAlpha) * vl2[i];
                z[e] = w[e] + vu[i] / rho;                                       // :82 / :541
            }
            if (KIND == 0) {
                z[0] = z[0] < 0 ? 0 : z[0];
            } else if (KIND >= 2) {
                const double blo = l_n[prob * (long)n + i0], bhi = mu_c[prob * (long)n + i0];
                z[0] = z[0] < blo ? blo : z[0];                                  // cwiseMax(l_min), :219 / :396
                z[0] = bhi < z[0] ? bhi : z[0];                                  // cwiseMin(l_max), :220 / :397
                if (KIND == 3) {                                                 // v o min(v o l_2, 0), :395-398
                    const double vv = v_sign[prob * (long)n + i0];
                    const double sg = (double)((vv > 0) - (vv < 0));
                    double m = sg * z[0];
                    m = 0 < m ? 0 : m;
                    z[0] = sg * m;
                }
            } else {                                                             // prox_circle, :505-519
                const long c = prob * (long)(n / 2) + i0 / 2;
                const double rad = l_n[c] * mu_c[c];                             // pybindings.cpp:57
                const double nrm = sqrt(z[0] * z[0] + z[U - 1] * z[U - 1]);
                if (nrm > rad) {
                    z[0] = z[0] * rad / nrm;
                    z[U - 1] = z[U - 1] * rad / nrm;
                }
            }
#pragma unroll
            for (int e = 0; e < U; ++e) {
                const int i = i0 + e;
                const double l2p = vl2p[i];
                vu[i] += rho * (kAlpha * l[e] + (1 - kAlpha) * l2p - z[e]);      // :83 / :543
                const double d = QP_LIKE ? fabs(rho * (z[e] - l2p)) : fabs(z[e] - l2p); // :84-85 / :544-545
                rd = fmax(rd, d);
                rp = fmax(rp, fabs(z[e] - (kAlpha * l[e] + (1 - kAlpha) * l2p))); // :86 / :546
                vl2[i] = z[e];
                vl2p[i] = z[e];                                                  // :87 / :547
            }
        }
        rd = wg_max(rd, red, t);
        rp = wg_max(rp, red, t);
        const double res_dual = QP_LIKE ? rd : rho * rd;
        const double res_prim = rp;
        bool stop = res_dual < eps;                                              // :88
        if (KIND == 1) {                                                         // :548
            const double nl = sqrt(any_sumsq(vl, n));
            stop = (res_prim < eps + kEpsRel * nl) && stop;
        }
        DQQ_WG_SYNC();
        if (stop) break;
        if (adaptive) {
            double delta = 0.0;
            if (sched.template update<QP_LIKE, false>(res_prim, res_dual, delta)) { // :90-120 / :550-580
                rho = sched.rho;
                any_load_matrix(A, ld, Pg, n, t);
                DQQ_WG_SYNC();
                for (int i = t; i < n; i += kAnyT) {
                    vmd[i] += delta;
                    A[i * ld + i] = vmd[i];
                }
                DQQ_WG_SYNC();
                chol_inverse_wg(A, Ainv, n, ld, t);
            }
        }
    }
    for (int i = t; i < n; i += kAnyT) x[prob * (long)n + i] = vl2[i];
    if (iters != nullptr && t == 0) iters[prob] = it_done;
    DQQ_WG_SYNC();
}

// Solver::iterative_refinement (Solver.cpp:15-44) for the rows x m system whose TRANSPOSED matrix sits in At and
// right-hand side in dd.  The solution is left in xs (m entries); At is overwritten (Cholesky workspace).
static DQQ_D void any_ir(double* At, double* K, double* Kinv, const double* dd, double* vAb, double* vKAb, double* xs,
                         double* vb, int m, int ld, int t, int& steps, int rows)
{
#pragma clang fp contract(off)
    for (int i = t; i < m; i += kAnyT) {
        double Ab = 0.0;
        for (int k = 0; k < rows; ++k) Ab += At[k * ld + i] * dd[k];           // A^T b, :19
        vAb[i] = Ab;
    }
    // A^T A + mu I (:20-21): one entry per thread and trip (m^2 entries over 256 threads; a row per thread would
    // leave all but m of them idle), each entry summed over k in index order as before
    for (long idx = t; idx < (long)m * m; idx += kAnyT) {
        const int i = (int)(idx / m), j = (int)(idx % m);
        double s = 0.0;
#pragma unroll 8
        for (int k = 0; k < rows; ++k) s += At[k * ld + i] * At[k * ld + j];
        K[i * ld + j] = (i == j) ? s + kMuIr : s;
    }
    DQQ_WG_SYNC();
    for (long idx = t; idx < (long)m * m; idx += kAnyT) At[(idx / m) * ld + idx % m] = K[(idx / m) * ld + idx % m];
    DQQ_WG_SYNC();
    chol_inverse_wg(At, Kinv, m, ld, t);                                        // :22-23
    for (int i = t; i < m; i += kAnyT) {
        vKAb[i] = any_row_dot(Kinv, ld, i, vAb, m);                             // :27
        xs[i] = 0.0;
    }
    DQQ_WG_SYNC();
    IrControl ctl;
    ctl.init();
    steps = 0;
    for (int it = 0; it < kIrMaxIter; ++it) {
        steps = it + 1;
        for (int i = t; i < m; i += kAnyT) vb[i] = kMuIr * any_row_dot(Kinv, ld, i, xs, m) + vKAb[i]; // :29
        DQQ_WG_SYNC();
        for (int i = t; i < m; i += kAnyT) xs[i] = vb[i];
        DQQ_WG_SYNC();
        for (int i = t; i < m; i += kAnyT) vb[i] = any_row_dot(K, ld, i, xs, m) - vAb[i];             // :30
        DQQ_WG_SYNC();
        const double res = sqrt(any_sumsq(vb, m));                              // :31
        DQQ_WG_SYNC();
        if (ctl.update(res)) break;                                             // :32-41
    }
}

// One problem, backward, by one workgroup: QP (KIND 0), QCQP (1), box QP (2; l_n = l_min, mu_c = l_max,
// grad_l_n = grad_l_min, grad_mu = grad_l_max).
template <int KIND>
static DQQ_D void any_bwd_problem(const double* __restrict__ P, const double* __restrict__ q,
                                  const double* __restrict__ l_n, const double* __restrict__ mu_c,
                                  const double* __restrict__ x, const double* __restrict__ grad_x,
                                  double* __restrict__ grad_P, double* __restrict__ grad_q, double* __restrict__ grad_l_n,
                                  double* __restrict__ grad_mu, double* __restrict__ gamma_out,
                                  double* __restrict__ dgamma_out, int* __restrict__ ir_steps, long prob, int n,
                                  double dual_eps, double* At, double* K, double* Kinv, double* vec, int t)
{
    // At, Kinv: LDS when they fit, else global; K (holds P, row stride ld, while a system is assembled) and vec
    // (any_bwd_vec_doubles) in global memory
#pragma clang fp contract(off)
    const int nc = n / 2;
    const int mmax = any_bwd_rows(KIND, n);
    const int ld = mmax | 1;
    double* vdd = vec;                     // mmax: right-hand side
    double* vAb = vdd + mmax;
    double* vKAb = vAb + mmax;
    double* xs = vKAb + mmax;
    double* vb = xs + mmax;
    double* vgam = vb + mmax;              // QP / QCQP: n ; box: 2n (lower | upper)
    double* vdg = vgam + 2 * mmax;         // box: 2n dgamma ; QCQP: nc dgamma
    double* vS = vdg + 2 * mmax;           // QCQP: nc
    double* vdl = vS + mmax;               // n
    double* va = vdl + n;                  // n
    int* perm = reinterpret_cast<int*>(va + n);  // up to 2n + 2 ints
    int* counts = perm + 2 * n + 2;        // [0] = number of active entries
    const double* Pg = P + prob * (long)n * n;
    const double* xg = x + prob * (long)n;
    const double* gg = grad_x + prob * (long)n;
    const double* qg = q + prob * (long)n;
    double* Pl = K;
    any_load_matrix(Pl, ld, Pg, n, t);
    DQQ_WG_SYNC();
    int steps = 0, steps_dual = 1;

    if (KIND == 0) {
        // dualFromPrimalQP, Solver.cpp:125-134
        for (int i = t; i < n; i += kAnyT) {
            double gamma = -(any_row_dot(Pl, ld, i, xg, n) + qg[i]);
            if (xg[i] > dual_eps) gamma = 0;
            vgam[i] = gamma;
        }
        DQQ_WG_SYNC();
        if (t == 0) { // the order (active..., inactive...) of :139-147
            int na = 0;
            for (int i = 0; i < n; ++i) if (vgam[i] < -kActiveEps) perm[na++] = i;
            int p = na;
            for (int i = 0; i < n; ++i) if (!(vgam[i] < -kActiveEps)) perm[p++] = i;
            counts[0] = na;
        }
        DQQ_WG_SYNC();
        const int na = counts[0], m = n;
        // A = [[diag(l_A), 0],[0, P_II]]; At = A^T, :148-174
        for (long idx = t; idx < (long)m * m; idx += kAnyT) {
            const int r = (int)(idx / m), c = (int)(idx % m); // At[r][c] = A[c][r]
            double val;
            if (c < na || r < na) val = (c == r) ? xg[perm[c]] : 0.0;
            else val = Pl[perm[c] * ld + perm[r]];
            At[r * ld + c] = val;
        }
        for (int p = t; p < m; p += kAnyT) vdd[p] = (p < na) ? 0.0 : gg[perm[p]];  // :175-184
        DQQ_WG_SYNC();
        any_ir(At, K, Kinv, vdd, vAb, vKAb, xs, vb, m, ld, t, steps, m);           // :186
        for (int p = t; p < m; p += kAnyT) vdl[perm[p]] = (p < na) ? 0.0 : xs[p]; // :187-191
        DQQ_WG_SYNC();
    } else if (KIND == 1) {
        // dualFromPrimalQCQP, Solver.cpp:584-617, per contact
        for (int i = t; i < n; i += kAnyT) va[i] = any_row_dot(Pl, ld, i, xg, n) + qg[i];
        DQQ_WG_SYNC();
        for (int c = t; c < nc; c += kAnyT) {
            const double ln = l_n[prob * (long)nc + c], mc = mu_c[prob * (long)nc + c];
            const double r = ln * mc;                                             // pybindings.cpp:65
            const double xa = xg[2 * c], xb = xg[2 * c + 1];
            double gamma = 0.0;
            const double slack = r + -sqrt(xa * xa + xb * xb);
            if (!(slack > dual_eps || r < dual_eps)) {
                const double ca = 2 * xa, cb = 2 * xb;
                const double G = ca * ca + cb * cb;
                const double rhs = ca * va[2 * c] + cb * va[2 * c + 1];
                const double L = sqrt(G);
                gamma = -((rhs / L) / L);
            }
            double S = -(r * r);                                                  // Solver.cpp:622-629
            S = S + (xa * xa + xb * xb);
            vgam[c] = gamma;
            vS[c] = S;
            vdg[c] = 0.0;
        }
        DQQ_WG_SYNC();
        if (t == 0) { // :637-641
            int na = 0;
            for (int c = 0; c < nc; ++c) {
                const double r = l_n[prob * (long)nc + c] * mu_c[prob * (long)nc + c];
                if (vS[c] > -kActiveEps && r > kActiveEps) perm[na++] = c;
            }
            counts[0] = na;
        }
        DQQ_WG_SYNC();
        const int na = counts[0], m = n + na;
        // A = [[diag(S_act), (diag(gamma) C^T)_act],[C_act, P + blkdiag(2 gamma_i I2)]]; At = A^T, :643-657
        for (long idx = t; idx < (long)m * m; idx += kAnyT) {
            const int rr = (int)(idx / m), cc = (int)(idx % m);
            const int row = cc, col = rr;
            double val;
            if (row < na) {
                const int cid = perm[row];
                if (col < na) val = (col == row) ? vS[cid] : 0.0;
                else { const int i = col - na; val = (i / 2 == cid) ? vgam[cid] * (2 * xg[i]) : 0.0; }
            } else {
                const int i = row - na;
                if (col < na) { const int cid = perm[col]; val = (i / 2 == cid) ? 2 * xg[i] : 0.0; }
                else { const int jj = col - na; const double d = (i == jj) ? 2 * vgam[i / 2] : 0.0; val = d + Pl[i * ld + jj]; }
            }
            At[rr * ld + cc] = val;
        }
        for (int p = t; p < m; p += kAnyT) vdd[p] = (p < na) ? 0.0 : gg[p - na];    // :659-667
        DQQ_WG_SYNC();
        any_ir(At, K, Kinv, vdd, vAb, vKAb, xs, vb, m, ld, t, steps, m);           // :669
        for (int p = t; p < m; p += kAnyT) {                                       // blgamma scatter, :670-679
            if (p < na) vdg[perm[p]] = xs[p];
            else vdl[p - na] = xs[p];
        }
        DQQ_WG_SYNC();
        for (int c = t; c < nc; c += kAnyT) {
            const double ln = l_n[prob * (long)nc + c], mc = mu_c[prob * (long)nc + c];
            const double gamma = vgam[c], dg = vdg[c];
            if (grad_l_n != nullptr) grad_l_n[prob * (long)nc + c] = QcqpContact::e2(gamma, ln, mc) * dg;
            if (grad_mu != nullptr) grad_mu[prob * (long)nc + c] = QcqpContact::e1(gamma, ln, mc) * dg;
            if (gamma_out != nullptr) gamma_out[prob * (long)nc + c] = gamma;
            if (dgamma_out != nullptr) dgamma_out[prob * (long)nc + c] = dg;
        }
    } else {
        // box QP: Solver::dualFromPrimalBoxQP (Solver.cpp:263-308) + solveDerivativesBoxQP (:310-371)
        const double* lo = l_n + prob * (long)n;
        const double* hi = mu_c + prob * (long)n;
        for (int i = t; i < 2 * n; i += kAnyT) { vgam[i] = 0.0; vdg[i] = 0.0; }
        if (t == 0) { // not_null bookkeeping, :268-283 / :315-327: per coordinate, lower before upper
            int nn = 0;
            for (int i = 0; i < n; ++i) {
                if (!(xg[i] - lo[i] > dual_eps)) perm[nn++] = i;
                if (!(xg[i] - hi[i] < -dual_eps)) perm[nn++] = n + i;
            }
            counts[0] = nn;
        }
        DQQ_WG_SYNC();
        const int nn = counts[0];
        // gamma_not_null = iterative_refinement(Id2, -P*l - q), :291-304
        for (int i = t; i < n; i += kAnyT) {
            double rhs = 0.0;
            for (int j = 0; j < n; ++j) rhs += (-Pl[i * ld + j]) * xg[j];
            vdd[i] = rhs - qg[i];
        }
        for (long idx = t; idx < (long)n * nn; idx += kAnyT) { // Id2: n x nn; At[i][p] = Id2(i, p)
            const int i = (int)(idx / nn), p = (int)(idx % nn), id = perm[p];
            At[i * ld + p] = (id < n) ? ((id == i) ? -1.0 : 0.0) : ((id - n == i) ? 1.0 : 0.0);
        }
        DQQ_WG_SYNC();
        if (nn > 0) { // nn == 0: the reference runs one loop body on empty vectors
            any_ir(At, K, Kinv, vdd, vAb, vKAb, xs, vb, nn, ld, t, steps_dual, n);
            for (int p = t; p < nn; p += kAnyT) vgam[perm[p]] = xs[p];
        }
        DQQ_WG_SYNC();
        // A = [[0, B],[Id2, P]], B.row(j) = gamma_j * Id2.col(j)^T, :341-369
        const int m = nn + n;
        any_load_matrix(Pl, ld, Pg, n, t); // the refinement above used K as workspace
        DQQ_WG_SYNC();
        for (long idx = t; idx < (long)m * m; idx += kAnyT) {
            const int rr = (int)(idx / m), cc = (int)(idx % m);
            const int row = cc, col = rr;
            double val = 0.0;
            if (row < nn) {
                if (col >= nn) {
                    const int id = perm[row], i = col - nn;
                    const double s = (id < n) ? ((id == i) ? -1.0 : 0.0) : ((id - n == i) ? 1.0 : 0.0);
                    val = vgam[id] * s;
                }
            } else {
                const int i = row - nn;
                if (col < nn) {
                    const int id = perm[col];
                    val = (id < n) ? ((id == i) ? -1.0 : 0.0) : ((id - n == i) ? 1.0 : 0.0);
                } else {
                    val = Pl[i * ld + (col - nn)];
                }
            }
            At[rr * ld + cc] = val;
        }
        for (int p = t; p < m; p += kAnyT) vdd[p] = (p < nn) ? 0.0 : gg[p - nn];   // :352-360
        DQQ_WG_SYNC();
        any_ir(At, K, Kinv, vdd, vAb, vKAb, xs, vb, m, ld, t, steps, m);          // :362
        for (int p = t; p < m; p += kAnyT) {
            if (p < nn) vdg[perm[p]] = xs[p];                                      // :363-366
            else vdl[p - nn] = xs[p];                                              // :367-369
        }
        DQQ_WG_SYNC();
        for (int i = t; i < n; i += kAnyT) {
            const double glo = vgam[i], ghi = vgam[n + i], dlo = vdg[i], dhi = vdg[n + i];
            if (grad_l_n != nullptr) grad_l_n[prob * (long)n + i] = -(dlo * glo);
            if (grad_mu != nullptr) grad_mu[prob * (long)n + i] = dhi * ghi;
            if (gamma_out != nullptr) { gamma_out[prob * 2L * n + i] = glo; gamma_out[prob * 2L * n + n + i] = ghi; }
            if (dgamma_out != nullptr) { dgamma_out[prob * 2L * n + i] = dlo; dgamma_out[prob * 2L * n + n + i] = dhi; }
        }
    }
    if (grad_q != nullptr) for (int i = t; i < n; i += kAnyT) grad_q[prob * (long)n + i] = -vdl[i];
    if (grad_P != nullptr) {
        double* Gp = grad_P + prob * (long)n * n;
        for (long idx = t; idx < (long)n * n; idx += kAnyT) Gp[idx] = -(vdl[idx / n] * xg[idx % n]);
    }
    if (ir_steps != nullptr && t == 0) {
        if (KIND == 2) { ir_steps[2 * prob] = steps_dual; ir_steps[2 * prob + 1] = steps; }
        else ir_steps[prob] = steps;
    }
    DQQ_WG_SYNC();
}

} // namespace dqq
