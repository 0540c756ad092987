/*
 * diffqcqp_oracle.c -- CPU restatement of the reference ADMM QP/QCQP solver.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference (quentinll/diffqcqp @ /root/reference) cannot
 * be built in this image (Eigen is absent, the pybind11 submodule is empty)
 * and none of its scripts hold an expected value.  This file is therefore a
 * from-scratch dense restatement of qcqplib/Solver.cpp in plain C, following
 * the reference statement by statement; what pins it is listed in
 * oracle/README.md (closed forms, KKT residuals, finite differences, and the
 * seed-5 inputs the reference left as four-digit constants in comments).  Third-party arithmetic restated here:
 * Eigen3 (version unpinned by the reference's CMake): LLT = unblocked lower
 * Cholesky, solveInPlace(Identity) = forward + backward substitution per
 * column, normalize() = divide by the 2-norm when it is > 0, norm() = sqrt of
 * the plain sum of squares, lpNorm<Infinity> = max |.|.
 *
 * All matrices are row-major double.  Build with -ffp-contract=off so that the
 * operation order written here is the operation order executed.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference).
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- helpers */

/* Eigen LLT (lower) + solveInPlace(Identity): qcqplib/Solver.cpp:23,76-77,
 * 100-101,114-115,535-536,559-560,573-574.  A (n x n, symmetric, only the
 * lower triangle is read) -> Ainv (full symmetric-by-construction result of the
 * two triangular solves).  L is n*n scratch.  Success is never checked by the
 * reference (Solver.cpp:76), so a non-PD input simply yields NaNs. */
static void chol_inverse(const double *A, int n, double *Ainv, double *L)
{
    int i, j, k, c;
    for (i = 0; i < n * n; ++i) L[i] = 0.0;
    for (k = 0; k < n; ++k) {
        double x = A[k * n + k];
        double s = 0.0;
        for (j = 0; j < k; ++j) s += L[k * n + j] * L[k * n + j];
        x = x - s;
        x = sqrt(x);
        L[k * n + k] = x;
        for (i = k + 1; i < n; ++i) {
            double t = 0.0;
            for (j = 0; j < k; ++j) t += L[i * n + j] * L[k * n + j];
            L[i * n + k] = (A[i * n + k] - t) / x;
        }
    }
    /* column c of the inverse: L y = e_c, then L^T x = y */
    for (c = 0; c < n; ++c) {
        for (i = 0; i < n; ++i) {
            double t = (i == c) ? 1.0 : 0.0;
            for (j = 0; j < i; ++j) t -= L[i * n + j] * Ainv[j * n + c];
            Ainv[i * n + c] = t / L[i * n + i];
        }
        for (i = n - 1; i >= 0; --i) {
            double t = Ainv[i * n + c];
            for (j = i + 1; j < n; ++j) t -= L[j * n + i] * Ainv[j * n + c];
            Ainv[i * n + c] = t / L[i * n + i];
        }
    }
}

static void matvec(const double *A, int rows, int cols, const double *v, double *out)
{
    int i, j;
    for (i = 0; i < rows; ++i) {
        double s = 0.0;
        for (j = 0; j < cols; ++j) s += A[i * cols + j] * v[j];
        out[i] = s;
    }
}

static double norm2(const double *v, int n)
{
    double s = 0.0;
    int i;
    for (i = 0; i < n; ++i) s += v[i] * v[i];
    return sqrt(s);
}

static void add_to_diag(double *A, int n, double d)
{
    int i;
    for (i = 0; i < n; ++i) A[i * n + i] += d;
}

/* Solver::power_iteration, qcqplib/Solver.cpp:46-59.  `epsilon` is unused by
 * the reference.  Returns the Rayleigh quotient after exactly max_iter steps. */
static double power_iteration(const double *A, int n, int max_iter, double *v, double *Av)
{
    int i, k;
    double c = 1 / sqrt((double)n), l_max;
    for (i = 0; i < n; ++i) v[i] = c;
    { /* v.normalize(): divide by the norm only when the squared norm is > 0 */
        double s = 0.0;
        for (i = 0; i < n; ++i) s += v[i] * v[i];
        if (s > 0) {
            double nn = sqrt(s);
            for (i = 0; i < n; ++i) v[i] /= nn;
        }
    }
    for (k = 0; k < max_iter; ++k) {
        matvec(A, n, n, v, Av);
        for (i = 0; i < n; ++i) v[i] = Av[i];
        {
            double s = 0.0;
            for (i = 0; i < n; ++i) s += v[i] * v[i];
            if (s > 0) {
                double nn = sqrt(s);
                for (i = 0; i < n; ++i) v[i] /= nn;
            }
        }
    }
    matvec(A, n, n, v, Av);
    l_max = 0.0;
    for (i = 0; i < n; ++i) l_max += v[i] * Av[i];
    return l_max;
}

/* Solver::prox_circle, qcqplib/Solver.cpp:505-519 */
static void prox_circle(double *l, const double *l_n, int nc)
{
    int i;
    for (i = 0; i < nc; ++i) {
        double a = l[2 * i], b = l[2 * i + 1];
        double nrm = sqrt(a * a + b * b);
        if (nrm > l_n[i]) {
            l[2 * i] = a * l_n[i] / nrm;
            l[2 * i + 1] = b * l_n[i] / nrm;
        }
    }
}

/* --------------------------------------------------------------- forward */

/* Solver::solveQP (kind 0), qcqplib/Solver.cpp:61-123, Solver::solveQCQP
 * (kind 1), qcqplib/Solver.cpp:521-582, Solver::solveBoxQP (kind 2), :198-261,
 * and Solver::solveSignedBoxQP (kind 3), :374-439.  The two box solvers are the
 * QP loop with another projection (:219-220 / :396-398); everything else --
 * 10 power steps, symmetric tau damping, dual-residual-only stop -- is the QP's.
 * `radius` is l_n o mu (pybindings.cpp:57) for kind 1; `l_min`, `l_max` for
 * kinds 2, 3; `v` for kind 3.  warm_start is accepted by the reference
 * and overwritten before it is ever read (Solver.cpp:70/80, 529/539), so it is
 * not a parameter here.  Returns the number of ADMM iterations executed. */
static int admm_solve_ex(int kind, const double *P_in, const double *q, const double *radius,
                         const double *l_min, const double *l_max, const double *v_in, int n, double epsilon,
                         double mu_prox, int max_iter, int adaptative_rho, double *x_out,
                         double *ws /* 2n^2 + n^2 + 8n */)
{
    const int qp_like = (kind != 1);
    const double mu_thresh = 10., alpha_relax = 1.5, eps_rel = 1e-4;
    double *P = ws, *Pinv = P + n * n, *L = Pinv + n * n;
    double *q_prox = L + n * n, *l = q_prox + n, *l_2 = l + n, *l_2_pred = l_2 + n, *u = l_2_pred + n,
           *rhs = u + n, *tmp = rhs + n, *Plqu = tmp + n;
    double Lmax, rho, res_dual, res_prim, tau_inc, tau_dec;
    int i, it, rho_up = 0, cpt = 0, iters = 0;
    memcpy(P, P_in, sizeof(double) * n * n); /* MatrixXd P by value, :61 / :521 */
    for (i = 0; i < n; ++i) { u[i] = 0; l_2[i] = 0; l_2_pred[i] = 0; }
    Lmax = power_iteration(P, n, qp_like ? 10 : 100, rhs, tmp); /* :71 / :530 / :209 / :385 */
    rho = sqrt(mu_prox * Lmax) * pow(Lmax / mu_prox, .4);          /* :72 / :531 */
    tau_inc = pow(Lmax / mu_prox, .15);                            /* :73 / :532 */
    tau_dec = tau_inc;
    for (i = 0; i < n; ++i) q_prox[i] = q[i];                      /* :74 / :533 */
    add_to_diag(P, n, rho + mu_prox);                              /* :75 / :534 */
    chol_inverse(P, n, Pinv, L);                                   /* :76-77 */
    for (it = 0; it < max_iter; ++it) {
        double rd = 0, rp = 0;
        int stop_prim_ok = 1;
        iters = it + 1;
        for (i = 0; i < n; ++i) rhs[i] = rho * l_2[i] - u[i] - q_prox[i];
        matvec(Pinv, n, n, rhs, l);                                /* :80 / :539 */
        for (i = 0; i < n; ++i) q_prox[i] = q[i] - mu_prox * l[i]; /* :81 / :540 */
        if (kind == 0) {
            for (i = 0; i < n; ++i) {                              /* :82 */
                double t = alpha_relax * l[i] + (1 - alpha_relax) * l_2[i] + u[i] / rho;
                l_2[i] = t < 0 ? 0 : t;                            /* cwiseMax(0) = std::max(t,0) */
            }
        } else if (kind == 2 || kind == 3) {
            for (i = 0; i < n; ++i) {                              /* :219-220 / :396-397 */
                double t = alpha_relax * l[i] + (1 - alpha_relax) * l_2[i] + u[i] / rho;
                t = t < l_min[i] ? l_min[i] : t;                   /* cwiseMax(l_min) */
                t = l_max[i] < t ? l_max[i] : t;                   /* cwiseMin(l_max) */
                if (kind == 3) {                                   /* :395, :398: v = sign(v); l_2 = v o min(v o l_2, 0) */
                    const double sg = (double)((v_in[i] > 0) - (v_in[i] < 0));
                    double m = sg * t;
                    m = 0 < m ? 0 : m;
                    t = sg * m;
                }
                l_2[i] = t;
            }
        } else {
            for (i = 0; i < n; ++i)                                /* :541 */
                l_2[i] = alpha_relax * l[i] + (1 - alpha_relax) * l_2[i] + u[i] / rho;
            prox_circle(l_2, radius, n / 2);                       /* :542 */
        }
        for (i = 0; i < n; ++i)                                    /* :83 / :543 */
            u[i] += rho * (alpha_relax * l[i] + (1 - alpha_relax) * l_2_pred[i] - l_2[i]);
        if (qp_like) {
            for (i = 0; i < n; ++i) {                              /* :84-85 */
                Plqu[i] = rho * (l_2[i] - l_2_pred[i]);
                if (fabs(Plqu[i]) > rd) rd = fabs(Plqu[i]);
            }
            res_dual = rd;
        } else {
            for (i = 0; i < n; ++i) {                              /* :544-545 */
                Plqu[i] = l_2[i] - l_2_pred[i];
                if (fabs(Plqu[i]) > rd) rd = fabs(Plqu[i]);
            }
            res_dual = rho * rd;
        }
        for (i = 0; i < n; ++i) {                                  /* :86 / :546 */
            double t = fabs(l_2[i] - (alpha_relax * l[i] + (1 - alpha_relax) * l_2_pred[i]));
            if (t > rp) rp = t;
        }
        res_prim = rp;
        for (i = 0; i < n; ++i) l_2_pred[i] = l_2[i];              /* :87 / :547 */
        if (kind == 1) stop_prim_ok = res_prim < epsilon + eps_rel * norm2(l, n); /* :548 */
        if (stop_prim_ok && res_dual < epsilon) break;             /* :88 / :548 */
        if (adaptative_rho) {
            if (res_prim > mu_thresh * res_dual) {                 /* :92 / :552 */
                if (cpt % 5 == 0) {
                    if (rho_up == -1) {
                        tau_inc = 1 + .8 * (tau_inc - 1);
                        if (qp_like) tau_dec = 1 + .8 * (tau_dec - 1); /* :94-97 vs :554-556 */
                    }
                    add_to_diag(P, n, rho * (tau_inc - 1));
                    rho *= tau_inc;
                    chol_inverse(P, n, Pinv, L);
                    rho_up = 1;
                }
                cpt++;
            } else if (res_dual > mu_thresh * res_prim) {          /* :106 / :566 */
                if (cpt % 5 == 0) {
                    if (rho_up == 1) {
                        if (qp_like) tau_inc = 1 + .8 * (tau_inc - 1); /* :108-111 vs :568-570 */
                        tau_dec = 1 + .8 * (tau_dec - 1);
                    }
                    add_to_diag(P, n, rho * (1. / tau_dec - 1));
                    rho /= tau_dec;
                    chol_inverse(P, n, Pinv, L);
                    rho_up = -1;
                }
                cpt++;
            }
        }
    }
    for (i = 0; i < n; ++i) x_out[i] = l_2[i];
    return iters;
}

static int admm_solve(int kind, const double *P_in, const double *q, const double *radius, int n,
                      double epsilon, double mu_prox, int max_iter, int adaptative_rho, double *x_out, double *ws)
{
    return admm_solve_ex(kind, P_in, q, radius, NULL, NULL, NULL, n, epsilon, mu_prox, max_iter, adaptative_rho,
                         x_out, ws);
}

/* -------------------------------------------------------------- backward */

/* Solver::iterative_refinement, qcqplib/Solver.cpp:15-44, with its defaults
 * mu_ir=1e-7, epsilon=1e-10, max_iter=10.  A is n x n.  Returns the number of
 * loop bodies executed.  ws: 3 n^2 + 4 n. */
/* Test knob (not in the reference): > 0 makes the refinement loop below run exactly that many bodies,
 * ignoring its exit tests.  The reference's exit (Solver.cpp:32-41) is decided by rounding noise; this lets a test
 * evaluate the reference formula at the OTHER exit and show that a kernel which left the loop after 3 bodies
 * where the oracle left after 1 (or vice versa) still returned what the reference computes at that exit. */
static int g_force_ir_steps = 0;
ORC_API void orc_set_force_ir_steps(int steps) { g_force_ir_steps = steps; }

static int iterative_refinement_rect(const double *A, const double *b, int rows, int n, double *x, double *ws)
{
    /* A is rows x n (row-major); the unknown has n = A.cols() entries.  ws: 3 n^2 + 4 n. */
    const double mu_ir = 1e-7, epsilon = 1e-10;
    const int max_iter = 10;
    double *K = ws, *Kinv = K + n * n, *L = Kinv + n * n;
    double *Ab = L + n * n, *KinvAb = Ab + n, *tmp = KinvAb + n, *delta = tmp + n;
    int i, j, k, it, not_improved = 0, steps = 0;
    double res, res_pred = DBL_MAX;
    for (i = 0; i < n; ++i) x[i] = 0;
    for (i = 0; i < n; ++i) { /* Ab = A^T b, :19 */
        double s = 0;
        for (k = 0; k < rows; ++k) s += A[k * n + i] * b[k];
        Ab[i] = s;
    }
    for (i = 0; i < n; ++i)   /* AA_tild = A^T A (+ mu_ir I), :20-21 */
        for (j = 0; j < n; ++j) {
            double s = 0;
            for (k = 0; k < rows; ++k) s += A[k * n + i] * A[k * n + j];
            K[i * n + j] = s;
        }
    add_to_diag(K, n, mu_ir);
    if (n > 0) chol_inverse(K, n, Kinv, L); /* :22-23 */
    matvec(Kinv, n, n, Ab, KinvAb);         /* :27 */
    for (it = 0; it < max_iter; ++it) {
        steps = it + 1;
        matvec(Kinv, n, n, x, tmp);         /* x = mu_ir*Kinv*x + KinvAb, :29 */
        for (i = 0; i < n; ++i) x[i] = mu_ir * tmp[i] + KinvAb[i];
        matvec(K, n, n, x, delta);          /* :30-31 */
        for (i = 0; i < n; ++i) delta[i] = delta[i] - Ab[i];
        res = norm2(delta, n);
        if (res_pred - res < epsilon) {     /* :32-38 */
            not_improved++;
        } else {
            res_pred = res;
            not_improved = 0;
        }
        if (g_force_ir_steps > 0) { if (steps >= g_force_ir_steps) break; else continue; }
        if (res < epsilon || not_improved == 2) break; /* :39 */
    }
    return steps;
}

static int iterative_refinement(const double *A, const double *b, int n, double *x, double *ws)
{
    return iterative_refinement_rect(A, b, n, n, x, ws);
}

/* Solver::dualFromPrimalQP, qcqplib/Solver.cpp:125-134 */
static void dual_from_primal_qp(const double *P, const double *q, const double *l, int n, double epsilon,
                                double *gamma)
{
    int i;
    matvec(P, n, n, l, gamma);
    for (i = 0; i < n; ++i) {
        gamma[i] = -(gamma[i] + q[i]);
        if (l[i] > epsilon) gamma[i] = 0;
    }
}

/* Solver::solveDerivativesQP, qcqplib/Solver.cpp:136-196.  ws: 4 n^2 + 8 n. */
static int solve_derivatives_qp(const double *P, const double *l, const double *gamma, const double *grad_l,
                                int n, double *bl, double *ws)
{
    double *A = ws, *dd = A + n * n, *b = dd + n, *irws = b + n;
    int *not_null = (int *)malloc(sizeof(int) * 2 * (n + 1)), *null_idx = not_null + n + 1;
    int na = 0, ni = 0, i, j, steps;
    for (i = 0; i < n; ++i) {                 /* :139-147 */
        if (gamma[i] < -1e-10) not_null[na++] = i; else null_idx[ni++] = i;
    }
    /* A = [[diag(l_A), B_tild],[C_tild, P_II]] with B_tild, C_tild taken from a
     * diagonal matrix / the identity on disjoint index sets => zeros (:148-173) */
    for (i = 0; i < n * n; ++i) A[i] = 0;
    for (i = 0; i < na; ++i) A[i * n + i] = l[not_null[i]];
    for (i = 0; i < ni; ++i)
        for (j = 0; j < ni; ++j) A[(na + i) * n + (na + j)] = P[null_idx[i] * n + null_idx[j]];
    /* A.transposeInPlace(), :174 */
    for (i = 0; i < n; ++i)
        for (j = i + 1; j < n; ++j) { double t = A[i * n + j]; A[i * n + j] = A[j * n + i]; A[j * n + i] = t; }
    for (i = 0; i < n; ++i) dd[i] = (i < na) ? 0. : grad_l[null_idx[i - na]]; /* :175-184 */
    steps = iterative_refinement(A, dd, n, b, irws);                            /* :186 */
    for (i = 0; i < n; ++i) bl[i] = 0;                                          /* :187-191 */
    for (i = 0; i < ni; ++i) bl[null_idx[i]] = b[na + i];
    free(not_null);
    return steps;
}

/* Solver::dualFromPrimalQCQP, qcqplib/Solver.cpp:584-617.  l_n is the radius
 * l_n o mu.  ws: n*nc + 3 nc^2 + n + 2 nc. */
static void dual_from_primal_qcqp(const double *P, const double *q, const double *l_n, const double *l, int n,
                                  double epsilon, double *gamma, double *ws)
{
    int nc = n / 2, i, j, k, na = 0;
    double *At = ws /* n x na */, *G = At + n * nc, *Ginv = G + nc * nc, *L = Ginv + nc * nc;
    double *Plq = L + nc * nc, *rhs = Plq + n, *sol = rhs + nc;
    int *not_null = (int *)malloc(sizeof(int) * (nc + 1));
    for (i = 0; i < nc; ++i) {                       /* :594-605 */
        double a = l[2 * i], b = l[2 * i + 1];
        double slack = l_n[i] + -sqrt(a * a + b * b);
        if (slack > epsilon || l_n[i] < epsilon) gamma[i] = 0; else { not_null[na++] = i; gamma[i] = 0; }
    }
    /* A_tild: columns of A (A(2i,i)=2 l(2i), A(2i+1,i)=2 l(2i+1)), :589-592,606-609 */
    for (i = 0; i < n * na; ++i) At[i] = 0;
    for (k = 0; k < na; ++k) {
        At[(2 * not_null[k]) * na + k] = 2 * l[2 * not_null[k]];
        At[(2 * not_null[k] + 1) * na + k] = 2 * l[2 * not_null[k] + 1];
    }
    matvec(P, n, n, l, Plq);
    for (i = 0; i < n; ++i) Plq[i] = Plq[i] + q[i];
    for (i = 0; i < na; ++i) {                       /* A_tild^T A_tild and A_tild^T (Pl+q), :611 */
        double s = 0;
        for (j = 0; j < na; ++j) {
            double t = 0;
            for (k = 0; k < n; ++k) t += At[k * na + i] * At[k * na + j];
            G[i * na + j] = t;
        }
        for (k = 0; k < n; ++k) s += At[k * na + i] * Plq[k];
        rhs[i] = s;
    }
    if (na > 0) {
        /* llt().solve(rhs): L y = rhs, L^T x = y (not an explicit inverse), :611 */
        for (i = 0; i < na * na; ++i) L[i] = 0;
        for (k = 0; k < na; ++k) {
            double x = G[k * na + k], s = 0;
            for (j = 0; j < k; ++j) s += L[k * na + j] * L[k * na + j];
            x = sqrt(x - s);
            L[k * na + k] = x;
            for (i = k + 1; i < na; ++i) {
                double t = 0;
                for (j = 0; j < k; ++j) t += L[i * na + j] * L[k * na + j];
                L[i * na + k] = (G[i * na + k] - t) / x;
            }
        }
        for (i = 0; i < na; ++i) {
            double t = rhs[i];
            for (j = 0; j < i; ++j) t -= L[i * na + j] * sol[j];
            sol[i] = t / L[i * na + i];
        }
        for (i = na - 1; i >= 0; --i) {
            double t = sol[i];
            for (j = i + 1; j < na; ++j) t -= L[j * na + i] * sol[j];
            sol[i] = t / L[i * na + i];
        }
        for (i = 0; i < na; ++i) gamma[not_null[i]] = -sol[i]; /* :611-616 */
    }
    free(not_null);
}

/* Solver::getE12QCQP, qcqplib/Solver.cpp:683-691 (diagonals only; raw l_n, mu) */
static void get_e12_qcqp(const double *l_n, const double *mu, const double *gamma, int nc, double *e1,
                         double *e2)
{
    int i;
    for (i = 0; i < nc; ++i) {
        e1[i] = 2 * gamma[i] * l_n[i] * l_n[i] * mu[i];
        e2[i] = 2 * gamma[i] * l_n[i] * mu[i] * mu[i];
    }
}

/* Solver::solveDerivativesQCQP, qcqplib/Solver.cpp:619-681.  l_n is the radius.
 * blgamma = [dgamma (nc); dl (n)].  ws: 4 m^2 + 8 m with m = n + nc. */
static int solve_derivatives_qcqp(const double *P, const double *l_n, const double *l, const double *gamma,
                                  const double *grad_l, int n, double *blgamma, double *ws)
{
    int nc = n / 2, i, j, na = 0, m, steps;
    int *not_null = (int *)malloc(sizeof(int) * (nc + 1));
    double *slack = (double *)malloc(sizeof(double) * (nc + 1));
    double *A, *dd, *b, *irws;
    for (i = 0; i < nc; ++i) {                               /* :621-634 */
        double a = l[2 * i], bb = l[2 * i + 1];
        slack[i] = -(l_n[i] * l_n[i]);
        slack[i] = slack[i] + (a * a + bb * bb);
    }
    for (i = 0; i < nc; ++i)                                 /* :637-642 */
        if (slack[i] > -1e-10 && l_n[i] > 1e-10) not_null[na++] = i;
    m = n + na;
    A = ws; dd = A + m * m; b = dd + m; irws = b + m;
    for (i = 0; i < m * m; ++i) A[i] = 0;
    for (i = 0; i < na; ++i) {                               /* :643-657 */
        int c = not_null[i];
        A[i * m + i] = slack[c];                             /* A_tild */
        A[i * m + na + 2 * c] = gamma[c] * (2 * l[2 * c]);   /* B_tild = (diag(gamma) C^T) rows */
        A[i * m + na + 2 * c + 1] = gamma[c] * (2 * l[2 * c + 1]);
        A[(na + 2 * c) * m + i] = 2 * l[2 * c];              /* C_tild */
        A[(na + 2 * c + 1) * m + i] = 2 * l[2 * c + 1];
    }
    for (i = 0; i < n; ++i)                                  /* D_tild + P, :632-633,651,656 */
        for (j = 0; j < n; ++j) {
            double d = (i == j) ? 2 * gamma[i / 2] : 0.0;
            A[(na + i) * m + na + j] = d + P[i * n + j];
        }
    for (i = 0; i < m; ++i)                                  /* transposeInPlace, :657 */
        for (j = i + 1; j < m; ++j) { double t = A[i * m + j]; A[i * m + j] = A[j * m + i]; A[j * m + i] = t; }
    for (i = 0; i < m; ++i) dd[i] = (i < na) ? 0. : grad_l[i - na]; /* :659-667 */
    steps = iterative_refinement(A, dd, m, b, irws);                /* :669 */
    for (i = 0; i < nc + n; ++i) blgamma[i] = 0;                    /* :670-679 */
    for (i = 0; i < m; ++i) {
        if (i < na) blgamma[not_null[i]] = b[i]; else blgamma[nc - na + i] = b[i];
    }
    free(not_null);
    free(slack);
    return steps;
}

/* ------------------------------------------------------------ box QP backward
 * The index bookkeeping shared by Solver::dualFromPrimalBoxQP (:263-308) and
 * Solver::solveDerivativesBoxQP (:310-371): coordinate i contributes the lower
 * multiplier i when l_i - l_min_i <= eps and the upper multiplier n + i when
 * l_i - l_max_i >= -eps, in that (interleaved) order.  Returns their number. */
static int box_not_null(const double *l_min, const double *l_max, const double *l, int n, double epsilon,
                        int *not_null)
{
    int i, nn = 0;
    for (i = 0; i < n; ++i) {
        if (!(l[i] - l_min[i] > epsilon)) not_null[nn++] = i;        /* :268-274 / :315-320 */
        if (!(l[i] - l_max[i] < -epsilon)) not_null[nn++] = n + i;   /* :275-282 / :321-327 */
    }
    return nn;
}

/* Id2 (n x nn), :291-300 / :331-340: column j holds -1 (lower bound) or +1 (upper bound) in the row of
 * its coordinate. */
static void box_id2(const int *not_null, int nn, int n, double *Id2)
{
    int j;
    for (j = 0; j < n * nn; ++j) Id2[j] = 0;
    for (j = 0; j < nn; ++j) {
        if (not_null[j] < n) Id2[not_null[j] * nn + j] = -1;
        else Id2[(not_null[j] - n) * nn + j] = 1;
    }
}

/* Solver::dualFromPrimalBoxQP, qcqplib/Solver.cpp:263-308 (the std::cout loop of :287-289 is debugging
 * output and not reproduced).  gamma: 2n.  ws: n*2n + 2n + 3(2n)^2 + 4(2n).  Returns the refinement steps. */
static int dual_from_primal_box(const double *P, const double *q, const double *l_min, const double *l_max,
                                const double *l, int n, double epsilon, double *gamma, double *ws)
{
    double *Id2 = ws, *rhs = Id2 + 2 * n * n, *gnn = rhs + n, *irws = gnn + 2 * n;
    int *not_null = (int *)malloc(sizeof(int) * (2 * n + 1));
    int nn, i, j, steps = 1; /* nn == 0: the reference's loop runs one body on empty vectors (res = 0) and leaves */
    for (i = 0; i < 2 * n; ++i) gamma[i] = 0;
    nn = box_not_null(l_min, l_max, l, n, epsilon, not_null);
    box_id2(not_null, nn, n, Id2);
    for (i = 0; i < n; ++i) {                                          /* -P*l - q, :301 */
        double s = 0;
        for (j = 0; j < n; ++j) s += (-P[i * n + j]) * l[j];
        rhs[i] = s - q[i];
    }
    if (nn > 0) steps = iterative_refinement_rect(Id2, rhs, n, nn, gnn, irws);
    for (j = 0; j < nn; ++j) gamma[not_null[j]] = gnn[j];              /* :302-304 */
    free(not_null);
    return steps;
}

/* Solver::solveDerivativesBoxQP, qcqplib/Solver.cpp:310-371.  blgamma: 3n = [dgamma (2n, scattered); dl (n)].
 * ws: (3n)^2 + 2*3n + n*2n + 3(3n)^2 + 4*3n. */
static int solve_derivatives_box(const double *P, const double *l_min, const double *l_max, const double *l,
                                 const double *gamma, const double *grad_l, int n, double epsilon,
                                 double *blgamma, double *ws)
{
    const int mmax = 3 * n;
    double *A = ws, *dd = A + mmax * mmax, *b = dd + mmax, *Id2 = b + mmax, *irws = Id2 + 2 * n * n;
    int *not_null = (int *)malloc(sizeof(int) * (2 * n + 1));
    int nn, m, i, j, steps;
    nn = box_not_null(l_min, l_max, l, n, epsilon, not_null);
    m = nn + n;
    box_id2(not_null, nn, n, Id2);
    /* A = [[0, B],[Id2, P]], B.row(j) = gamma(not_null[j]) * Id2.col(j)^T, :341-350 */
    for (i = 0; i < m * m; ++i) A[i] = 0;
    for (j = 0; j < nn; ++j)
        for (i = 0; i < n; ++i) A[j * m + nn + i] = gamma[not_null[j]] * Id2[i * nn + j];
    for (i = 0; i < n; ++i) {
        for (j = 0; j < nn; ++j) A[(nn + i) * m + j] = Id2[i * nn + j];
        for (j = 0; j < n; ++j) A[(nn + i) * m + nn + j] = P[i * n + j];
    }
    for (i = 0; i < m; ++i)                                            /* A.transposeInPlace(), :351 */
        for (j = i + 1; j < m; ++j) { double t = A[i * m + j]; A[i * m + j] = A[j * m + i]; A[j * m + i] = t; }
    for (i = 0; i < m; ++i) dd[i] = (i < nn) ? 0. : grad_l[i - nn];     /* :352-360 */
    steps = iterative_refinement(A, dd, m, b, irws);                   /* :362 */
    for (i = 0; i < 3 * n; ++i) blgamma[i] = 0;                         /* :363-369 */
    for (j = 0; j < nn; ++j) blgamma[not_null[j]] = b[j];
    for (i = 0; i < n; ++i) blgamma[2 * n + i] = b[nn + i];
    free(not_null);
    return steps;
}

/* ---------------------------------------------- single-problem entry points
 * Same composition as the pybind11 module `diffqcqp` (pybindings.cpp:17-30,
 * 54-71).  Return value: iteration / refinement-step count (diagnostic the
 * reference does not expose). */

static size_t fwd_ws_doubles(int n) { return (size_t)3 * n * n + 8 * n + 16; }
static size_t bwd_ws_doubles(int n)
{
    size_t m = (size_t)n + n / 2 + 1;
    return 5 * m * m + 16 * m + 64;
}

ORC_API int orc_solveQP(const double *P, const double *q, const double *warm_start, int n, double epsilon,
                        double mu_prox, int max_iter, int adaptative_rho, double *x)
{
    double *ws = (double *)malloc(sizeof(double) * fwd_ws_doubles(n));
    int it;
    (void)warm_start; /* dead in the reference: Solver.cpp:70 then :80 */
    it = admm_solve(0, P, q, NULL, n, epsilon, mu_prox, max_iter, adaptative_rho, x, ws);
    free(ws);
    return it;
}

ORC_API int orc_solveQCQP(const double *P, const double *q, const double *l_n, const double *mu,
                          const double *warm_start, int n, double epsilon, double mu_prox, int max_iter,
                          int adaptative_rho, double *x)
{
    int nc = n / 2, i, it;
    double *ws = (double *)malloc(sizeof(double) * (fwd_ws_doubles(n) + nc + 1));
    double *mul_n = ws + fwd_ws_doubles(n);
    (void)warm_start;
    for (i = 0; i < nc; ++i) mul_n[i] = l_n[i] * mu[i]; /* pybindings.cpp:57 */
    it = admm_solve(1, P, q, mul_n, n, epsilon, mu_prox, max_iter, adaptative_rho, x, ws);
    free(ws);
    return it;
}

/* pybindings.cpp:24-30 */
ORC_API int orc_solveDerivativesQP(const double *P, const double *q, const double *l, const double *grad_l,
                                   int n, double epsilon, double *bl)
{
    double *ws = (double *)malloc(sizeof(double) * (bwd_ws_doubles(n) + n));
    double *gamma = ws + bwd_ws_doubles(n);
    int steps;
    dual_from_primal_qp(P, q, l, n, epsilon, gamma);
    steps = solve_derivatives_qp(P, l, gamma, grad_l, n, bl, ws);
    free(ws);
    return steps;
}

/* pybindings.cpp:62-71.  e1, e2: the nc diagonal entries of E1, E2;
 * blgamma: nc + n.  gamma_out (nc) may be NULL. */
ORC_API int orc_solveDerivativesQCQP(const double *P, const double *q, const double *l_n, const double *mu,
                                     const double *l, const double *grad_l, int n, double epsilon, double *e1,
                                     double *e2, double *blgamma, double *gamma_out)
{
    int nc = n / 2, i, steps;
    double *ws = (double *)malloc(sizeof(double) * (bwd_ws_doubles(n) + 2 * (nc + 1)));
    double *mul_n = ws + bwd_ws_doubles(n), *gamma = mul_n + nc + 1;
    for (i = 0; i < nc; ++i) mul_n[i] = l_n[i] * mu[i];
    dual_from_primal_qcqp(P, q, mul_n, l, n, epsilon, gamma, ws);
    get_e12_qcqp(l_n, mu, gamma, nc, e1, e2);
    steps = solve_derivatives_qcqp(P, mul_n, l, gamma, grad_l, n, blgamma, ws);
    if (gamma_out) for (i = 0; i < nc; ++i) gamma_out[i] = gamma[i];
    free(ws);
    return steps;
}

/* pybindings.cpp:32-37 */
ORC_API int orc_solveBoxQP(const double *P, const double *q, const double *l_min, const double *l_max,
                           const double *warm_start, int n, double epsilon, double mu_prox, int max_iter,
                           int adaptative_rho, double *x)
{
    double *ws = (double *)malloc(sizeof(double) * fwd_ws_doubles(n));
    int it;
    (void)warm_start; /* dead in the reference: Solver.cpp:208 then :217 */
    it = admm_solve_ex(2, P, q, NULL, l_min, l_max, NULL, n, epsilon, mu_prox, max_iter, adaptative_rho, x, ws);
    free(ws);
    return it;
}

/* pybindings.cpp:47-52 */
ORC_API int orc_solveSignedBoxQP(const double *P, const double *q, const double *l_min, const double *l_max,
                                 const double *v, const double *warm_start, int n, double epsilon,
                                 double mu_prox, int max_iter, int adaptative_rho, double *x)
{
    double *ws = (double *)malloc(sizeof(double) * fwd_ws_doubles(n));
    int it;
    (void)warm_start; /* dead in the reference: Solver.cpp:384 then :393 */
    it = admm_solve_ex(3, P, q, NULL, l_min, l_max, v, n, epsilon, mu_prox, max_iter, adaptative_rho, x, ws);
    free(ws);
    return it;
}

static size_t box_ws_doubles(int n)
{
    size_t m = (size_t)3 * n + 1;
    return 4 * m * m + 2 * (size_t)n * n + 16 * m + 64;
}

/* pybindings.cpp:39-45: returns (blgamma (3n), gamma (2n)).  steps_out (may be NULL): [0] refinement steps
 * of the dual recovery, [1] of the derivative system (also the return value). */
ORC_API int orc_solveDerivativesBoxQP(const double *P, const double *q, const double *l_min,
                                      const double *l_max, const double *l, const double *grad_l, int n,
                                      double epsilon, double *blgamma, double *gamma, int *steps_out)
{
    double *ws = (double *)malloc(sizeof(double) * box_ws_doubles(n));
    int s0, s1;
    s0 = dual_from_primal_box(P, q, l_min, l_max, l, n, epsilon, gamma, ws);
    s1 = solve_derivatives_box(P, l_min, l_max, l, gamma, grad_l, n, epsilon, blgamma, ws);
    if (steps_out) { steps_out[0] = s0; steps_out[1] = s1; }
    free(ws);
    return s1;
}

/* ------------------------------------------------------ batched entry points
 * The batch loops of qcqp.py:29-31, 45-51, 149-151, 167-180 with the gradient
 * assembly of qcqp.py:48-51 / 173-180 done in place of torch.bmm.  Layouts are
 * torch-contiguous: P (B,n,n), q/x/grad_x (B,n,1), l_n/mu (B,nc,1).
 * nthreads <= 1: serial (the reference's execution model); otherwise OpenMP. */

ORC_API void orc_qp_fwd_batch(const double *P, const double *q, long B, int n, double eps, double mu_prox,
                              int max_iter, double *x, int *iters, int nthreads)
{
    long b;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (b = 0; b < B; ++b) {
        int it = orc_solveQP(P + b * n * n, q + b * n, NULL, n, eps, mu_prox, max_iter, 1, x + b * n);
        if (iters) iters[b] = it;
    }
}

ORC_API void orc_qcqp_fwd_batch(const double *P, const double *q, const double *l_n, const double *mu, long B,
                                int n, double eps, double mu_prox, int max_iter, double *x, int *iters,
                                int nthreads)
{
    long b;
    int nc = n / 2;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (b = 0; b < B; ++b) {
        int it = orc_solveQCQP(P + b * n * n, q + b * n, l_n + b * nc, mu + b * nc, NULL, n, eps, mu_prox,
                               max_iter, 1, x + b * n);
        if (iters) iters[b] = it;
    }
}

/* grad_P = -dl x^T (qcqp.py:48-49), grad_q = -dl (qcqp.py:50-51) */
ORC_API void orc_qp_bwd_batch(const double *P, const double *q, const double *x, const double *grad_x, long B,
                              int n, double *grad_P, double *grad_q, int *ir_steps, int nthreads)
{
    long b;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (b = 0; b < B; ++b) {
        double *dl = (double *)malloc(sizeof(double) * n);
        int i, j;
        int st = orc_solveDerivativesQP(P + b * n * n, q + b * n, x + b * n, grad_x + b * n, n, 1e-10, dl);
        if (ir_steps) ir_steps[b] = st;
        if (grad_P)
            for (i = 0; i < n; ++i)
                for (j = 0; j < n; ++j) grad_P[b * n * n + i * n + j] = -(dl[i] * x[b * n + j]);
        if (grad_q)
            for (i = 0; i < n; ++i) grad_q[b * n + i] = -dl[i];
        free(dl);
    }
}

/* qcqp.py:173-180: grad_l_n = E2 dgamma, grad_mu = E1 dgamma (E diagonal) */
ORC_API void orc_qcqp_bwd_batch(const double *P, const double *q, const double *l_n, const double *mu,
                                const double *x, const double *grad_x, long B, int n, double *grad_P,
                                double *grad_q, double *grad_l_n, double *grad_mu, int *ir_steps, int nthreads)
{
    long b;
    int nc = n / 2;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (b = 0; b < B; ++b) {
        double *buf = (double *)malloc(sizeof(double) * (3 * nc + n + 4));
        double *e1 = buf, *e2 = e1 + nc, *blg = e2 + nc;
        const double *dg = blg, *dl = blg + nc;
        int i, j;
        int st = orc_solveDerivativesQCQP(P + b * n * n, q + b * n, l_n + b * nc, mu + b * nc, x + b * n,
                                          grad_x + b * n, n, 1e-10, e1, e2, blg, NULL);
        if (ir_steps) ir_steps[b] = st;
        if (grad_P)
            for (i = 0; i < n; ++i)
                for (j = 0; j < n; ++j) grad_P[b * n * n + i * n + j] = -(dl[i] * x[b * n + j]);
        if (grad_q)
            for (i = 0; i < n; ++i) grad_q[b * n + i] = -dl[i];
        if (grad_l_n)
            for (i = 0; i < nc; ++i) grad_l_n[b * nc + i] = e2[i] * dg[i];
        if (grad_mu)
            for (i = 0; i < nc; ++i) grad_mu[b * nc + i] = e1[i] * dg[i];
        free(buf);
    }
}

/* The batch loops of BoxQPFn2 / SignedBoxQPFn2.forward, qcqp.py:60-62 / :103-105.  v == NULL: box QP. */
ORC_API void orc_boxqp_fwd_batch(const double *P, const double *q, const double *l_min, const double *l_max,
                                 const double *v, long B, int n, double eps, double mu_prox, int max_iter,
                                 double *x, int *iters, int nthreads)
{
    long b;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (b = 0; b < B; ++b) {
        int it = v ? orc_solveSignedBoxQP(P + b * n * n, q + b * n, l_min + b * n, l_max + b * n, v + b * n, NULL, n,
                                          eps, mu_prox, max_iter, 1, x + b * n)
                   : orc_solveBoxQP(P + b * n * n, q + b * n, l_min + b * n, l_max + b * n, NULL, n, eps, mu_prox,
                                    max_iter, 1, x + b * n);
        if (iters) iters[b] = it;
    }
}

/* BoxQPFn2.backward as intended by qcqp.py:79-93 (the shipped Python does not run: SURVEY.md section 2 #7):
 * grad_P = -dl x^T, grad_q = -dl, and for the bounds the sensitivities of the complementarity rows
 * gamma_j c_j(l) with c_lower = l_min - l, c_upper = l - l_max:
 *   grad_l_min = -dgamma_lower o gamma_lower        (qcqp.py:91)
 *   grad_l_max = +dgamma_upper o gamma_upper        (qcqp.py:93 writes a minus sign; finite differences
 *                                                    -- tests/test_oracle.py -- say plus)
 * gamma_out (B,2n), ir_steps (B,2) may be NULL. */
ORC_API void orc_boxqp_bwd_batch(const double *P, const double *q, const double *l_min, const double *l_max,
                                 const double *x, const double *grad_x, long B, int n, double *grad_P,
                                 double *grad_q, double *grad_l_min, double *grad_l_max, double *gamma_out,
                                 int *ir_steps, int nthreads)
{
    long b;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (b = 0; b < B; ++b) {
        double *blg = (double *)malloc(sizeof(double) * 5 * n);
        double *gam = blg + 3 * n;
        const double *dg = blg, *dl = blg + 2 * n;
        int i, j, st[2];
        orc_solveDerivativesBoxQP(P + b * n * n, q + b * n, l_min + b * n, l_max + b * n, x + b * n, grad_x + b * n,
                                  n, 1e-10, blg, gam, st);
        if (ir_steps) { ir_steps[2 * b] = st[0]; ir_steps[2 * b + 1] = st[1]; }
        if (grad_P)
            for (i = 0; i < n; ++i)
                for (j = 0; j < n; ++j) grad_P[b * n * n + i * n + j] = -(dl[i] * x[b * n + j]);
        if (grad_q)
            for (i = 0; i < n; ++i) grad_q[b * n + i] = -dl[i];
        if (grad_l_min)
            for (i = 0; i < n; ++i) grad_l_min[b * n + i] = -(dg[i] * gam[i]);
        if (grad_l_max)
            for (i = 0; i < n; ++i) grad_l_max[b * n + i] = dg[n + i] * gam[n + i];
        if (gamma_out)
            for (i = 0; i < 2 * n; ++i) gamma_out[b * 2 * n + i] = gam[i];
        free(blg);
    }
}

ORC_API int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
