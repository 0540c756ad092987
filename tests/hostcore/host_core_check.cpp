// host_core_check.cpp -- TEST ARTEFACT.  Instantiates the host/device math
// templates of diffqcqp_amd/csrc (admm_core.h, kkt_core.h) for the CPU with one
// "lane" per problem, so that the per-problem arithmetic the HIP kernels run can
// be compared with the oracle on a machine without a GPU.  Nothing in the
// product links or loads this file.
#include "../../diffqcqp_amd/csrc/admm_core.h"
#include "../../diffqcqp_amd/csrc/kkt_core.h"

using namespace dqq;

template <int KIND, int E>
static int fwd_one(const double* p, const double* q, const double* rad, double eps, double mu, int max_iter,
                   int adaptive, double* x)
{
    double pp[E], qq[E], xx[E], rr[E / 2];
    for (int e = 0; e < E; ++e) { pp[e] = p[e]; qq[e] = q[e]; }
    for (int c = 0; c < E / 2; ++c) rr[c] = rad ? rad[c] : 0.0;
    int it = admm_fwd_diag<KIND, E, HostGroup>(pp, qq, rr, E, eps, mu, max_iter, adaptive, true, xx);
    for (int e = 0; e < E; ++e) x[e] = xx[e];
    return it;
}

template <int KIND, int E>
static int box_one(const double* p, const double* q, const double* lo, const double* hi, const double* v, double eps,
                   double mu, int max_iter, int adaptive, double* x)
{
    double pp[E], qq[E], xx[E], rr[E / 2], sg[E];
    for (int e = 0; e < E; ++e) { pp[e] = p[e]; qq[e] = q[e]; sg[e] = v ? (double)((v[e] > 0) - (v[e] < 0)) : 0.0; }
    for (int c = 0; c < E / 2; ++c) rr[c] = 0.0;
    int it = admm_fwd_diag<KIND, E, HostGroup>(pp, qq, rr, E, eps, mu, max_iter, adaptive, true, xx, lo, hi, sg);
    for (int e = 0; e < E; ++e) x[e] = xx[e];
    return it;
}

extern "C" {

// diagonal-P box QP (v == NULL) / signed box QP forward, one problem
__attribute__((visibility("default"))) int hostcore_box_fwd(int n, const double* p, const double* q, const double* lo,
                                                            const double* hi, const double* v, double eps, double mu,
                                                            int max_iter, int adaptive, double* x)
{
#define CASE(NN) if (n == NN) return v ? box_one<3, NN>(p, q, lo, hi, v, eps, mu, max_iter, adaptive, x) \
                                       : box_one<2, NN>(p, q, lo, hi, v, eps, mu, max_iter, adaptive, x);
    CASE(2) CASE(4) CASE(8) CASE(16)
#undef CASE
    return -1;
}

// diagonal-P box QP backward, one problem: the per-coordinate blocks + the two refinement loops exactly as
// bwd_diag.hip drives them.  out: dl (n), gamma (2n), dgamma (2n); steps[2].
__attribute__((visibility("default"))) void hostcore_box_bwd(int n, const double* p, const double* q, const double* lo,
                                                             const double* hi, const double* x, const double* g,
                                                             double* dl, double* gamma, double* dgamma, int* steps)
{
    BoxCoord* c = new BoxCoord[n];
    for (int i = 0; i < n; ++i) c[i].setup_dual(p[i], q[i], x[i], lo[i], hi[i], kActiveEps);
    IrControl ctl;
    ctl.init();
    for (int it = 0; it < kIrMaxIter; ++it) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) { double d[3]; c[i].step_dual(d); s += d[0]; s += d[1]; }
        steps[0] = it + 1;
        if (ctl.update(sqrt(s))) break;
    }
    for (int i = 0; i < n; ++i) c[i].setup_derivative(p[i], g[i]);
    ctl.init();
    double* tail = new double[n];
    for (int it = 0; it < kIrMaxIter; ++it) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) { double d[3]; c[i].step_derivative(d); s += d[0]; s += d[1]; tail[i] = d[2]; }
        for (int i = 0; i < n; ++i) s += tail[i];
        steps[1] = it + 1;
        if (ctl.update(sqrt(s))) break;
    }
    for (int i = 0; i < n; ++i) {
        dl[i] = c[i].dl();
        gamma[i] = c[i].gamma_lo; gamma[n + i] = c[i].gamma_hi;
        dgamma[i] = c[i].dgamma_lo(); dgamma[n + i] = c[i].dgamma_hi();
    }
    delete[] tail;
    delete[] c;
}

// diagonal-P forward, one problem; p = diagonal (n), rad = l_n*mu (n/2) for kind 1
__attribute__((visibility("default"))) int hostcore_fwd(int kind, int n, const double* p, const double* q,
                                                        const double* rad, double eps, double mu, int max_iter,
                                                        int adaptive, double* x)
{
#define CASE(K, NN) if (kind == K && n == NN) return fwd_one<K, NN>(p, q, rad, eps, mu, max_iter, adaptive, x);
    CASE(0, 2) CASE(0, 4) CASE(0, 8) CASE(0, 16) CASE(1, 2) CASE(1, 4) CASE(1, 8) CASE(1, 16)
#undef CASE
    return -1;
}

// diagonal-P QP backward, one problem: the per-lane blocks + the refinement loop
// exactly as bwd_diag.hip drives them (residual summed in index order).
__attribute__((visibility("default"))) int hostcore_qp_bwd(int n, const double* p, const double* q, const double* x,
                                                           const double* g, double* dl)
{
    QpCoord* c = new QpCoord[n];
    for (int i = 0; i < n; ++i) c[i].setup(p[i], q[i], x[i], g[i]);
    IrControl ctl;
    ctl.init();
    int steps = 0;
    for (int it = 0; it < kIrMaxIter; ++it) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += c[i].step();
        steps = it + 1;
        if (ctl.update(sqrt(s))) break;
    }
    for (int i = 0; i < n; ++i) dl[i] = c[i].dl();
    delete[] c;
    return steps;
}

__attribute__((visibility("default"))) int hostcore_qcqp_bwd(int n, const double* p, const double* q,
                                                             const double* l_n, const double* mu, const double* x,
                                                             const double* g, double* dl, double* grad_l_n,
                                                             double* grad_mu)
{
    const int nc = n / 2;
    QcqpContact* c = new QcqpContact[nc];
    double* rs = new double[n + nc];
    for (int i = 0; i < nc; ++i)
        c[i].setup(p[2 * i], p[2 * i + 1], q[2 * i], q[2 * i + 1], x[2 * i], x[2 * i + 1], g[2 * i], g[2 * i + 1], l_n[i],
                   mu[i]);
    IrControl ctl;
    ctl.init();
    int steps = 0;
    for (int it = 0; it < kIrMaxIter; ++it) {
        for (int i = 0; i < nc; ++i) {
            double d[3];
            c[i].step(d);
            rs[i] = d[0];
            rs[nc + 2 * i] = d[1];
            rs[nc + 2 * i + 1] = d[2];
        }
        double s = 0.0;
        for (int i = 0; i < n + nc; ++i) s += rs[i];
        steps = it + 1;
        if (ctl.update(sqrt(s))) break;
    }
    for (int i = 0; i < nc; ++i) {
        dl[2 * i] = c[i].dla();
        dl[2 * i + 1] = c[i].dlb();
        grad_l_n[i] = QcqpContact::e2(c[i].gamma, l_n[i], mu[i]) * c[i].dgamma();
        grad_mu[i] = QcqpContact::e1(c[i].gamma, l_n[i], mu[i]) * c[i].dgamma();
    }
    delete[] c;
    delete[] rs;
    return steps;
}
}
