// ref_capi.cpp -- extern "C" shim over the REFERENCE's own Solver class, for oracle/_ref/libref.so.
//
// Test infrastructure (never product; see oracle/README.md).  DORMANT in this image: the reference includes
// <Eigen/Dense> (qcqplib/Solver.hpp:1, Solver.cpp:2) and no Eigen headers exist here, so oracle/ref_build.sh exits
// without building and tests/test_oracle_vs_ref.py skips.  On a machine with Eigen3 headers
// (EIGEN3_INCLUDE_DIR=/usr/include/eigen3 sh oracle/ref_build.sh) this file and the reference's qcqplib/Solver.cpp --
// compiled where it lies under /root/reference, nothing copied -- give the binary that pins the C restatement
// (oracle/diffqcqp_oracle.c) to the reference itself.
//
// Each function is the composition the reference's pybind11 module performs (reference pybindings.cpp:17-71), on plain
// pointers: row-major (n,n) matrices as numpy hands them over (py::EigenDRef copies them into column-major MatrixXd,
// as the by-value `MatrixXd P` parameters of Solver do), vectors of n doubles.
#include "qcqplib/Solver.hpp"   // -I/root/reference

#include <tuple>

namespace {
typedef Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> RowMat;
MatrixXd mat(const double* p, int n) { return Eigen::Map<const RowMat>(p, n, n); }
VectorXd vec(const double* p, int n) { return Eigen::Map<const VectorXd>(p, n); }
void put(const VectorXd& v, double* out) { for (int i = 0; i < (int)v.size(); ++i) out[i] = v(i); }
void put(const MatrixXd& m, double* out)
{
    for (int i = 0; i < (int)m.rows(); ++i)
        for (int j = 0; j < (int)m.cols(); ++j) out[i * m.cols() + j] = m(i, j);
}
}

extern "C" {

// pybindings.cpp:17-22
void ref_solveQP(const double* P, const double* q, const double* warm_start, int n, double epsilon, double mu_prox,
                 int max_iter, int adaptative_rho, double* x_out)
{
    Solver solver;
    put(solver.solveQP(mat(P, n), vec(q, n), vec(warm_start, n), epsilon, mu_prox, max_iter, adaptative_rho != 0), x_out);
}

// pybindings.cpp:24-30
void ref_solveDerivativesQP(const double* P, const double* q, const double* l, const double* grad_l, int n,
                            double epsilon, double* bl_out)
{
    Solver solver;
    const MatrixXd Pm = mat(P, n);
    const VectorXd qv = vec(q, n), lv = vec(l, n), gv = vec(grad_l, n);
    const VectorXd gamma = solver.dualFromPrimalQP(Pm, qv, lv, epsilon);
    put(solver.solveDerivativesQP(Pm, qv, lv, gamma, gv, epsilon), bl_out);
}

// pybindings.cpp:32-37
void ref_solveBoxQP(const double* P, const double* q, const double* l_min, const double* l_max,
                    const double* warm_start, int n, double epsilon, double mu_prox, int max_iter, int adaptative_rho,
                    double* x_out)
{
    Solver solver;
    put(solver.solveBoxQP(mat(P, n), vec(q, n), vec(l_min, n), vec(l_max, n), vec(warm_start, n), epsilon, mu_prox,
                          max_iter, adaptative_rho != 0), x_out);
}

// pybindings.cpp:39-45: blgamma (3n), gamma (2n)
void ref_solveDerivativesBoxQP(const double* P, const double* q, const double* l_min, const double* l_max,
                               const double* l, const double* grad_l, int n, double epsilon, double* blgamma_out,
                               double* gamma_out)
{
    Solver solver;
    const MatrixXd Pm = mat(P, n);
    const VectorXd qv = vec(q, n), lo = vec(l_min, n), hi = vec(l_max, n), lv = vec(l, n), gv = vec(grad_l, n);
    const VectorXd gamma = solver.dualFromPrimalBoxQP(Pm, qv, lo, hi, lv, epsilon);
    put(solver.solveDerivativesBoxQP(Pm, qv, lo, hi, lv, gamma, gv, epsilon), blgamma_out);
    put(gamma, gamma_out);
}

// pybindings.cpp:47-52
void ref_solveSignedBoxQP(const double* P, const double* q, const double* l_min, const double* l_max, const double* v,
                          const double* warm_start, int n, double epsilon, double mu_prox, int max_iter,
                          int adaptative_rho, double* x_out)
{
    Solver solver;
    put(solver.solveSignedBoxQP(mat(P, n), vec(q, n), vec(l_min, n), vec(l_max, n), vec(v, n), vec(warm_start, n),
                                epsilon, mu_prox, max_iter, adaptative_rho != 0), x_out);
}

// pybindings.cpp:54-60 (the radius l_n o mu is formed here, :57); l_n, mu: n/2 doubles
void ref_solveQCQP(const double* P, const double* q, const double* l_n, const double* mu, const double* warm_start,
                   int n, double epsilon, double mu_prox, int max_iter, int adaptative_rho, double* x_out)
{
    Solver solver;
    const VectorXd mul_n = vec(l_n, n / 2).cwiseProduct(vec(mu, n / 2));
    put(solver.solveQCQP(mat(P, n), vec(q, n), mul_n, vec(warm_start, n), epsilon, mu_prox, max_iter,
                         adaptative_rho != 0), x_out);
}

// pybindings.cpp:62-71: E1, E2 (n/2 x n/2, row-major), blgamma (n/2 + n)
void ref_solveDerivativesQCQP(const double* P, const double* q, const double* l_n, const double* mu, const double* l,
                              const double* grad_l, int n, double epsilon, double* E1_out, double* E2_out,
                              double* blgamma_out)
{
    Solver solver;
    const MatrixXd Pm = mat(P, n);
    const VectorXd qv = vec(q, n), ln = vec(l_n, n / 2), muv = vec(mu, n / 2), lv = vec(l, n), gv = vec(grad_l, n);
    const VectorXd mul_n = ln.cwiseProduct(muv);
    const VectorXd gamma = solver.dualFromPrimalQCQP(Pm, qv, mul_n, lv, epsilon);
    MatrixXd E1, E2;
    std::tie(E1, E2) = solver.getE12QCQP(ln, muv, gamma);
    put(E1, E1_out);
    put(E2, E2_out);
    put(solver.solveDerivativesQCQP(Pm, qv, mul_n, lv, gamma, gv, epsilon), blgamma_out);
}

#define DQQ_STR2(x) #x
#define DQQ_STR(x) DQQ_STR2(x)
const char* ref_version(void)
{
    return "quentinll/diffqcqp qcqplib/Solver.cpp, Eigen " DQQ_STR(EIGEN_WORLD_VERSION) "." DQQ_STR(EIGEN_MAJOR_VERSION) "." DQQ_STR(EIGEN_MINOR_VERSION);
}

} // extern "C"
