// pybind_module.cpp -- the pybind11 face of the C ABI (include/diffqcqp_hip.h), module `diffqcqp_amd._dqq`.
//
// The reference reaches its solver through a pybind11 module (`diffqcqp`, pybindings.cpp:74-83: one call per
// PROBLEM, numpy in / numpy out).  This module is the batched counterpart: one call per BATCH, every function of
// the C ABI under its own name with the same argument order.  Pointers are passed as Python ints (what
// `torch.Tensor.data_ptr()` and `torch.cuda.current_stream().cuda_stream` return) or None; nothing is copied, no torch
// type crosses the boundary.  It adds no logic of its own -- `diffqcqp_amd/_capi.py` binds the very same symbols with
// ctypes when this module has not been built; the pybind11 call costs about a microsecond where the 16-argument
// ctypes call costs several (it matters at B = 1: the reference's published figure is a single problem).
// Host-only translation unit (g++): the HIP code is behind the C ABI.
#include <pybind11/pybind11.h>

#include <cstdint>

#include "diffqcqp_hip.h"

namespace py = pybind11;

namespace {

template <typename T>
T* ptr(const py::object& o)
{
    return o.is_none() ? nullptr : reinterpret_cast<T*>(o.cast<std::uintptr_t>());
}
using O = const py::object&;

} // namespace

PYBIND11_MODULE(_dqq, m)
{
    m.doc() = "pybind11 binding of libdiffqcqp_hip.so (include/diffqcqp_hip.h): batched ADMM QP / QCQP on MI355X";
    m.def("dqq_workspace_bytes", [](std::int64_t B) { return dqq_workspace_bytes(B); });
    m.def("dqq_scratch_bytes", [](int kind, int pass, int N, std::int64_t B, int p_layout) {
        return dqq_scratch_bytes(kind, pass, N, B, p_layout);
    });
    m.def("dqq_max_n", [](int kind, int p_layout) { return dqq_max_n(kind, p_layout); });
    m.def("dqq_workspace_reset", [](O ws, std::size_t ws_bytes, O stream) {
        return dqq_workspace_reset(ptr<void>(ws), ws_bytes, ptr<void>(stream));
    });
    m.def("dqq_workspace_status", [](O ws, std::size_t ws_bytes, O stream) {
        int dirty = 0;
        const int rc = dqq_workspace_status(ptr<const void>(ws), ws_bytes, ptr<void>(stream), &dirty);
        return py::make_tuple(rc, dirty);
    });
    m.def("dqq_version", []() { return py::bytes(dqq_version()); });
    m.def("dqq_set_option", [](const py::bytes& name, int value) { return dqq_set_option(std::string(name).c_str(), value); });
    m.def("dqq_hint_flags", [](int kind, int pass, int N, std::int64_t B, unsigned long long last_report) {
        return dqq_hint_flags(kind, pass, N, B, last_report);
    });
    m.def("dqq_device_pointer", [](O pinned_host) {
        void* dev = nullptr;
        const int rc = dqq_device_pointer(ptr<void>(pinned_host), &dev);
        return py::make_tuple(rc, reinterpret_cast<std::uintptr_t>(dev));
    });
    m.def("dqq_get_option", [](const py::bytes& name) {
        int v = 0;
        const int rc = dqq_get_option(std::string(name).c_str(), &v);
        return py::make_tuple(rc, v);
    });
    m.def("dqq_qp_fwd_f64", [](O P, O q, O x, std::int64_t B, int N, double eps, double mu_prox, int max_iter,
                               int adaptive_rho, int p_layout, O iters, O pdiag_out, O flags_out, O ws,
                               std::size_t ws_bytes, O stream) {
        return dqq_qp_fwd_f64(ptr<const double>(P), ptr<const double>(q), ptr<double>(x), B, N, eps, mu_prox, max_iter,
                              adaptive_rho, p_layout, ptr<int>(iters), ptr<double>(pdiag_out),
                              ptr<unsigned char>(flags_out), ptr<void>(ws), ws_bytes, ptr<void>(stream));
    });
    m.def("dqq_qp_bwd_f64", [](O P, O q, O x, O grad_x, O grad_P, O grad_q, std::int64_t B, int N, double epsilon,
                               int p_layout, O ir_steps, O pdiag, O flags, O report, O ws, std::size_t ws_bytes, O stream) {
        return dqq_qp_bwd_f64(ptr<const double>(P), ptr<const double>(q), ptr<const double>(x), ptr<const double>(grad_x),
                              ptr<double>(grad_P), ptr<double>(grad_q), B, N, epsilon, p_layout, ptr<int>(ir_steps),
                              ptr<const double>(pdiag), ptr<const unsigned char>(flags), ptr<unsigned long long>(report),
                              ptr<void>(ws), ws_bytes, ptr<void>(stream));
    });
    m.def("dqq_qcqp_fwd_f64", [](O P, O q, O l_n, O mu, O x, std::int64_t B, int N, double eps, double mu_prox,
                                 int max_iter, int adaptive_rho, int p_layout, O iters, O pdiag_out, O flags_out, O ws,
                                 std::size_t ws_bytes, O stream) {
        return dqq_qcqp_fwd_f64(ptr<const double>(P), ptr<const double>(q), ptr<const double>(l_n), ptr<const double>(mu),
                                ptr<double>(x), B, N, eps, mu_prox, max_iter, adaptive_rho, p_layout, ptr<int>(iters),
                                ptr<double>(pdiag_out), ptr<unsigned char>(flags_out), ptr<void>(ws), ws_bytes,
                                ptr<void>(stream));
    });
    m.def("dqq_qcqp_bwd_f64", [](O P, O q, O l_n, O mu, O x, O grad_x, O grad_P, O grad_q, O grad_l_n, O grad_mu, O gamma,
                                 O dgamma, std::int64_t B, int N, double epsilon, int p_layout, O ir_steps, O pdiag,
                                 O flags, O report, O ws, std::size_t ws_bytes, O stream) {
        return dqq_qcqp_bwd_f64(ptr<const double>(P), ptr<const double>(q), ptr<const double>(l_n), ptr<const double>(mu),
                                ptr<const double>(x), ptr<const double>(grad_x), ptr<double>(grad_P), ptr<double>(grad_q),
                                ptr<double>(grad_l_n), ptr<double>(grad_mu), ptr<double>(gamma), ptr<double>(dgamma), B, N,
                                epsilon, p_layout, ptr<int>(ir_steps), ptr<const double>(pdiag),
                                ptr<const unsigned char>(flags), ptr<unsigned long long>(report), ptr<void>(ws), ws_bytes,
                                ptr<void>(stream));
    });
    m.def("dqq_boxqp_fwd_f64", [](O P, O q, O l_min, O l_max, O x, std::int64_t B, int N, double eps, double mu_prox,
                                  int max_iter, int adaptive_rho, int p_layout, O iters, O pdiag_out, O flags_out, O ws,
                                  std::size_t ws_bytes, O stream) {
        return dqq_boxqp_fwd_f64(ptr<const double>(P), ptr<const double>(q), ptr<const double>(l_min),
                                 ptr<const double>(l_max), ptr<double>(x), B, N, eps, mu_prox, max_iter, adaptive_rho,
                                 p_layout, ptr<int>(iters), ptr<double>(pdiag_out), ptr<unsigned char>(flags_out),
                                 ptr<void>(ws), ws_bytes, ptr<void>(stream));
    });
    m.def("dqq_signedboxqp_fwd_f64", [](O P, O q, O l_min, O l_max, O v, O x, std::int64_t B, int N, double eps,
                                        double mu_prox, int max_iter, int adaptive_rho, int p_layout, O iters, O pdiag_out,
                                        O flags_out, O ws, std::size_t ws_bytes, O stream) {
        return dqq_signedboxqp_fwd_f64(ptr<const double>(P), ptr<const double>(q), ptr<const double>(l_min),
                                       ptr<const double>(l_max), ptr<const double>(v), ptr<double>(x), B, N, eps, mu_prox,
                                       max_iter, adaptive_rho, p_layout, ptr<int>(iters), ptr<double>(pdiag_out),
                                       ptr<unsigned char>(flags_out), ptr<void>(ws), ws_bytes, ptr<void>(stream));
    });
    m.def("dqq_boxqp_bwd_f64", [](O P, O q, O l_min, O l_max, O x, O grad_x, O grad_P, O grad_q, O grad_l_min,
                                  O grad_l_max, O gamma, O dgamma, std::int64_t B, int N, double epsilon, int p_layout,
                                  O ir_steps, O pdiag, O flags, O ws, std::size_t ws_bytes, O stream) {
        return dqq_boxqp_bwd_f64(ptr<const double>(P), ptr<const double>(q), ptr<const double>(l_min),
                                 ptr<const double>(l_max), ptr<const double>(x), ptr<const double>(grad_x),
                                 ptr<double>(grad_P), ptr<double>(grad_q), ptr<double>(grad_l_min), ptr<double>(grad_l_max),
                                 ptr<double>(gamma), ptr<double>(dgamma), B, N, epsilon, p_layout, ptr<int>(ir_steps),
                                 ptr<const double>(pdiag), ptr<const unsigned char>(flags), ptr<void>(ws), ws_bytes,
                                 ptr<void>(stream));
    });
}
