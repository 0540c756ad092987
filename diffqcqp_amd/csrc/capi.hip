// capi.hip -- the extern "C" boundary declared in include/diffqcqp_hip.h:
// argument checks, path selection (diagonal fast path / general dense kernel /
// both, chained through the fallback work-list), launches on the caller's stream.
#include <atomic>
#include <cstring>

#include "launch.h"

namespace dqq {
#if defined(DQQ_TUNING)
// developer build: the knobs of tuning.h as process-wide atomics (defined here, declared there)
#define DQQ_KNOB_DEF(name, dflt) std::atomic<int> g_##name{dflt};
DQQ_KNOB_DEF(fwd_lpp, 0) DQQ_KNOB_DEF(wpb, 0) DQQ_KNOB_DEF(fuse_fallback, -1)
DQQ_KNOB_DEF(fwd_respread, 16) DQQ_KNOB_DEF(fwd_respread2, 8) DQQ_KNOB_DEF(fwd_respread2_from, 48) DQQ_KNOB_DEF(lane_dense, 1) DQQ_KNOB_DEF(lane_defer, 0)
DQQ_KNOB_DEF(dense_teams, 1) DQQ_KNOB_DEF(small_fwd, 1) DQQ_KNOB_DEF(small_bwd, 1) DQQ_KNOB_DEF(lane_bwd, 1)
DQQ_KNOB_DEF(fwd_feedback, 1) DQQ_KNOB_DEF(bwd_skip_classify, 1)
#undef DQQ_KNOB_DEF
#endif
}

namespace {

struct Option {
    const char* name;
    std::atomic<int>* slot;
};
// What dqq_set_option / dqq_get_option know.  Shipped build: the three route counters (diagnostics; "set" resets them) and
// nothing else -- no name here changes what a call does.  Developer build (-DDQQ_TUNING): also the knobs of tuning.h.
Option g_options[] = {{"lane_list_drains", &dqq::g_lane_list_drains},
                      {"bwd_whole_batches", &dqq::g_bwd_whole_batches},
                      {"fwd_feedback_routes", &dqq::g_fwd_feedback_routes},
#if defined(DQQ_TUNING)
#define DQQ_KNOB_OPT(name) {#name, &dqq::g_##name},
                      DQQ_KNOB_OPT(fwd_lpp) DQQ_KNOB_OPT(wpb) DQQ_KNOB_OPT(fuse_fallback)
                      DQQ_KNOB_OPT(fwd_respread) DQQ_KNOB_OPT(fwd_respread2) DQQ_KNOB_OPT(fwd_respread2_from) DQQ_KNOB_OPT(lane_dense) DQQ_KNOB_OPT(lane_defer)
                      DQQ_KNOB_OPT(dense_teams) DQQ_KNOB_OPT(small_fwd) DQQ_KNOB_OPT(small_bwd) DQQ_KNOB_OPT(lane_bwd)
                      DQQ_KNOB_OPT(fwd_feedback) DQQ_KNOB_OPT(bwd_skip_classify)
#undef DQQ_KNOB_OPT
#endif
};

// p_layout as passed = layout | flags
constexpr int kAllFlags = DQQ_F_REFERENCE_ORDER | DQQ_F_EXPECT_DENSE | DQQ_F_EXPECT_LONG_LIST;
int layout_of(int p_layout) { return p_layout & 0xff; }
bool ref_order_of(int p_layout) { return (p_layout & DQQ_F_REFERENCE_ORDER) != 0; }
int hints_of(int p_layout) { return p_layout & (DQQ_F_EXPECT_DENSE | DQQ_F_EXPECT_LONG_LIST); }

int check_common(int64_t B, int N, int p_layout, bool qcqp)
{
    if (B < 0 || N < 1 || B > 0x7fffffffLL) return DQQ_E_BAD_SIZE;
    if (qcqp && (N % 2) != 0) return DQQ_E_BAD_SIZE;
    const int layout = layout_of(p_layout);
    if ((p_layout & ~(0xff | kAllFlags)) != 0) return DQQ_E_BAD_LAYOUT;   // unknown flag bits
    if (layout != DQQ_P_AUTO && layout != DQQ_P_DENSE && layout != DQQ_P_DIAG) return DQQ_E_BAD_LAYOUT;
    return 0;
}

int check_ws(const void* ws, size_t bytes, int64_t B, size_t scratch_bytes = 0)
{
    if (ws == nullptr || bytes < dqq_workspace_bytes(B) + scratch_bytes) return DQQ_E_WORKSPACE;
    return 0;
}

// the scratch slice behind the work-list (dqq_scratch_bytes)
double* scratch_of(void* ws, int64_t B)
{
    return reinterpret_cast<double*>(static_cast<char*>(ws) + dqq_workspace_bytes(B));
}

// A launch of the work-list chain failed: the entries the fast kernel queued will never be drained.  Re-zero the header
// so that the stale count cannot leak into the next call on this workspace (best effort; the error is what is returned).
void reset_worklist(void* ws, hipStream_t s)
{
    if (ws != nullptr) (void)hipMemsetAsync(ws, 0, sizeof(int) * dqq::kWsEntries, s);
}

} // namespace

extern "C" {

size_t dqq_workspace_bytes(int64_t B)
{
    if (B < 0) B = 0;
    // header + the entry area: B slots of the plain list, or the 32 segments of the N >= 32 list (launch.h)
    size_t n = (size_t)dqq::kWsEntries + (size_t)dqq::kWsEntryInts((long)B);
    n = (n + 63) & ~(size_t)63;
    return n * sizeof(int);
}

size_t dqq_scratch_bytes(int kind, int pass, int N, int64_t B, int p_layout)
{
    if (B <= 0 || N < 1 || kind < 0 || kind > 3 || (pass != 0 && pass != 1)) return 0;
    if (pass == 0) return dqq::fwd_needs_any(kind, N) ? dqq::any_scratch_bytes(kind, false, N, (long)B) : 0;
    if (kind == dqq::kKindSignedBox) return 0; // no backward
    return dqq::bwd_needs_any(kind, N, ref_order_of(p_layout)) ? dqq::any_scratch_bytes(kind, true, N, (long)B) : 0;
}

int dqq_workspace_reset(void* workspace, size_t workspace_bytes, void* stream)
{
    if (workspace == nullptr) return DQQ_E_NULLPTR;
    const size_t head = sizeof(int) * (size_t)dqq::kWsEntries;
    if (workspace_bytes < head) return DQQ_E_WORKSPACE;
    return (int)hipMemsetAsync(workspace, 0, head, static_cast<hipStream_t>(stream));
}

int dqq_workspace_status(const void* workspace, size_t workspace_bytes, void* stream, int* dirty)
{
    if (workspace == nullptr || dirty == nullptr) return DQQ_E_NULLPTR;
    if (workspace_bytes < sizeof(int) * (size_t)dqq::kWsEntries) return DQQ_E_WORKSPACE;
    int word = 0;
    hipError_t e = hipMemcpyAsync(&word, static_cast<const int*>(workspace) + dqq::kWsDirty, sizeof(int),
                                  hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return (int)e;
    *dirty = word != 0 ? 1 : 0;
    return 0;
}

int dqq_max_n(int kind, int p_layout) { return dqq::public_max_n(kind, ref_order_of(p_layout)); }

const char* dqq_version(void) { return "diffqcqp_hip 0.2.0 gfx950"; }

int dqq_set_option(const char* name, int value)
{
    if (name == nullptr) return DQQ_E_NULLPTR;
    for (auto& o : g_options)
        if (std::strcmp(o.name, name) == 0) { o.slot->store(value); return 0; }
    return DQQ_E_BAD_OPTION;
}

int dqq_hint_flags(int kind, int pass, int N, int64_t B, unsigned long long last_report)
{
    // pure: the flags a caller may OR into p_layout for a call of (kind, pass, N, B) when the last backward of that kind and N
    // left `last_report` in the caller's report word (launch.h).  0 whenever the word says nothing about such a batch.
    if (!dqq::hint_applies(kind, N) || B <= 0 || (pass != 0 && pass != 1)) return 0;
    int flags = 0;
    if (pass == 0) {
        // forward, N = 8: half of the batch or more sat in 16-problem blocks with a non-diagonal problem
        if (N == 8 && 2 * dqq::report_count_in_blocks(last_report, (long)B) >= B) flags |= DQQ_F_EXPECT_DENSE;
        return flags;
    }
    if (!dqq::bwd_lane_dense_supported(kind, N, (long)B)) return 0;
    int streak = 0;
    const long c = dqq::report_count(last_report, (long)B, &streak);
    if (4 * c >= 3 * B && streak >= 1) flags |= DQQ_F_EXPECT_DENSE;              // all non-diagonal, twice running
    if (dqq::bwd_lane_dense_supported(kind, N, c)) flags |= DQQ_F_EXPECT_LONG_LIST; // a list that fills the chip
    return flags;
}

int dqq_device_pointer(void* pinned_host, void** device)
{
    if (pinned_host == nullptr || device == nullptr) return DQQ_E_NULLPTR;
    if ((reinterpret_cast<uintptr_t>(pinned_host) & 7) != 0) return DQQ_E_BAD_SIZE;
    return (int)hipHostGetDevicePointer(device, pinned_host, 0);   // (pinned / registered host memory only)
}

int dqq_get_option(const char* name, int* value)
{
    if (name == nullptr || value == nullptr) return DQQ_E_NULLPTR;
    for (auto& o : g_options)
        if (std::strcmp(o.name, name) == 0) { *value = o.slot->load(); return 0; }
    return DQQ_E_BAD_OPTION;
}

// Routing is a function of (kind, N, B, p_layout) only -- p_layout including the caller's hint flags --: two calls with the
// same arguments launch the same kernels, whatever ran before them, on whatever thread or stream.  The library keeps no
// state.  (The developer build, -DDQQ_TUNING, adds the knobs of tuning.h.)
static int fwd_dispatch(int kind, dqq::FwdArgs& a, void* workspace, size_t workspace_bytes, hipStream_t s)
{
    if (a.B == 0) return 0;
    const bool fast_ok = dqq::fwd_diag_supported(a.N);
    const bool dense_ok = dqq::fwd_dense_supported(kind, a.N);
    if (a.layout == DQQ_P_DIAG) {
        if (!fast_ok) return DQQ_E_UNSUPPORTED_N;
        return (int)dqq::launch_fwd_diag(kind, a, dqq::knob_fwd_lpp(), dqq::knob_wpb(), dqq::knob_fuse_fallback(), s, nullptr);
    }
    const size_t scratch = dqq_scratch_bytes(kind, 0, a.N, a.B, a.layout); // > 0: the global-memory kernels take the call
    if (a.layout == DQQ_P_DENSE || !fast_ok) {
        if (a.layout == DQQ_P_DENSE && fast_ok && dqq::knob_lane_dense() != 0 && dqq::knob_fuse_fallback() != 0 &&
            dqq::fwd_diag_takes_dense(kind, a.N, a.B))
            return (int)dqq::launch_fwd_diag(kind, a, dqq::knob_fwd_lpp(), dqq::knob_wpb(), 1, s, nullptr);
        if (!dense_ok) return DQQ_E_UNSUPPORTED_N;
        if (scratch > 0) {
            if (int rc = check_ws(workspace, workspace_bytes, a.B, scratch)) return rc;
            a.scratch = scratch_of(workspace, a.B);
        }
        return (int)dqq::launch_fwd_dense(kind, a, false, s);
    }
    // DQQ_P_AUTO: fast path over every tile; non-diagonal tiles are solved inside it (small N) or
    // queued for the dense kernel launched right behind it
    if (int rc = check_ws(workspace, workspace_bytes, a.B, scratch)) return rc;
    a.ws = static_cast<int*>(workspace);
    if (scratch > 0) a.scratch = scratch_of(workspace, a.B);
    const bool fused = dqq::fwd_diag_will_fuse(a.N, a.B, a.layout, dqq::knob_fuse_fallback());
    if (!fused && !dense_ok) return DQQ_E_UNSUPPORTED_N; // a queued tile would never be solved: refuse up front
    bool needs_fallback = true;
    hipError_t e = dqq::launch_fwd_diag(kind, a, dqq::knob_fwd_lpp(), dqq::knob_wpb(), dqq::knob_fuse_fallback(), s, &needs_fallback);
    if (e != hipSuccess) return (int)e;
    if (needs_fallback) {
        e = dqq::launch_fwd_dense(kind, a, true, s);
        if (e != hipSuccess) reset_worklist(workspace, s);
    }
    return (int)e;
}

static int bwd_dispatch(int kind, dqq::BwdArgs& a, void* workspace, size_t workspace_bytes, hipStream_t s)
{
    if (a.B == 0) return 0;
    const bool fast_ok = dqq::bwd_diag_supported(a.N);
    const bool dense_ok = dqq::bwd_dense_supported(kind, a.N);
    if (a.layout == DQQ_P_DIAG) {
        if (!fast_ok) return DQQ_E_UNSUPPORTED_N;
        return (int)dqq::launch_bwd_diag(kind, a, dqq::knob_wpb(), dqq::knob_fuse_fallback(), s, nullptr);
    }
    const size_t scratch = dqq_scratch_bytes(kind, 1, a.N, a.B, a.layout | (a.ref_order ? DQQ_F_REFERENCE_ORDER : 0));
    if (a.layout == DQQ_P_DENSE || !fast_ok) {
        if (!dense_ok) return DQQ_E_UNSUPPORTED_N;
        if (scratch > 0) {
            if (int rc = check_ws(workspace, workspace_bytes, a.B, scratch)) return rc;
            a.scratch = scratch_of(workspace, a.B);
        }
        return (int)dqq::launch_bwd_dense(kind, a, false, s);
    }
    if (int rc = check_ws(workspace, workspace_bytes, a.B, scratch)) return rc;
    a.ws = static_cast<int*>(workspace);
    if (scratch > 0) a.scratch = scratch_of(workspace, a.B);
    const bool fused = dqq::bwd_diag_will_fuse(kind, a.N, a.B, a.layout, dqq::knob_fuse_fallback());
    if (!fused && !dense_ok) return DQQ_E_UNSUPPORTED_N; // (box QP, N > 32): nothing could drain the work-list
    // the caller expects all of it non-diagonal (DQQ_F_EXPECT_DENSE): one launch of the lane-per-problem kernel over the whole
    // batch, which also recounts for the caller's next hint (bwd_lane_dense.hip REPORT; a diagonal problem gets the same bits there)
    if (!fused && dqq::bwd_lane_takes_auto_batch(kind, a.N, a.B, a.hints)) {
        dqq::g_bwd_whole_batches.fetch_add(1, std::memory_order_relaxed);
        hipError_t e2 = dqq::launch_bwd_lane_dense(kind, a, 2, s);
        if (e2 != hipSuccess) reset_worklist(workspace, s);
        return (int)e2;
    }
    bool needs_fallback = true;
    hipError_t e = dqq::launch_bwd_diag(kind, a, dqq::knob_wpb(), dqq::knob_fuse_fallback(), s, &needs_fallback);
    if (e != hipSuccess) return (int)e;
    if (needs_fallback) {
        e = dqq::launch_bwd_dense(kind, a, true, s);
        if (e != hipSuccess) reset_worklist(workspace, s);
    }
    return (int)e;
}

int dqq_qp_fwd_f64(const double* P, const double* q, double* x, int64_t B, int N, double eps, double mu_prox,
                   int max_iter, int adaptive_rho, int p_layout, int* iters, double* pdiag_out,
                   unsigned char* diag_flags_out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (int rc = check_common(B, N, p_layout, false)) return rc;
    if (B > 0 && (P == nullptr || q == nullptr || x == nullptr)) return DQQ_E_NULLPTR;
    const int layout = layout_of(p_layout);
    const bool keep = layout == DQQ_P_AUTO && dqq::fwd_diag_supported(N);
    dqq::FwdArgs a{P,        q,     nullptr, nullptr, nullptr, x, (long)B, N, eps, mu_prox, max_iter, adaptive_rho ? 1 : 0,
                   layout,   iters, nullptr, keep ? pdiag_out : nullptr, keep ? diag_flags_out : nullptr};
    a.ref_order = ref_order_of(p_layout);
    a.hints = hints_of(p_layout);
    if (!keep && diag_flags_out != nullptr && B > 0) { // nothing will be verified: flag every problem 0
        hipError_t e = hipMemsetAsync(diag_flags_out, 0, (size_t)B, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return (int)e;
    }
    return fwd_dispatch(0, a, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int dqq_qcqp_fwd_f64(const double* P, const double* q, const double* l_n, const double* mu, double* x, int64_t B,
                     int N, double eps, double mu_prox, int max_iter, int adaptive_rho, int p_layout, int* iters,
                     double* pdiag_out, unsigned char* diag_flags_out, void* workspace, size_t workspace_bytes,
                     void* stream)
{
    if (int rc = check_common(B, N, p_layout, true)) return rc;
    if (B > 0 && (P == nullptr || q == nullptr || l_n == nullptr || mu == nullptr || x == nullptr))
        return DQQ_E_NULLPTR;
    const int layout = layout_of(p_layout);
    const bool keep = layout == DQQ_P_AUTO && dqq::fwd_diag_supported(N);
    dqq::FwdArgs a{P,     q,       l_n, mu, nullptr, x, (long)B, N, eps, mu_prox, max_iter, adaptive_rho ? 1 : 0, layout,
                   iters, nullptr, keep ? pdiag_out : nullptr, keep ? diag_flags_out : nullptr};
    a.ref_order = ref_order_of(p_layout);
    a.hints = hints_of(p_layout);
    if (!keep && diag_flags_out != nullptr && B > 0) {
        hipError_t e = hipMemsetAsync(diag_flags_out, 0, (size_t)B, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return (int)e;
    }
    return fwd_dispatch(1, a, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

// Box QP (kind 2) and signed box QP (kind 3) share the forward plumbing: v == nullptr selects the box QP.
static int box_fwd(const double* P, const double* q, const double* l_min, const double* l_max, const double* v,
                   bool is_signed, double* x, int64_t B, int N, double eps, double mu_prox, int max_iter,
                   int adaptive_rho, int p_layout, int* iters, double* pdiag_out, unsigned char* diag_flags_out,
                   void* workspace, size_t workspace_bytes, void* stream)
{
    if (int rc = check_common(B, N, p_layout, false)) return rc;
    if (B > 0 && (P == nullptr || q == nullptr || l_min == nullptr || l_max == nullptr || x == nullptr ||
                  (is_signed && v == nullptr)))
        return DQQ_E_NULLPTR;
    const int layout = layout_of(p_layout);
    const bool keep = layout == DQQ_P_AUTO && dqq::fwd_diag_supported(N);
    dqq::FwdArgs a{P,        q,     l_min,   l_max, v, x, (long)B, N, eps, mu_prox, max_iter, adaptive_rho ? 1 : 0,
                   layout,   iters, nullptr, keep ? pdiag_out : nullptr, keep ? diag_flags_out : nullptr};
    a.ref_order = ref_order_of(p_layout);
    a.hints = hints_of(p_layout);
    if (!keep && diag_flags_out != nullptr && B > 0) {
        hipError_t e = hipMemsetAsync(diag_flags_out, 0, (size_t)B, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return (int)e;
    }
    return fwd_dispatch(is_signed ? dqq::kKindSignedBox : dqq::kKindBox, a, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream));
}

int dqq_boxqp_fwd_f64(const double* P, const double* q, const double* l_min, const double* l_max, double* x, int64_t B,
                      int N, double eps, double mu_prox, int max_iter, int adaptive_rho, int p_layout, int* iters,
                      double* pdiag_out, unsigned char* diag_flags_out, void* workspace, size_t workspace_bytes,
                      void* stream)
{
    return box_fwd(P, q, l_min, l_max, nullptr, false, x, B, N, eps, mu_prox, max_iter, adaptive_rho, p_layout, iters,
                   pdiag_out, diag_flags_out, workspace, workspace_bytes, stream);
}

int dqq_signedboxqp_fwd_f64(const double* P, const double* q, const double* l_min, const double* l_max,
                            const double* v, double* x, int64_t B, int N, double eps, double mu_prox, int max_iter,
                            int adaptive_rho, int p_layout, int* iters, double* pdiag_out,
                            unsigned char* diag_flags_out, void* workspace, size_t workspace_bytes, void* stream)
{
    return box_fwd(P, q, l_min, l_max, v, true, x, B, N, eps, mu_prox, max_iter, adaptive_rho, p_layout, iters,
                   pdiag_out, diag_flags_out, workspace, workspace_bytes, stream);
}

int dqq_qp_bwd_f64(const double* P, const double* q, const double* x, const double* grad_x, double* grad_P,
                   double* grad_q, int64_t B, int N, double epsilon, int p_layout, int* ir_steps, const double* pdiag,
                   const unsigned char* diag_flags, unsigned long long* report, void* workspace, size_t workspace_bytes,
                   void* stream)
{
    if (int rc = check_common(B, N, p_layout, false)) return rc;
    if (B > 0 && (P == nullptr || q == nullptr || x == nullptr || grad_x == nullptr)) return DQQ_E_NULLPTR;
    dqq::BwdArgs a{P,     q,          nullptr, nullptr, x,       grad_x, grad_P,  grad_q,   nullptr,  nullptr,
                   pdiag, diag_flags, nullptr, nullptr, (long)B, N,      epsilon, layout_of(p_layout), ir_steps, nullptr};
    a.ref_order = ref_order_of(p_layout);
    a.hints = hints_of(p_layout);
    a.report = report;
    return bwd_dispatch(0, a, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int dqq_qcqp_bwd_f64(const double* P, const double* q, const double* l_n, const double* mu, const double* x,
                     const double* grad_x, double* grad_P, double* grad_q, double* grad_l_n, double* grad_mu,
                     double* gamma, double* dgamma, int64_t B, int N, double epsilon, int p_layout, int* ir_steps,
                     const double* pdiag, const unsigned char* diag_flags, unsigned long long* report, void* workspace,
                     size_t workspace_bytes, void* stream)
{
    if (int rc = check_common(B, N, p_layout, true)) return rc;
    if (B > 0 && (P == nullptr || q == nullptr || l_n == nullptr || mu == nullptr || x == nullptr ||
                  grad_x == nullptr))
        return DQQ_E_NULLPTR;
    dqq::BwdArgs a{P,     q,          l_n,   mu,     x,       grad_x, grad_P,  grad_q,   grad_l_n, grad_mu,
                   pdiag, diag_flags, gamma, dgamma, (long)B, N,      epsilon, layout_of(p_layout), ir_steps, nullptr};
    a.ref_order = ref_order_of(p_layout);
    a.hints = hints_of(p_layout);
    a.report = report;
    return bwd_dispatch(1, a, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int dqq_boxqp_bwd_f64(const double* P, const double* q, const double* l_min, const double* l_max, const double* x,
                      const double* grad_x, double* grad_P, double* grad_q, double* grad_l_min, double* grad_l_max,
                      double* gamma, double* dgamma, int64_t B, int N, double epsilon, int p_layout, int* ir_steps,
                      const double* pdiag, const unsigned char* diag_flags, void* workspace, size_t workspace_bytes,
                      void* stream)
{
    if (int rc = check_common(B, N, p_layout, false)) return rc;
    if (B > 0 && (P == nullptr || q == nullptr || l_min == nullptr || l_max == nullptr || x == nullptr ||
                  grad_x == nullptr))
        return DQQ_E_NULLPTR;
    dqq::BwdArgs a{P,     q,          l_min, l_max,  x,       grad_x, grad_P,  grad_q,   grad_l_min, grad_l_max,
                   pdiag, diag_flags, gamma, dgamma, (long)B, N,      epsilon, layout_of(p_layout), ir_steps,   nullptr};
    a.ref_order = ref_order_of(p_layout);
    return bwd_dispatch(dqq::kKindBox, a, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

} // extern "C"
