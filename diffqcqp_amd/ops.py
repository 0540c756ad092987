"""Tensor-level entry points over the C ABI: one asynchronous launch per batch.

Inputs are float64 CUDA (ROCm) tensors in the reference's layouts -- P (B,N,N),
q (B,N,1), l_n / mu (B,N/2,1) -- and every call runs on torch's current stream.
torch is used here for device memory and streams only.

Workspace: every op takes `workspace=` (ops.make_workspace); without it a per-(device, stream) cache is used.  HIP-graph
capture: warm the capture stream up with one eager call first (or pass workspace=); a cached workspace that was live
during a capture is never replaced or freed afterwards, so a replayed graph cannot write into recycled memory.
"""
import os

import torch

from . import _capi

_workspaces = {}

try:  # raw stream handle of torch's current stream without building a Stream object (private but stable)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream


class _device_guard:
    """`with torch.cuda.device(dev)` only when dev is not already current (the context manager costs
    several microseconds per call, the hot path is tens of microseconds)."""

    def __init__(self, dev):
        self.ctx = None if dev.index == torch.cuda.current_device() else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


_MAX_WORKSPACES = 16   # cached (device, stream) pairs; the least recently used one is dropped beyond that
_pinned = []           # workspaces handed out while a stream capture was going on: a captured graph has their address
                       # baked in, so they are never replaced, evicted or freed
_need_cache = {}       # (kind, pas, N, B, flags) -> bytes: dqq_workspace_bytes + dqq_scratch_bytes are functions of exactly these


def _hints(kind, pas, N, B, layout, dev):
    """(flags, report address) for this call: the caller's side of include/diffqcqp_hip.h's hint protocol.  Only DQQ_P_AUTO
    batches of QP / QCQP, N <= 8; a caller that already passes hint flags in `layout` keeps them."""
    if (layout & 0xff) != _capi.P_AUTO or N > 8 or kind > 1 or (layout & (_capi.F_EXPECT_DENSE | _capi.F_EXPECT_LONG_LIST)):
        return 0, None
    flags, report = _capi.hint(kind, pas, N, B, dev.index)
    if flags and _capturing():   # (asked only when a hint is about to be followed) a graph is replayed on batches the word
        flags = 0                # knows nothing about: what goes into it is the argument-determined route
    return flags, report


def _capturing():
    try:
        return torch.cuda.is_current_stream_capturing()
    except Exception:  # pragma: no cover
        return False


def workspace_bytes(B, kind=0, pas=0, N=8, layout=0):
    """Bytes of `workspace` a call needs: the work-list + whatever scratch the kernels of (kind, pas, N, B, layout flags) use
    -- asked of the library (dqq_workspace_bytes + dqq_scratch_bytes), never guessed here."""
    key = (int(kind), int(pas), int(N), int(B), int(layout) & _capi.F_REFERENCE_ORDER)
    need = _need_cache.get(key)
    if need is None:
        lib = _capi.lib()
        need = lib.dqq_workspace_bytes(int(B)) + lib.dqq_scratch_bytes(*key)
        if len(_need_cache) >= 1024:   # (B varies freely in a caller's hands: bounded)
            _need_cache.clear()
        _need_cache[key] = need
    return need


def make_workspace(device, B, kind=0, pas=0, N=8, layout=0):
    """A caller-owned workspace for the `workspace=` argument of the ops below (int32, zero-initialised work-list header,
    uninitialised scratch behind it).  A caller that captures ops calls into a HIP graph should own one per stream and
    keep it alive as long as the graph: the captured launches hold its address."""
    lib = _capi.lib()
    head = lib.dqq_workspace_bytes(int(B))
    need = max(workspace_bytes(B, kind, pas, N, layout), 4096)
    ws = torch.empty((need + 3) // 4, dtype=torch.int32, device=device)
    ws[: (head + 3) // 4].zero_()     # only the work-list must start zeroed (= dqq_workspace_reset); the scratch needs no initialisation
    return ws


def _workspace(device, B, stream=None, kind=0, pas=0, N=8, given=None, layout=0):
    """The work-list (+ the scratch of the global-memory kernels behind it, when (kind, pas, N) needs any): the
    caller's (`given`, checked for size) or one cached per (device, stream); the kernels leave the work-list empty again
    (include/diffqcqp_hip.h: dqq_workspace_bytes, dqq_scratch_bytes).
    Lifetime rule of the cache: an entry is replaced when a larger call arrives and evicted beyond 16 streams -- except
    entries that were handed out during a stream capture, which stay alive for the life of the process."""
    need = workspace_bytes(B, kind, pas, N, layout)
    if given is not None:
        if not (given.is_cuda and given.dtype is torch.int32 and given.is_contiguous() and given.numel() * 4 >= need):
            raise ValueError("workspace must be a contiguous int32 GPU tensor of at least %d bytes (ops.make_workspace)" % need)
        return given
    if stream is None:
        stream = _raw_stream(device.index)
    key = (device.index, stream)
    ws = _workspaces.get(key)
    cap = _capturing()
    if ws is None or ws.numel() * 4 < need:
        if cap:   # (created OR grown: either way an allocation and a zero-fill would be captured, and the cache entry replaced)
            raise RuntimeError("diffqcqp_amd.ops: a call inside a stream capture needs a workspace that is not there yet (first "
                               "call on this stream, or a larger batch than any before) -- it would be allocated and zero-filled "
                               "inside the capture; warm the stream up with the largest call first or pass workspace=")
        ws = make_workspace(device, B, kind, pas, N, layout)
        _workspaces.pop(key, None)
        _workspaces[key] = ws
        while len(_workspaces) > _MAX_WORKSPACES:
            _workspaces.pop(next(iter(_workspaces)))   # (pinned ones stay referenced by _pinned)
    elif len(_workspaces) > 1:
        _workspaces[key] = _workspaces.pop(key)  # most recently used last
    if cap and not any(w is ws for w in _pinned):
        _pinned.append(ws)
    return ws


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _prep(t, name, shape=None):
    if t.is_cuda and t.dtype is torch.float64 and t.is_contiguous() and (shape is None or t.shape == shape):
        return t  # the common case: nothing to do (data_ptr() ignores autograd state)
    if not t.is_cuda:
        raise ValueError("%s must live on the GPU (got %s)" % (name, t.device))
    if t.dtype != torch.float64:
        t = t.to(torch.float64)
    t = t.detach().contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t


def _out(t, shape, name):
    """A caller-provided output buffer: the kernels write float64 into it, so anything else must be refused here."""
    if t is None:
        return None
    if not (t.is_cuda and t.dtype is torch.float64 and t.is_contiguous() and tuple(t.shape) == tuple(shape)):
        raise ValueError("%s must be a contiguous float64 GPU tensor of shape %s (got %s %s on %s)"
                         % (name, tuple(shape), t.dtype, tuple(t.shape), t.device))
    return t


def _dims(P, q, layout):
    """layout: DQQ_P_* optionally ORed with _capi.F_REFERENCE_ORDER (include/diffqcqp_hip.h), passed to the C ABI as is."""
    B, N = q.shape[0], q.shape[1]
    pshape = (B, N) if (layout & 0xff) == _capi.P_DIAG else (B, N, N)
    return B, N, pshape


def diag_cache(q):
    """Buffers a forward can fill for the backward of the same problems: (pdiag (B,N), flags (B) uint8)."""
    B, N = q.shape[0], q.shape[1]
    return (torch.empty((B, N), dtype=torch.float64, device=q.device),
            torch.empty(B, dtype=torch.uint8, device=q.device))


def qp_forward(P, q, eps, max_iter, mu_prox=1e-7, adaptive_rho=True, layout=_capi.P_AUTO, return_iters=False,
               out=None, cache=None, workspace=None):
    """Batched QP solve min 1/2 x'Px + q'x, x >= 0 (reference qcqp.py:24-33). -> x (B,N,1).
    cache: optional `diag_cache(q)` buffers; pass the same pair to qp_backward (P must be unchanged)."""
    B, N, pshape = _dims(P, q, layout)
    P, q = _prep(P, "P", pshape), _prep(q, "q", (B, N, 1))
    x = _out(out, (B, N, 1), "out") if out is not None else torch.empty((B, N, 1), dtype=torch.float64, device=q.device)
    iters = torch.empty(B, dtype=torch.int32, device=q.device) if return_iters else None
    stream = _raw_stream(q.device.index)
    ws = _workspace(q.device, B, stream, 0, 0, N, workspace, layout)
    hf, _ = _hints(0, 0, N, B, layout, q.device)
    with _device_guard(q.device):
        pd, fl = cache if cache is not None else (None, None)
        rc = _capi.lib().dqq_qp_fwd_f64(_ptr(P), _ptr(q), _ptr(x), B, N, float(eps), float(mu_prox), int(max_iter),
                                        int(bool(adaptive_rho)), layout | hf, _ptr(iters), _ptr(pd), _ptr(fl), _ptr(ws),
                                        ws.numel() * 4, stream)
    _capi.check(rc, "dqq_qp_fwd_f64")
    return (x, iters) if return_iters else x


def qcqp_forward(P, q, l_n, mu, eps, max_iter, mu_prox=1e-7, adaptive_rho=True, layout=_capi.P_AUTO,
                 return_iters=False, out=None, cache=None, workspace=None):
    """Batched QCQP solve, ||x_(i)|| <= mu_i*l_n_i per contact (reference qcqp.py:144-153)."""
    B, N, pshape = _dims(P, q, layout)
    P, q = _prep(P, "P", pshape), _prep(q, "q", (B, N, 1))
    l_n, mu = _prep(l_n, "l_n", (B, N // 2, 1)), _prep(mu, "mu", (B, N // 2, 1))
    x = _out(out, (B, N, 1), "out") if out is not None else torch.empty((B, N, 1), dtype=torch.float64, device=q.device)
    iters = torch.empty(B, dtype=torch.int32, device=q.device) if return_iters else None
    stream = _raw_stream(q.device.index)
    ws = _workspace(q.device, B, stream, 1, 0, N, workspace, layout)
    hf, _ = _hints(1, 0, N, B, layout, q.device)
    with _device_guard(q.device):
        pd, fl = cache if cache is not None else (None, None)
        rc = _capi.lib().dqq_qcqp_fwd_f64(_ptr(P), _ptr(q), _ptr(l_n), _ptr(mu), _ptr(x), B, N, float(eps),
                                          float(mu_prox), int(max_iter), int(bool(adaptive_rho)), layout | hf, _ptr(iters),
                                          _ptr(pd), _ptr(fl), _ptr(ws), ws.numel() * 4, stream)
    _capi.check(rc, "dqq_qcqp_fwd_f64")
    return (x, iters) if return_iters else x


def boxqp_forward(P, q, l_min, l_max, eps, max_iter, v=None, mu_prox=1e-7, adaptive_rho=True, layout=_capi.P_AUTO,
                  return_iters=False, out=None, cache=None, workspace=None):
    """Batched box QP solve, l_min <= x <= l_max (reference qcqp.py:56-65); with `v` the signed box QP,
    additionally sign(v_i) x_i <= 0 (reference qcqp.py:99-108)."""
    B, N, pshape = _dims(P, q, layout)
    P, q = _prep(P, "P", pshape), _prep(q, "q", (B, N, 1))
    l_min, l_max = _prep(l_min, "l_min", (B, N, 1)), _prep(l_max, "l_max", (B, N, 1))
    x = _out(out, (B, N, 1), "out") if out is not None else torch.empty((B, N, 1), dtype=torch.float64, device=q.device)
    iters = torch.empty(B, dtype=torch.int32, device=q.device) if return_iters else None
    stream = _raw_stream(q.device.index)
    ws = _workspace(q.device, B, stream, 2 if v is None else 3, 0, N, workspace, layout)
    with _device_guard(q.device):
        pd, fl = cache if cache is not None else (None, None)
        tail = (B, N, float(eps), float(mu_prox), int(max_iter), int(bool(adaptive_rho)), layout, _ptr(iters), _ptr(pd),
                _ptr(fl), _ptr(ws), ws.numel() * 4, stream)
        if v is None:
            what = "dqq_boxqp_fwd_f64"
            rc = _capi.lib().dqq_boxqp_fwd_f64(_ptr(P), _ptr(q), _ptr(l_min), _ptr(l_max), _ptr(x), *tail)
        else:
            what = "dqq_signedboxqp_fwd_f64"
            v = _prep(v, "v", (B, N, 1))
            rc = _capi.lib().dqq_signedboxqp_fwd_f64(_ptr(P), _ptr(q), _ptr(l_min), _ptr(l_max), _ptr(v), _ptr(x), *tail)
    _capi.check(rc, what)
    return (x, iters) if return_iters else x


def qp_backward(P, q, x, grad_x, need_P=True, need_q=True, layout=_capi.P_AUTO, return_steps=False, out=None,
                epsilon=1e-10, cache=None, workspace=None):
    """Implicit-function backward of the QP (reference qcqp.py:36-52). -> (grad_P|None, grad_q|None)"""
    B, N, pshape = _dims(P, q, layout)
    P, q = _prep(P, "P", pshape), _prep(q, "q", (B, N, 1))
    x, grad_x = _prep(x, "x", (B, N, 1)), _prep(grad_x, "grad_x", (B, N, 1))
    dev = q.device
    if out is not None:
        gP, gq = _out(out[0], pshape, "out[0]"), _out(out[1], (B, N, 1), "out[1]")
    else:
        gP = torch.empty(pshape, dtype=torch.float64, device=dev) if need_P else None
        gq = torch.empty((B, N, 1), dtype=torch.float64, device=dev) if need_q else None
    steps = torch.empty(B, dtype=torch.int32, device=dev) if return_steps else None
    stream = _raw_stream(dev.index)
    ws = _workspace(dev, B, stream, 0, 1, N, workspace, layout)
    hf, report = _hints(0, 1, N, B, layout, dev)
    with _device_guard(dev):
        pd, fl = cache if cache is not None else (None, None)
        rc = _capi.lib().dqq_qp_bwd_f64(_ptr(P), _ptr(q), _ptr(x), _ptr(grad_x), _ptr(gP), _ptr(gq), B, N,
                                        float(epsilon), layout | hf, _ptr(steps), _ptr(pd), _ptr(fl), report, _ptr(ws),
                                        ws.numel() * 4, stream)
    _capi.check(rc, "dqq_qp_bwd_f64")
    return (gP, gq, steps) if return_steps else (gP, gq)


def qcqp_backward(P, q, l_n, mu, x, grad_x, need=(True, True, True, True), layout=_capi.P_AUTO, return_steps=False,
                  out=None, epsilon=1e-10, duals=None, cache=None, workspace=None):
    """Implicit-function backward of the QCQP (reference qcqp.py:156-181).
    -> (grad_P, grad_q, grad_l_n, grad_mu), None where not needed.  duals: optional pair of (B,N/2,1)
    tensors that receive the contact duals gamma and their derivative terms dgamma."""
    B, N, pshape = _dims(P, q, layout)
    P, q = _prep(P, "P", pshape), _prep(q, "q", (B, N, 1))
    l_n, mu = _prep(l_n, "l_n", (B, N // 2, 1)), _prep(mu, "mu", (B, N // 2, 1))
    x, grad_x = _prep(x, "x", (B, N, 1)), _prep(grad_x, "grad_x", (B, N, 1))
    dev = q.device
    if out is not None:
        gP, gq = _out(out[0], pshape, "out[0]"), _out(out[1], (B, N, 1), "out[1]")
        gl, gm = _out(out[2], (B, N // 2, 1), "out[2]"), _out(out[3], (B, N // 2, 1), "out[3]")
    else:
        gP = torch.empty(pshape, dtype=torch.float64, device=dev) if need[0] else None
        gq = torch.empty((B, N, 1), dtype=torch.float64, device=dev) if need[1] else None
        gl = torch.empty((B, N // 2, 1), dtype=torch.float64, device=dev) if need[2] else None
        gm = torch.empty((B, N // 2, 1), dtype=torch.float64, device=dev) if need[3] else None
    steps = torch.empty(B, dtype=torch.int32, device=dev) if return_steps else None
    stream = _raw_stream(dev.index)
    ws = _workspace(dev, B, stream, 1, 1, N, workspace, layout)
    hf, report = _hints(1, 1, N, B, layout, dev)
    with _device_guard(dev):
        gam, dgam = duals if duals is not None else (None, None)
        pd, fl = cache if cache is not None else (None, None)
        rc = _capi.lib().dqq_qcqp_bwd_f64(_ptr(P), _ptr(q), _ptr(l_n), _ptr(mu), _ptr(x), _ptr(grad_x), _ptr(gP),
                                          _ptr(gq), _ptr(gl), _ptr(gm), _ptr(gam), _ptr(dgam), B, N, float(epsilon),
                                          layout | hf, _ptr(steps), _ptr(pd), _ptr(fl), report, _ptr(ws), ws.numel() * 4,
                                          stream)
    _capi.check(rc, "dqq_qcqp_bwd_f64")
    return (gP, gq, gl, gm, steps) if return_steps else (gP, gq, gl, gm)


def boxqp_backward(P, q, l_min, l_max, x, grad_x, need=(True, True, True, True), layout=_capi.P_AUTO,
                   return_steps=False, out=None, duals=None, epsilon=1e-10, cache=None, workspace=None):
    """Implicit-function backward of the box QP: what BoxQPFn2.backward (reference qcqp.py:67-94) spells out,
    with the signs finite differences confirm.  -> (grad_P, grad_q, grad_l_min, grad_l_max), None where not
    needed.  duals: optional pair of (B,2N) tensors that receive gamma and dgamma ([lower | upper]).
    return_steps: also the (B,2) refinement step counts (dual recovery, derivative system)."""
    B, N, pshape = _dims(P, q, layout)
    P, q = _prep(P, "P", pshape), _prep(q, "q", (B, N, 1))
    l_min, l_max = _prep(l_min, "l_min", (B, N, 1)), _prep(l_max, "l_max", (B, N, 1))
    x, grad_x = _prep(x, "x", (B, N, 1)), _prep(grad_x, "grad_x", (B, N, 1))
    dev = q.device
    if out is not None:
        gP, gq = _out(out[0], pshape, "out[0]"), _out(out[1], (B, N, 1), "out[1]")
        glo, ghi = _out(out[2], (B, N, 1), "out[2]"), _out(out[3], (B, N, 1), "out[3]")
    else:
        gP = torch.empty(pshape, dtype=torch.float64, device=dev) if need[0] else None
        gq = torch.empty((B, N, 1), dtype=torch.float64, device=dev) if need[1] else None
        glo = torch.empty((B, N, 1), dtype=torch.float64, device=dev) if need[2] else None
        ghi = torch.empty((B, N, 1), dtype=torch.float64, device=dev) if need[3] else None
    steps = torch.empty((B, 2), dtype=torch.int32, device=dev) if return_steps else None
    stream = _raw_stream(dev.index)
    ws = _workspace(dev, B, stream, 2, 1, N, workspace, layout)
    with _device_guard(dev):
        gam, dgam = duals if duals is not None else (None, None)
        pd, fl = cache if cache is not None else (None, None)
        rc = _capi.lib().dqq_boxqp_bwd_f64(_ptr(P), _ptr(q), _ptr(l_min), _ptr(l_max), _ptr(x), _ptr(grad_x), _ptr(gP),
                                           _ptr(gq), _ptr(glo), _ptr(ghi), _ptr(gam), _ptr(dgam), B, N, float(epsilon),
                                           layout, _ptr(steps), _ptr(pd), _ptr(fl), _ptr(ws), ws.numel() * 4, stream)
    _capi.check(rc, "dqq_boxqp_bwd_f64")
    return (gP, gq, glo, ghi, steps) if return_steps else (gP, gq, glo, ghi)
