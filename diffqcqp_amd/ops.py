"""Tensor-level entry points over the C ABI: one asynchronous launch per batch.

Inputs are float64 CUDA (ROCm) tensors in the reference's layouts -- P (B,N,N),
q (B,N,1), l_n / mu (B,N/2,1) -- and every call runs on torch's current stream.
torch is used here for device memory and streams only.
"""
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


def _workspace(device, B, stream=None, kind=0, pas=0, N=8):
    """Zero-initialised fallback work-list (+ the scratch of the global-memory kernels behind it, when (kind, pas, N)
    needs any), cached per (device, stream); the kernels leave the work-list empty again
    (include/diffqcqp_hip.h: dqq_workspace_bytes, dqq_scratch_bytes)."""
    if stream is None:
        stream = _raw_stream(device.index)
    key = (device.index, stream)
    lib = _capi.lib()
    need = lib.dqq_workspace_bytes(int(B))
    if N > 21:  # dqq_max_n: nothing below needs scratch
        need += lib.dqq_scratch_bytes(int(kind), int(pas), int(N), int(B))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.zeros(max((need + 3) // 4, 1024), dtype=torch.int32, device=device)
        _workspaces.pop(key, None)
        _workspaces[key] = ws
        while len(_workspaces) > _MAX_WORKSPACES:
            _workspaces.pop(next(iter(_workspaces)))
    elif len(_workspaces) > 1:
        _workspaces[key] = _workspaces.pop(key)  # most recently used last
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
    B, N = q.shape[0], q.shape[1]
    pshape = (B, N) if layout == _capi.P_DIAG else (B, N, N)
    return B, N, pshape


def diag_cache(q):
    """Buffers a forward can fill for the backward of the same problems: (pdiag (B,N), flags (B) uint8)."""
    B, N = q.shape[0], q.shape[1]
    return (torch.empty((B, N), dtype=torch.float64, device=q.device),
            torch.empty(B, dtype=torch.uint8, device=q.device))


def qp_forward(P, q, eps, max_iter, mu_prox=1e-7, adaptive_rho=True, layout=_capi.P_AUTO, return_iters=False,
               out=None, cache=None):
    """Batched QP solve min 1/2 x'Px + q'x, x >= 0 (reference qcqp.py:24-33). -> x (B,N,1).
    cache: optional `diag_cache(q)` buffers; pass the same pair to qp_backward (P must be unchanged)."""
    B, N, pshape = _dims(P, q, layout)
    P, q = _prep(P, "P", pshape), _prep(q, "q", (B, N, 1))
    x = _out(out, (B, N, 1), "out") if out is not None else torch.empty((B, N, 1), dtype=torch.float64, device=q.device)
    iters = torch.empty(B, dtype=torch.int32, device=q.device) if return_iters else None
    stream = _raw_stream(q.device.index)
    ws = _workspace(q.device, B, stream, 0, 0, N)
    with _device_guard(q.device):
        pd, fl = cache if cache is not None else (None, None)
        rc = _capi.lib().dqq_qp_fwd_f64(_ptr(P), _ptr(q), _ptr(x), B, N, float(eps), float(mu_prox), int(max_iter),
                                        int(bool(adaptive_rho)), layout, _ptr(iters), _ptr(pd), _ptr(fl), _ptr(ws),
                                        ws.numel() * 4, stream)
    _capi.check(rc, "dqq_qp_fwd_f64")
    return (x, iters) if return_iters else x


def qcqp_forward(P, q, l_n, mu, eps, max_iter, mu_prox=1e-7, adaptive_rho=True, layout=_capi.P_AUTO,
                 return_iters=False, out=None, cache=None):
    """Batched QCQP solve, ||x_(i)|| <= mu_i*l_n_i per contact (reference qcqp.py:144-153)."""
    B, N, pshape = _dims(P, q, layout)
    P, q = _prep(P, "P", pshape), _prep(q, "q", (B, N, 1))
    l_n, mu = _prep(l_n, "l_n", (B, N // 2, 1)), _prep(mu, "mu", (B, N // 2, 1))
    x = _out(out, (B, N, 1), "out") if out is not None else torch.empty((B, N, 1), dtype=torch.float64, device=q.device)
    iters = torch.empty(B, dtype=torch.int32, device=q.device) if return_iters else None
    stream = _raw_stream(q.device.index)
    ws = _workspace(q.device, B, stream, 1, 0, N)
    with _device_guard(q.device):
        pd, fl = cache if cache is not None else (None, None)
        rc = _capi.lib().dqq_qcqp_fwd_f64(_ptr(P), _ptr(q), _ptr(l_n), _ptr(mu), _ptr(x), B, N, float(eps),
                                          float(mu_prox), int(max_iter), int(bool(adaptive_rho)), layout, _ptr(iters),
                                          _ptr(pd), _ptr(fl), _ptr(ws), ws.numel() * 4, stream)
    _capi.check(rc, "dqq_qcqp_fwd_f64")
    return (x, iters) if return_iters else x


def boxqp_forward(P, q, l_min, l_max, eps, max_iter, v=None, mu_prox=1e-7, adaptive_rho=True, layout=_capi.P_AUTO,
                  return_iters=False, out=None, cache=None):
    """Batched box QP solve, l_min <= x <= l_max (reference qcqp.py:56-65); with `v` the signed box QP,
    additionally sign(v_i) x_i <= 0 (reference qcqp.py:99-108)."""
    B, N, pshape = _dims(P, q, layout)
    P, q = _prep(P, "P", pshape), _prep(q, "q", (B, N, 1))
    l_min, l_max = _prep(l_min, "l_min", (B, N, 1)), _prep(l_max, "l_max", (B, N, 1))
    x = _out(out, (B, N, 1), "out") if out is not None else torch.empty((B, N, 1), dtype=torch.float64, device=q.device)
    iters = torch.empty(B, dtype=torch.int32, device=q.device) if return_iters else None
    stream = _raw_stream(q.device.index)
    ws = _workspace(q.device, B, stream, 2 if v is None else 3, 0, N)
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
                epsilon=1e-10, cache=None):
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
    ws = _workspace(dev, B, stream, 0, 1, N)
    with _device_guard(dev):
        pd, fl = cache if cache is not None else (None, None)
        rc = _capi.lib().dqq_qp_bwd_f64(_ptr(P), _ptr(q), _ptr(x), _ptr(grad_x), _ptr(gP), _ptr(gq), B, N,
                                        float(epsilon), layout, _ptr(steps), _ptr(pd), _ptr(fl), _ptr(ws),
                                        ws.numel() * 4, stream)
    _capi.check(rc, "dqq_qp_bwd_f64")
    return (gP, gq, steps) if return_steps else (gP, gq)


def qcqp_backward(P, q, l_n, mu, x, grad_x, need=(True, True, True, True), layout=_capi.P_AUTO, return_steps=False,
                  out=None, epsilon=1e-10, duals=None, cache=None):
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
    ws = _workspace(dev, B, stream, 1, 1, N)
    with _device_guard(dev):
        gam, dgam = duals if duals is not None else (None, None)
        pd, fl = cache if cache is not None else (None, None)
        rc = _capi.lib().dqq_qcqp_bwd_f64(_ptr(P), _ptr(q), _ptr(l_n), _ptr(mu), _ptr(x), _ptr(grad_x), _ptr(gP),
                                          _ptr(gq), _ptr(gl), _ptr(gm), _ptr(gam), _ptr(dgam), B, N, float(epsilon),
                                          layout, _ptr(steps), _ptr(pd), _ptr(fl), _ptr(ws), ws.numel() * 4, stream)
    _capi.check(rc, "dqq_qcqp_bwd_f64")
    return (gP, gq, gl, gm, steps) if return_steps else (gP, gq, gl, gm)


def boxqp_backward(P, q, l_min, l_max, x, grad_x, need=(True, True, True, True), layout=_capi.P_AUTO,
                   return_steps=False, out=None, duals=None, epsilon=1e-10, cache=None):
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
    ws = _workspace(dev, B, stream, 2, 1, N)
    with _device_guard(dev):
        gam, dgam = duals if duals is not None else (None, None)
        pd, fl = cache if cache is not None else (None, None)
        rc = _capi.lib().dqq_boxqp_bwd_f64(_ptr(P), _ptr(q), _ptr(l_min), _ptr(l_max), _ptr(x), _ptr(grad_x), _ptr(gP),
                                           _ptr(gq), _ptr(glo), _ptr(ghi), _ptr(gam), _ptr(dgam), B, N, float(epsilon),
                                           layout, _ptr(steps), _ptr(pd), _ptr(fl), _ptr(ws), ws.numel() * 4, stream)
    _capi.check(rc, "dqq_boxqp_bwd_f64")
    return (gP, gq, glo, ghi, steps) if return_steps else (gP, gq, glo, ghi)
