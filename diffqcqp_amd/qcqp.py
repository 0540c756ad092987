"""Drop-in for the reference's `qcqp.py`: the `QPFn2` and `QCQPFn2`
`torch.autograd.Function`s with the same forward/backward signatures, argument
order, output shapes and `None` padding (reference qcqp.py:22-52, 141-181), but
each pass is ONE launch of a hand-written HIP kernel over the whole batch
instead of a Python loop over per-problem C++ calls.

    from diffqcqp_amd.qcqp import QPFn2, QCQPFn2
    x = QPFn2.apply(P, q, warm_start, eps, max_iter)          # (B,N,1)
    x = QCQPFn2.apply(P, q, l_n, mu, warm_start, eps, max_iter)
    x = BoxQPFn2.apply(P, q, l_min, l_max, warm_start, eps, max_iter)            # qcqp.py:54-94
    x = SignedBoxQPFn2.apply(P, q, l_min, l_max, v, warm_start, eps, max_iter)   # qcqp.py:97-137, forward only

Behaviour kept from the reference:
  * importing this module sets torch's default dtype to float64 (qcqp.py:13);
  * `warm_start` is accepted and has no effect on the result (the reference
    overwrites it before reading it, Solver.cpp:70/80, 529/539); it gets no grad;
  * backward honours ctx.needs_input_grad and returns 6 / 8 values.
Differences: tensors on the GPU are used in place and results stay there; CPU
tensors are staged through cuda:0 and the result is returned on the CPU.  There
is no CPU solver in this package: without a GPU and the HIP library the calls
raise.
"""
import torch
from torch.autograd import Function

from . import ops

torch.set_default_dtype(torch.double)


_FAST_N = (2, 4, 8, 16, 32, 64)

# How the Functions below hand P to the C ABI (include/diffqcqp_hip.h: p_layout).  "auto" (default): every tile's
# off-diagonals are verified in-kernel, diagonal tiles take the fast path, the others the general kernels -- nothing is
# assumed.  "dense": straight to the general kernels, for callers who know their P is dense (a Delassus matrix): saves
# the verifying pass.  A module-level default because the reference's signatures (qcqp.py:24, 144) have no slot for it.
# "auto_expect_dense": "auto" with the hint flag DQQ_F_EXPECT_DENSE given by the CALLER instead of derived from a report word --
# still verified in-kernel, bit-identical results on any input (a wrong expectation costs time only), but an ARGUMENT: it
# holds on a first call and inside a captured HIP graph, where the report word's hints do not.
_LAYOUTS = {"auto": ops._capi.P_AUTO, "dense": ops._capi.P_DENSE,
            "auto_expect_dense": ops._capi.P_AUTO | ops._capi.F_EXPECT_DENSE}
_default_layout = ops._capi.P_AUTO


def set_default_layout(layout):
    """layout: "auto", "dense" or "auto_expect_dense".  Returns the previous setting (as a string)."""
    global _default_layout
    prev = [k for k, v in _LAYOUTS.items() if v == _default_layout][0]
    if layout not in _LAYOUTS:
        raise ValueError("layout must be one of %s" % sorted(_LAYOUTS))
    _default_layout = _LAYOUTS[layout]
    return prev


def get_default_layout():
    return [k for k, v in _LAYOUTS.items() if v == _default_layout][0]


def _cache_for(ctx, qd, n_inputs):
    """Buffers for the verified diagonal of P (forward -> backward of the same problems), only when a backward
    can follow (some input requires grad) and the diagonal fast path exists for this N."""
    if (_default_layout & 0xff) == ops._capi.P_AUTO and qd.shape[1] in _FAST_N and any(ctx.needs_input_grad[:n_inputs]):
        return ops.diag_cache(qd)
    return None


def _saved_cache(saved):
    return (saved[-2], saved[-1]) if saved[-1].dtype is torch.uint8 else None


def _device_for(t):
    if t.is_cuda:
        return t.device
    if not torch.cuda.is_available():
        raise RuntimeError("diffqcqp_amd needs an MI355X (ROCm) device: the solver is a HIP kernel and "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class QPFn2(Function):
    @staticmethod
    def forward(ctx, P, q, warm_start, eps, max_iter, mu_prox=1e-7):
        if q.is_cuda:  # resident tensors: no staging, no copies
            Pd, qd = P.detach(), q.detach()
        else:
            dev = _device_for(q)
            Pd, qd = P.detach().to(dev), q.detach().to(dev)
        cache = _cache_for(ctx, qd, 2)  # verified diagonal of P, reused by backward instead of re-reading P
        l_2 = ops.qp_forward(Pd, qd, eps, max_iter, mu_prox, adaptive_rho=True, cache=cache, layout=_default_layout)
        ctx.save_for_backward(Pd, qd, l_2, *(cache or ()))
        ctx.home = q.device
        ctx.layout = _default_layout  # the backward of these problems takes the same route
        return l_2 if q.is_cuda else l_2.to(q.device)

    @staticmethod
    def backward(ctx, grad_l):
        saved = ctx.saved_tensors
        P, q, l = saved[:3]
        need_P, need_q = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_P, grad_q = None, None
        if need_P or need_q:
            grad_P, grad_q = ops.qp_backward(P, q, l, grad_l.to(l.device), need_P, need_q, cache=_saved_cache(saved), layout=ctx.layout)
            if ctx.home != l.device:
                grad_P = None if grad_P is None else grad_P.to(ctx.home)
                grad_q = None if grad_q is None else grad_q.to(ctx.home)
        return grad_P, grad_q, None, None, None, None


class QCQPFn2(Function):
    @staticmethod
    def forward(ctx, P, q, l_n, mu, warm_start, eps, max_iter, mu_prox=1e-7):
        if q.is_cuda:
            Pd, qd, lnd, mud = P.detach(), q.detach(), l_n.detach(), mu.detach()
        else:
            dev = _device_for(q)
            Pd, qd = P.detach().to(dev), q.detach().to(dev)
            lnd, mud = l_n.detach().to(dev), mu.detach().to(dev)
        cache = _cache_for(ctx, qd, 4)
        l_2 = ops.qcqp_forward(Pd, qd, lnd, mud, eps, max_iter, mu_prox, adaptive_rho=True, cache=cache, layout=_default_layout)
        ctx.save_for_backward(Pd, qd, lnd, mud, l_2, *(cache or ()))
        ctx.home = q.device
        ctx.layout = _default_layout  # the backward of these problems takes the same route
        return l_2 if q.is_cuda else l_2.to(q.device)

    @staticmethod
    def backward(ctx, grad_l):
        saved = ctx.saved_tensors
        P, q, l_n, mu, l = saved[:5]
        need = tuple(ctx.needs_input_grad[0:4])
        grads = (None, None, None, None)
        if any(need):
            grads = ops.qcqp_backward(P, q, l_n, mu, l, grad_l.to(l.device), need, cache=_saved_cache(saved), layout=ctx.layout)
            if ctx.home != l.device:
                grads = tuple(None if g is None else g.to(ctx.home) for g in grads)
        return grads + (None, None, None, None)


class BoxQPFn2(Function):
    """min 1/2 x'Px + q'x, l_min <= x <= l_max (reference qcqp.py:54-94).

    The reference's forward works; its backward does not run (wrong unpack counts, swapped saved tensors,
    `.asDiagonal()` on a tensor -- SURVEY.md section 2 #7).  This backward computes what that code spells out,
    grad_P = -dl l', grad_q = -dl, grad_l_min = -dgamma_lo*gamma_lo, and grad_l_max = +dgamma_hi*gamma_hi: the
    reference writes a minus sign there (qcqp.py:93), finite differences say plus (tests/test_oracle.py)."""

    @staticmethod
    def forward(ctx, P, q, l_min, l_max, warm_start, eps, max_iter, mu_prox=1e-7):
        tensors = (P, q, l_min, l_max)
        if q.is_cuda:
            Pd, qd, lod, hid = (t.detach() for t in tensors)
        else:
            dev = _device_for(q)
            Pd, qd, lod, hid = (t.detach().to(dev) for t in tensors)
        cache = _cache_for(ctx, qd, 4)
        l_2 = ops.boxqp_forward(Pd, qd, lod, hid, eps, max_iter, mu_prox=mu_prox, adaptive_rho=True, cache=cache, layout=_default_layout)
        ctx.save_for_backward(Pd, qd, lod, hid, l_2, *(cache or ()))
        ctx.home = q.device
        ctx.layout = _default_layout  # the backward of these problems takes the same route
        return l_2 if q.is_cuda else l_2.to(q.device)

    @staticmethod
    def backward(ctx, grad_l):
        saved = ctx.saved_tensors
        P, q, l_min, l_max, l = saved[:5]
        need = tuple(ctx.needs_input_grad[0:4])
        grads = (None, None, None, None)
        if any(need):
            grads = ops.boxqp_backward(P, q, l_min, l_max, l, grad_l.to(l.device), need, cache=_saved_cache(saved), layout=ctx.layout)
            if ctx.home != l.device:
                grads = tuple(None if g is None else g.to(ctx.home) for g in grads)
        return grads + (None, None, None, None)


class SignedBoxQPFn2(Function):
    """The box QP with the extra constraint sign(v_i) x_i <= 0 (reference qcqp.py:97-137).  Forward only: the
    reference marks its backward "not implemented" (qcqp.py:111) -- it would differentiate the plain box QP,
    ignoring v -- so asking for a gradient raises instead of returning something wrong."""

    @staticmethod
    def forward(ctx, P, q, l_min, l_max, v, warm_start, eps, max_iter, mu_prox=1e-7):
        tensors = (P, q, l_min, l_max, v)
        if q.is_cuda:
            Pd, qd, lod, hid, vd = (t.detach() for t in tensors)
        else:
            dev = _device_for(q)
            Pd, qd, lod, hid, vd = (t.detach().to(dev) for t in tensors)
        l_2 = ops.boxqp_forward(Pd, qd, lod, hid, eps, max_iter, v=vd, mu_prox=mu_prox, adaptive_rho=True,
                                layout=_default_layout)
        return l_2 if q.is_cuda else l_2.to(q.device)

    @staticmethod
    def backward(ctx, grad_l):
        raise NotImplementedError("SignedBoxQPFn2 has no backward (not implemented in the reference either, "
                                  "qcqp.py:111)")
