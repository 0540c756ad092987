"""Drop-in for the reference's `qcqp_no_batch.py`: the unbatched twins of `QPFn2` / `QCQPFn2`
(reference qcqp_no_batch.py:23-51, 54-108) for ONE problem with `P (N,N)`, `q (N,1)`, `l_n, mu (N/2,1)`.

    from diffqcqp_amd.qcqp_no_batch import QPFn2, QCQPFn2
    l = QPFn2.apply(P, q, warm_start, eps, max_iter)        # (N,)   (the reference returns the 1-D solution)

Same return shapes as the reference: the solution is 1-D `(N,)`, `grad_P` is `(N,N)`, `grad_q` is `(N,1)`,
`grad_l_n` / `grad_mu` are `(N/2,1)`.  A shape adapter over the batched HIP path with B = 1 -- a single
small problem cannot use the GPU well; batch your problems and use `diffqcqp_amd.qcqp` when you can.
"""
import torch
from torch.autograd import Function

from . import ops
from .qcqp import _device_for

torch.set_default_dtype(torch.double)


def _b(t, dev, shape):
    return t.detach().to(dev).reshape(shape).contiguous()


class QPFn2(Function):
    @staticmethod
    def forward(ctx, P, q, warm_start, eps, max_iter, mu_prox=1e-7):
        dev = _device_for(q)
        n = q.numel()
        Pd, qd = _b(P, dev, (1, n, n)), _b(q, dev, (1, n, 1))
        l_2 = ops.qp_forward(Pd, qd, eps, max_iter, mu_prox, adaptive_rho=True)
        ctx.save_for_backward(Pd, qd, l_2)
        ctx.home = q.device
        return l_2.reshape(n).to(q.device)

    @staticmethod
    def backward(ctx, grad_l):
        P, q, l = ctx.saved_tensors
        n = q.shape[1]
        need_P, need_q = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_P, grad_q = None, None
        if need_P or need_q:
            gP, gq = ops.qp_backward(P, q, l, _b(grad_l, l.device, (1, n, 1)), need_P, need_q)
            grad_P = None if gP is None else gP.reshape(n, n).to(ctx.home)   # -dl l^T, qcqp_no_batch.py:47
            grad_q = None if gq is None else gq.reshape(n, 1).to(ctx.home)   # -dl.unsqueeze(-1), :49
        return grad_P, grad_q, None, None, None, None


class QCQPFn2(Function):
    @staticmethod
    def forward(ctx, P, q, l_n, mu, warm_start, eps, max_iter, mu_prox=1e-7):
        dev = _device_for(q)
        n = q.numel()
        Pd, qd = _b(P, dev, (1, n, n)), _b(q, dev, (1, n, 1))
        lnd, mud = _b(l_n, dev, (1, n // 2, 1)), _b(mu, dev, (1, n // 2, 1))
        l_2 = ops.qcqp_forward(Pd, qd, lnd, mud, eps, max_iter, mu_prox, adaptive_rho=True)
        ctx.save_for_backward(Pd, qd, lnd, mud, l_2)
        ctx.home = q.device
        return l_2.reshape(n).to(q.device)

    @staticmethod
    def backward(ctx, grad_l):
        P, q, l_n, mu, l = ctx.saved_tensors
        n = q.shape[1]
        need = tuple(ctx.needs_input_grad[0:4])
        out = [None, None, None, None]
        if any(need):
            g = ops.qcqp_backward(P, q, l_n, mu, l, _b(grad_l, l.device, (1, n, 1)), need)
            shapes = ((n, n), (n, 1), (n // 2, 1), (n // 2, 1))
            out = [None if t is None else t.reshape(s).to(ctx.home) for t, s in zip(g, shapes)]
        return tuple(out) + (None, None, None, None)
