"""Module-level numpy API with the names, argument order and keyword defaults of the reference's
pybind11 module `diffqcqp` (reference pybindings.cpp:74-83), for consumers that call the raw solver
instead of the autograd Functions:

    from diffqcqp_amd.diffqcqp import solveQP, solveQCQP, solveDerivativesQP, solveDerivativesQCQP
    from diffqcqp_amd.diffqcqp import solveBoxQP, solveSignedBoxQP, solveDerivativesBoxQP

Each call solves ONE problem given as numpy arrays (vectors may be (N,) or (N,1), any float dtype --
converted to float64 like pybind11 does), on the GPU through the same C ABI as the batched path
(B = 1, host buffers staged over PCIe), and returns fresh numpy arrays shaped like the reference's:
`solveQP/solveQCQP -> (N,)`, `solveDerivativesQP -> (N,)`, `solveDerivativesQCQP -> (E1 (nc,nc),
E2 (nc,nc), blgamma (nc+N,))`, `solveBoxQP/solveSignedBoxQP -> (N,)`, `solveDerivativesBoxQP ->
(blgamma (3N,), gamma (2N,))`.  `warm_start` is accepted and ignored (dead in the reference).
The batched functions `*_batch` take stacked problems and are what you want for throughput.
"""
import numpy as np
import torch

from . import ops
from .qcqp import _device_for


def _dev():
    return _device_for(torch.empty(0))


def _t(a, shape, dev):
    return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64))).reshape(shape).to(dev)


def solveQP(P, q, warm_start, epsilon=1e-10, mu_prox=1e-7, max_iter=1000, adaptative_rho=True):
    dev = _dev()
    n = np.asarray(q).size
    x = ops.qp_forward(_t(P, (1, n, n), dev), _t(q, (1, n, 1), dev), epsilon, max_iter, mu_prox,
                       adaptive_rho=adaptative_rho)
    return x.reshape(n).cpu().numpy()


def solveQCQP(P, q, l_n, mu, warm_start, epsilon=1e-10, mu_prox=1e-7, max_iter=1000, adaptative_rho=True):
    dev = _dev()
    n = np.asarray(q).size
    x = ops.qcqp_forward(_t(P, (1, n, n), dev), _t(q, (1, n, 1), dev), _t(l_n, (1, n // 2, 1), dev),
                         _t(mu, (1, n // 2, 1), dev), epsilon, max_iter, mu_prox, adaptive_rho=adaptative_rho)
    return x.reshape(n).cpu().numpy()


def solveDerivativesQP(P, q, l, grad_l, epsilon=1e-10):
    """-> bl (N,): the solution of the differentiated KKT system (grad_q = -bl, grad_P = -bl l^T)."""
    dev = _dev()
    n = np.asarray(q).size
    _, gq = ops.qp_backward(_t(P, (1, n, n), dev), _t(q, (1, n, 1), dev), _t(l, (1, n, 1), dev),
                            _t(grad_l, (1, n, 1), dev), need_P=False, need_q=True, epsilon=epsilon)
    return (-gq).reshape(n).cpu().numpy()


def solveDerivativesQCQP(P, q, l_n, mu, l, grad_l, epsilon=1e-10):
    """-> (E1, E2, blgamma) like pybindings.cpp:62-71: E1/E2 dense (nc,nc) diagonal matrices,
    blgamma = [dgamma (nc); dl (N)]."""
    dev = _dev()
    n = np.asarray(q).size
    nc = n // 2
    ln_t, mu_t = _t(l_n, (1, nc, 1), dev), _t(mu, (1, nc, 1), dev)
    gam = torch.empty((1, nc, 1), dtype=torch.float64, device=dev)
    dgam = torch.empty_like(gam)
    _, gq, _, _ = ops.qcqp_backward(_t(P, (1, n, n), dev), _t(q, (1, n, 1), dev), ln_t, mu_t, _t(l, (1, n, 1), dev),
                                    _t(grad_l, (1, n, 1), dev), need=(False, True, False, False), epsilon=epsilon,
                                    duals=(gam, dgam))
    g, ln1, mu1 = gam.reshape(nc).cpu().numpy(), ln_t.reshape(nc).cpu().numpy(), mu_t.reshape(nc).cpu().numpy()
    E1 = np.diag(2 * g * ln1 * ln1 * mu1)   # getE12QCQP, Solver.cpp:683-691
    E2 = np.diag(2 * g * ln1 * mu1 * mu1)
    blgamma = np.concatenate([dgam.reshape(nc).cpu().numpy(), (-gq).reshape(n).cpu().numpy()])
    return E1, E2, blgamma


def solveBoxQP(P, q, l_min, l_max, warm_start, epsilon=1e-10, mu_prox=1e-7, max_iter=1000, adaptative_rho=True):
    """pybindings.cpp:32-37, :77"""
    dev = _dev()
    n = np.asarray(q).size
    x = ops.boxqp_forward(_t(P, (1, n, n), dev), _t(q, (1, n, 1), dev), _t(l_min, (1, n, 1), dev),
                          _t(l_max, (1, n, 1), dev), epsilon, max_iter, mu_prox=mu_prox, adaptive_rho=adaptative_rho)
    return x.reshape(n).cpu().numpy()


def solveSignedBoxQP(P, q, l_min, l_max, v, warm_start, epsilon=1e-10, mu_prox=1e-7, max_iter=1000,
                     adaptative_rho=True):
    """pybindings.cpp:47-52, :78"""
    dev = _dev()
    n = np.asarray(q).size
    x = ops.boxqp_forward(_t(P, (1, n, n), dev), _t(q, (1, n, 1), dev), _t(l_min, (1, n, 1), dev),
                          _t(l_max, (1, n, 1), dev), epsilon, max_iter, v=_t(v, (1, n, 1), dev), mu_prox=mu_prox,
                          adaptive_rho=adaptative_rho)
    return x.reshape(n).cpu().numpy()


def solveDerivativesBoxQP(P, q, l_min, l_max, l, grad_l, epsilon=1e-10):
    """-> (blgamma (3N,), gamma (2N,)) like pybindings.cpp:39-45: blgamma = [dgamma_lower (N); dgamma_upper (N);
    dl (N)], gamma = [lower multipliers (N); upper multipliers (N)]."""
    dev = _dev()
    n = np.asarray(q).size
    gam = torch.empty((1, 2 * n), dtype=torch.float64, device=dev)
    dgam = torch.empty_like(gam)
    _, gq, _, _ = ops.boxqp_backward(_t(P, (1, n, n), dev), _t(q, (1, n, 1), dev), _t(l_min, (1, n, 1), dev),
                                     _t(l_max, (1, n, 1), dev), _t(l, (1, n, 1), dev), _t(grad_l, (1, n, 1), dev),
                                     need=(False, True, False, False), epsilon=epsilon, duals=(gam, dgam))
    blgamma = np.concatenate([dgam.reshape(2 * n).cpu().numpy(), (-gq).reshape(n).cpu().numpy()])
    return blgamma, gam.reshape(2 * n).cpu().numpy()
