"""ctypes binding of oracle/liboracle.so -- the CPU restatement of the reference.

TEST INFRASTRUCTURE ONLY (see the header of diffqcqp_oracle.c): imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package `diffqcqp_amd`.  PARITY UNPINNED: the reference cannot be built
or imported in this image; see oracle/README.md for what pins this restatement.

The module-level functions carry the names, argument order and defaults of the
reference's pybind11 module (`/root/reference/pybindings.cpp:74-83`); the
`*_batch` functions restate the batch loops of `/root/reference/qcqp.py`.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile liboracle.so with the committed Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "diffqcqp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_max_threads.restype = ctypes.c_int
        for name in ("orc_solveQP", "orc_solveQCQP", "orc_solveDerivativesQP", "orc_solveDerivativesQCQP",
                     "orc_solveBoxQP", "orc_solveSignedBoxQP", "orc_solveDerivativesBoxQP"):
            getattr(_lib, name).restype = ctypes.c_int
        for name in ("orc_qp_fwd_batch", "orc_qcqp_fwd_batch", "orc_qp_bwd_batch", "orc_qcqp_bwd_batch",
                     "orc_boxqp_fwd_batch", "orc_boxqp_bwd_batch"):
            getattr(_lib, name).restype = None
    return _lib


def _c(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return None if a is None else a.ctypes.data_as(_D)


def _ip(a):
    return None if a is None else a.ctypes.data_as(_I)


# ---- single-problem API: names/kwargs of pybindings.cpp:76-82 -----------------

def solveQP(P, q, warm_start, epsilon=1e-10, mu_prox=1e-7, max_iter=1000, adaptative_rho=True,
            return_iters=False):
    P, q = _c(P), _c(q).reshape(-1)
    n = q.size
    x = np.empty(n)
    it = lib().orc_solveQP(_p(P), _p(q), None, ctypes.c_int(n), ctypes.c_double(epsilon),
                           ctypes.c_double(mu_prox), ctypes.c_int(int(max_iter)), ctypes.c_int(bool(adaptative_rho)),
                           _p(x))
    return (x, it) if return_iters else x


def solveQCQP(P, q, l_n, mu, warm_start, epsilon=1e-10, mu_prox=1e-7, max_iter=1000, adaptative_rho=True,
              return_iters=False):
    P, q, l_n, mu = _c(P), _c(q).reshape(-1), _c(l_n).reshape(-1), _c(mu).reshape(-1)
    n = q.size
    x = np.empty(n)
    it = lib().orc_solveQCQP(_p(P), _p(q), _p(l_n), _p(mu), None, ctypes.c_int(n), ctypes.c_double(epsilon),
                             ctypes.c_double(mu_prox), ctypes.c_int(int(max_iter)),
                             ctypes.c_int(bool(adaptative_rho)), _p(x))
    return (x, it) if return_iters else x


def solveDerivativesQP(P, q, l, grad_l, epsilon=1e-10, return_steps=False):
    P, q, l, grad_l = _c(P), _c(q).reshape(-1), _c(l).reshape(-1), _c(grad_l).reshape(-1)
    n = q.size
    bl = np.empty(n)
    st = lib().orc_solveDerivativesQP(_p(P), _p(q), _p(l), _p(grad_l), ctypes.c_int(n), ctypes.c_double(epsilon),
                                      _p(bl))
    return (bl, st) if return_steps else bl


def solveDerivativesQCQP(P, q, l_n, mu, l, grad_l, epsilon=1e-10, return_steps=False):
    """Returns (E1 (nc,nc), E2 (nc,nc), blgamma (nc+n,)) like pybindings.cpp:62-71."""
    P, q, l_n, mu = _c(P), _c(q).reshape(-1), _c(l_n).reshape(-1), _c(mu).reshape(-1)
    l, grad_l = _c(l).reshape(-1), _c(grad_l).reshape(-1)
    n = q.size
    nc = n // 2
    e1, e2, blg, gam = np.empty(nc), np.empty(nc), np.empty(nc + n), np.empty(nc)
    st = lib().orc_solveDerivativesQCQP(_p(P), _p(q), _p(l_n), _p(mu), _p(l), _p(grad_l), ctypes.c_int(n),
                                        ctypes.c_double(epsilon), _p(e1), _p(e2), _p(blg), _p(gam))
    out = (np.diag(e1), np.diag(e2), blg)
    return out + (st, gam) if return_steps else out


def solveBoxQP(P, q, l_min, l_max, warm_start, epsilon=1e-10, mu_prox=1e-7, max_iter=1000, adaptative_rho=True,
               return_iters=False):
    """pybindings.cpp:32-37, :77"""
    P, q, l_min, l_max = _c(P), _c(q).reshape(-1), _c(l_min).reshape(-1), _c(l_max).reshape(-1)
    n = q.size
    x = np.empty(n)
    it = lib().orc_solveBoxQP(_p(P), _p(q), _p(l_min), _p(l_max), None, ctypes.c_int(n), ctypes.c_double(epsilon),
                              ctypes.c_double(mu_prox), ctypes.c_int(int(max_iter)),
                              ctypes.c_int(bool(adaptative_rho)), _p(x))
    return (x, it) if return_iters else x


def solveSignedBoxQP(P, q, l_min, l_max, v, warm_start, epsilon=1e-10, mu_prox=1e-7, max_iter=1000,
                     adaptative_rho=True, return_iters=False):
    """pybindings.cpp:47-52, :78"""
    P, q, l_min, l_max, v = _c(P), _c(q).reshape(-1), _c(l_min).reshape(-1), _c(l_max).reshape(-1), _c(v).reshape(-1)
    n = q.size
    x = np.empty(n)
    it = lib().orc_solveSignedBoxQP(_p(P), _p(q), _p(l_min), _p(l_max), _p(v), None, ctypes.c_int(n),
                                    ctypes.c_double(epsilon), ctypes.c_double(mu_prox), ctypes.c_int(int(max_iter)),
                                    ctypes.c_int(bool(adaptative_rho)), _p(x))
    return (x, it) if return_iters else x


def solveDerivativesBoxQP(P, q, l_min, l_max, l, grad_l, epsilon=1e-10, return_steps=False):
    """Returns (blgamma (3n,), gamma (2n,)) like pybindings.cpp:39-45, :81."""
    P, q, l_min, l_max = _c(P), _c(q).reshape(-1), _c(l_min).reshape(-1), _c(l_max).reshape(-1)
    l, grad_l = _c(l).reshape(-1), _c(grad_l).reshape(-1)
    n = q.size
    blg, gam = np.empty(3 * n), np.empty(2 * n)
    st = np.zeros(2, dtype=np.int32)
    lib().orc_solveDerivativesBoxQP(_p(P), _p(q), _p(l_min), _p(l_max), _p(l), _p(grad_l), ctypes.c_int(n),
                                    ctypes.c_double(epsilon), _p(blg), _p(gam), _ip(st))
    return (blg, gam, st) if return_steps else (blg, gam)


# ---- batched API: the loops of qcqp.py:24-52, 144-181 ---------------------------

def set_force_ir_steps(steps):
    """Test knob: > 0 = iterative_refinement runs exactly `steps` loop bodies (0 restores the reference's exits)."""
    lib().orc_set_force_ir_steps(ctypes.c_int(int(steps)))


def max_threads():
    return lib().orc_max_threads()


def qp_fwd_batch(P, q, eps, max_iter, mu_prox=1e-7, nthreads=1):
    P, q = _c(P), _c(q)
    B, n = q.shape[0], q.shape[1]
    x = np.empty((B, n, 1))
    iters = np.empty(B, dtype=np.int32)
    lib().orc_qp_fwd_batch(_p(P), _p(q), ctypes.c_long(B), ctypes.c_int(n), ctypes.c_double(eps),
                           ctypes.c_double(mu_prox), ctypes.c_int(int(max_iter)), _p(x), _ip(iters),
                           ctypes.c_int(nthreads))
    return x, iters


def qcqp_fwd_batch(P, q, l_n, mu, eps, max_iter, mu_prox=1e-7, nthreads=1):
    P, q, l_n, mu = _c(P), _c(q), _c(l_n), _c(mu)
    B, n = q.shape[0], q.shape[1]
    x = np.empty((B, n, 1))
    iters = np.empty(B, dtype=np.int32)
    lib().orc_qcqp_fwd_batch(_p(P), _p(q), _p(l_n), _p(mu), ctypes.c_long(B), ctypes.c_int(n),
                             ctypes.c_double(eps), ctypes.c_double(mu_prox), ctypes.c_int(int(max_iter)), _p(x),
                             _ip(iters), ctypes.c_int(nthreads))
    return x, iters


def qp_bwd_batch(P, q, x, grad_x, nthreads=1):
    """-> grad_P (B,n,n), grad_q (B,n,1), ir_steps (B,)"""
    P, q, x, grad_x = _c(P), _c(q), _c(x), _c(grad_x)
    B, n = q.shape[0], q.shape[1]
    gP, gq = np.empty((B, n, n)), np.empty((B, n, 1))
    steps = np.empty(B, dtype=np.int32)
    lib().orc_qp_bwd_batch(_p(P), _p(q), _p(x), _p(grad_x), ctypes.c_long(B), ctypes.c_int(n), _p(gP), _p(gq),
                           _ip(steps), ctypes.c_int(nthreads))
    return gP, gq, steps


def qcqp_bwd_batch(P, q, l_n, mu, x, grad_x, nthreads=1):
    """-> grad_P, grad_q, grad_l_n (B,nc,1), grad_mu (B,nc,1), ir_steps"""
    P, q, l_n, mu, x, grad_x = _c(P), _c(q), _c(l_n), _c(mu), _c(x), _c(grad_x)
    B, n = q.shape[0], q.shape[1]
    nc = n // 2
    gP, gq = np.empty((B, n, n)), np.empty((B, n, 1))
    gl, gm = np.empty((B, nc, 1)), np.empty((B, nc, 1))
    steps = np.empty(B, dtype=np.int32)
    lib().orc_qcqp_bwd_batch(_p(P), _p(q), _p(l_n), _p(mu), _p(x), _p(grad_x), ctypes.c_long(B), ctypes.c_int(n),
                             _p(gP), _p(gq), _p(gl), _p(gm), _ip(steps), ctypes.c_int(nthreads))
    return gP, gq, gl, gm, steps


def boxqp_fwd_batch(P, q, l_min, l_max, eps, max_iter, v=None, mu_prox=1e-7, nthreads=1):
    """BoxQPFn2.forward (qcqp.py:56-65); with v: SignedBoxQPFn2.forward (qcqp.py:99-108)."""
    P, q, l_min, l_max = _c(P), _c(q), _c(l_min), _c(l_max)
    v = None if v is None else _c(v)
    B, n = q.shape[0], q.shape[1]
    x = np.empty((B, n, 1))
    iters = np.empty(B, dtype=np.int32)
    lib().orc_boxqp_fwd_batch(_p(P), _p(q), _p(l_min), _p(l_max), _p(v), ctypes.c_long(B), ctypes.c_int(n),
                              ctypes.c_double(eps), ctypes.c_double(mu_prox), ctypes.c_int(int(max_iter)), _p(x),
                              _ip(iters), ctypes.c_int(nthreads))
    return x, iters


def boxqp_bwd_batch(P, q, l_min, l_max, x, grad_x, nthreads=1):
    """-> grad_P, grad_q, grad_l_min (B,n,1), grad_l_max (B,n,1), gamma (B,2n), ir_steps (B,2)"""
    P, q, l_min, l_max, x, grad_x = _c(P), _c(q), _c(l_min), _c(l_max), _c(x), _c(grad_x)
    B, n = q.shape[0], q.shape[1]
    gP, gq = np.empty((B, n, n)), np.empty((B, n, 1))
    glo, ghi = np.empty((B, n, 1)), np.empty((B, n, 1))
    gam = np.empty((B, 2 * n))
    steps = np.empty((B, 2), dtype=np.int32)
    lib().orc_boxqp_bwd_batch(_p(P), _p(q), _p(l_min), _p(l_max), _p(x), _p(grad_x), ctypes.c_long(B),
                              ctypes.c_int(n), _p(gP), _p(gq), _p(glo), _p(ghi), _p(gam), _ip(steps),
                              ctypes.c_int(nthreads))
    return gP, gq, glo, ghi, gam, steps
