"""CPU check of the per-problem math templates the HIP kernels instantiate
(diffqcqp_amd/csrc/admm_core.h, kkt_core.h), compiled for the host with one lane
per problem (tests/hostcore/host_core_check.cpp) and compared with the oracle.
The host build is a test artefact; the product never calls it."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import make_problem

HERE = os.path.dirname(os.path.abspath(__file__))
D = ctypes.POINTER(ctypes.c_double)


def _p(a):
    return a.ctypes.data_as(D)


@pytest.fixture(scope="module")
def hostcore():
    src = os.path.join(HERE, "hostcore", "host_core_check.cpp")
    so = os.path.join(HERE, "hostcore", "libhostcore.so")
    deps = [src] + [os.path.join(HERE, "..", "diffqcqp_amd", "csrc", f) for f in ("admm_core.h", "kkt_core.h", "common.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                               "-fvisibility=hidden", "-o", so, src])
    return ctypes.CDLL(so)


@pytest.mark.parametrize("kind,N,p_lo", [("qp", 8, 0.1), ("qp", 8, 0.0), ("qp", 16, 0.1), ("qp", 2, 0.1),
                                         ("qcqp", 8, 0.1), ("qcqp", 4, 0.1), ("qcqp", 16, 0.1)])
def test_forward_core_follows_oracle_trajectory(oracle, hostcore, kind, N, p_lo):
    B = 300
    d = make_problem(kind, B, N, 61 + N, p_lo=p_lo)
    P, q = d["P"].numpy(), d["q"].numpy()
    p = np.ascontiguousarray(np.diagonal(P, axis1=1, axis2=2))
    if kind == "qp":
        xo, ito = oracle.qp_fwd_batch(P, q, 1e-7, 1000)
        rad = np.zeros((B, N // 2))
    else:
        xo, ito = oracle.qcqp_fwd_batch(P, q, d["l_n"].numpy(), d["mu"].numpy(), 1e-7, 1000)
        rad = np.ascontiguousarray((d["l_n"] * d["mu"]).numpy()[:, :, 0])
    xh, ith = np.zeros((B, N)), np.zeros(B, dtype=int)
    qq = np.ascontiguousarray(q[:, :, 0])
    for b in range(B):
        ith[b] = hostcore.hostcore_fwd(0 if kind == "qp" else 1, N, _p(p[b]), _p(qq[b]), _p(rad[b]),
                                       ctypes.c_double(1e-7), ctypes.c_double(1e-7), 1000, 1, _p(xh[b]))
    assert np.array_equal(ith, ito), "rho schedule / stopping iteration must match the reference algorithm"
    assert (np.abs(xh - xo[:, :, 0]) / np.maximum(1.0, np.abs(xo[:, :, 0]))).max() < 1e-11


def test_backward_cores_are_bit_exact(oracle, hostcore):
    B, N = 400, 8
    d = make_problem("qcqp", B, N, 71)
    P, q, g = d["P"].numpy(), d["q"].numpy(), d["grad_x"].numpy()
    ln, mu = d["l_n"].numpy(), d["mu"].numpy()
    p = np.ascontiguousarray(np.diagonal(P, axis1=1, axis2=2))
    qq, gg = np.ascontiguousarray(q[:, :, 0]), np.ascontiguousarray(g[:, :, 0])
    # QP
    x, _ = oracle.qp_fwd_batch(P, q, 1e-7, 1000)
    gP, gq, st = oracle.qp_bwd_batch(P, q, x, g)
    xx = np.ascontiguousarray(x[:, :, 0])
    dl, sth = np.zeros((B, N)), np.zeros(B, dtype=int)
    for b in range(B):
        sth[b] = hostcore.hostcore_qp_bwd(N, _p(p[b]), _p(qq[b]), _p(xx[b]), _p(gg[b]), _p(dl[b]))
    assert np.array_equal(-dl, gq[:, :, 0]) and np.array_equal(sth, st)
    # QCQP (1-step and 3-step refinement exits both occur)
    x, _ = oracle.qcqp_fwd_batch(P, q, ln, mu, 1e-7, 1000)
    gP, gq, gl, gm, st = oracle.qcqp_bwd_batch(P, q, ln, mu, x, g)
    assert set(np.unique(st)) >= {1, 3}
    xx = np.ascontiguousarray(x[:, :, 0])
    l1, m1 = np.ascontiguousarray(ln[:, :, 0]), np.ascontiguousarray(mu[:, :, 0])
    gl2, gm2 = np.zeros((B, N // 2)), np.zeros((B, N // 2))
    for b in range(B):
        sth[b] = hostcore.hostcore_qcqp_bwd(N, _p(p[b]), _p(qq[b]), _p(l1[b]), _p(m1[b]), _p(xx[b]), _p(gg[b]),
                                            _p(dl[b]), _p(gl2[b]), _p(gm2[b]))
    assert np.array_equal(sth, st)
    assert np.array_equal(-dl, gq[:, :, 0])
    assert np.array_equal(gl2, gl[:, :, 0]) and np.array_equal(gm2, gm[:, :, 0])


@pytest.mark.parametrize("kind,N", [("box", 8), ("box", 2), ("box", 16), ("sbox", 8), ("sbox", 4)])
def test_box_forward_core_follows_oracle_trajectory(oracle, hostcore, kind, N):
    B = 300
    d = make_problem(kind, B, N, 161 + N)
    P, q = d["P"].numpy(), d["q"].numpy()
    lo, hi = d["l_min"].numpy(), d["l_max"].numpy()
    v = d["v"].numpy() if kind == "sbox" else None
    xo, ito = oracle.boxqp_fwd_batch(P, q, lo, hi, 1e-7, 1000, v=v)
    p = np.ascontiguousarray(np.diagonal(P, axis1=1, axis2=2))
    qq, l1, h1 = (np.ascontiguousarray(a[:, :, 0]) for a in (q, lo, hi))
    v1 = None if v is None else np.ascontiguousarray(v[:, :, 0])
    xh, ith = np.zeros((B, N)), np.zeros(B, dtype=int)
    for b in range(B):
        ith[b] = hostcore.hostcore_box_fwd(N, _p(p[b]), _p(qq[b]), _p(l1[b]), _p(h1[b]), None if v1 is None else _p(v1[b]),
                                           ctypes.c_double(1e-7), ctypes.c_double(1e-7), 1000, 1, _p(xh[b]))
    assert np.array_equal(ith, ito)
    assert (np.abs(xh - xo[:, :, 0]) / np.maximum(1.0, np.abs(xo[:, :, 0]))).max() < 1e-11


def test_box_backward_core_is_bit_exact(oracle, hostcore):
    """Per-coordinate blocks (1x1 / 2x2 dual, up to 3x3 derivative system, incl. pinned coordinates with both
    multipliers) against the oracle's dense (3N)^2 solve: identical bits, identical refinement exits."""
    B, N = 400, 8
    d = make_problem("box", B, N, 171)
    P, q, g = d["P"].numpy(), d["q"].numpy(), d["grad_x"].numpy()
    lo, hi = d["l_min"].numpy().copy(), d["l_max"].numpy().copy()
    x, _ = oracle.boxqp_fwd_batch(P, q, lo, hi, 1e-7, 1000)
    hi[::5, 2, 0] = lo[::5, 2, 0]          # pinned coordinates: both bounds active
    x[::5, 2, 0] = lo[::5, 2, 0]
    gP, gq, glo, ghi, gam, st = oracle.boxqp_bwd_batch(P, q, lo, hi, x, g)
    p = np.ascontiguousarray(np.diagonal(P, axis1=1, axis2=2))
    qq, gg, xx, l1, h1 = (np.ascontiguousarray(a[:, :, 0]) for a in (q, g, x, lo, hi))
    dl, gm, dg = np.zeros((B, N)), np.zeros((B, 2 * N)), np.zeros((B, 2 * N))
    sth = np.zeros((B, 2), dtype=np.int32)
    for b in range(B):
        hostcore.hostcore_box_bwd(N, _p(p[b]), _p(qq[b]), _p(l1[b]), _p(h1[b]), _p(xx[b]), _p(gg[b]), _p(dl[b]),
                                  _p(gm[b]), _p(dg[b]), sth[b].ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    assert np.array_equal(sth, st)
    assert np.array_equal(-dl, gq[:, :, 0]) and np.array_equal(gm, gam)
    assert np.array_equal(-(dg[:, :N] * gm[:, :N]), glo[:, :, 0]) and np.array_equal(dg[:, N:] * gm[:, N:], ghi[:, :, 0])
