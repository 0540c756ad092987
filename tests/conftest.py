import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- kernel-selection knobs in the tests -----------------------------------------------------------------------------------
# The shipped library has none (csrc/tuning.h: compile-time constants; dqq_set_option knows the three route counters only).
# Tests that drive the alternative kernels -- other lane layouts, the queued instead of the fused fallback, ... -- need the
# DEVELOPER build (python -m diffqcqp_amd.build --tuning) and SKIP on the shipped one -- and run a second time, bound to that
# build, from tests/test_gpu_developer_build.py: one `pytest -m gpu` covers both.  The two former knobs that change NUMERICS ("dense_wave64", "wave_qcqp_bwd": the
# reference-order kernels for 16 < N <= 64) are a per-call flag of the C ABI now (DQQ_F_REFERENCE_ORDER): `knob` keeps a
# test-side switch for them and `OpsWithFlags` (the `ops` fixture of test_gpu_parity.py) ORs the flag into every call's layout.
KNOB_DEFAULTS = {"fwd_lpp": 0, "wpb": 0, "fuse_fallback": -1, "fwd_respread": 16, "fwd_respread2": 8, "fwd_respread2_from": 48,
                 "lane_dense": 1, "lane_defer": 0, "dense_teams": 1, "small_fwd": 1, "small_bwd": 1, "lane_bwd": 1,
                 "fwd_feedback": 1, "bwd_skip_classify": 1}
COUNTERS = ("lane_list_drains", "bwd_whole_batches", "fwd_feedback_routes")
_ref_order = {"dense_wave64": 1, "wave_qcqp_bwd": 1}


def reference_order_flag():
    """F_REFERENCE_ORDER if a test has switched either of the two numerics switches off, else 0."""
    from diffqcqp_amd import _capi
    return _capi.F_REFERENCE_ORDER if (_ref_order["dense_wave64"] == 0 or _ref_order["wave_qcqp_bwd"] == 0) else 0


def knob(name, value):
    from diffqcqp_amd import _capi
    if name in _ref_order:
        _ref_order[name] = int(value)
        return
    if name in COUNTERS:
        return _capi.set_option(name, value)
    assert name in KNOB_DEFAULTS, name
    if _capi.tuning_build():
        return _capi.set_option(name, value)
    if int(value) != KNOB_DEFAULTS[name]:
        pytest.skip("knob %s: needs the developer build of the library (-DDQQ_TUNING, csrc/tuning.h)" % name)


class OpsWithFlags:
    """diffqcqp_amd.ops with the test-side reference-order switch ORed into every call's `layout`."""
    _WRAPPED = ("qp_forward", "qcqp_forward", "boxqp_forward", "qp_backward", "qcqp_backward", "boxqp_backward")

    def __init__(self, ops):
        self._ops = ops

    def __getattr__(self, name):
        fn = getattr(self._ops, name)
        if name not in self._WRAPPED:
            return fn

        def call(*a, **kw):
            kw["layout"] = kw.get("layout", 0) | reference_order_flag()
            return fn(*a, **kw)
        return call


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with oracle/Makefile."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


def make_problem(kind, B, N, seed, structure="diag", p_lo=0.1, dtype=np.float64):
    """Seeded synthetic batch in the reference's layouts (SURVEY.md 8d distributions).
    kind: 'qp', 'qcqp', 'box' (+ l_min, l_max) or 'sbox' (+ l_min, l_max, v).
    structure: 'diag' P = diag(U(p_lo, p_lo+1)); 'dense' P = S S^T/N + 0.1 I; 'mixed' alternates."""
    import torch
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(B, N, generator=g, dtype=torch.float64) + p_lo
    Pdiag = torch.diag_embed(p)
    S = torch.rand(B, N, N, generator=g, dtype=torch.float64)
    Pdense = torch.bmm(S, S.transpose(1, 2)) / N + 0.1 * torch.eye(N, dtype=torch.float64)
    if structure == "diag":
        P = Pdiag
    elif structure == "dense":
        P = Pdense
    else:
        sel = (torch.arange(B) % 3 == 1).view(B, 1, 1)
        P = torch.where(sel, Pdense, Pdiag)
    q = 2 * torch.rand(B, N, 1, generator=g, dtype=torch.float64) - 1
    out = {"P": P.contiguous(), "q": q, "grad_x": torch.randn(B, N, 1, generator=g, dtype=torch.float64)}
    if kind == "qcqp":
        out["l_n"] = torch.rand(B, N // 2, 1, generator=g, dtype=torch.float64)
        out["mu"] = torch.rand(B, N // 2, 1, generator=g, dtype=torch.float64)
    if kind in ("box", "sbox"):
        # the reference's own box test, Solver.cpp:807-808: l_min in [-1.5,-0.5], l_max in [0.5,1.5]; here
        # scaled by 0.6 so that, with q ~ U(-1,1), bounds are active on roughly half of the coordinates
        out["l_min"] = -0.6 * (torch.rand(B, N, 1, generator=g, dtype=torch.float64) + 0.5)
        out["l_max"] = 0.6 * (torch.rand(B, N, 1, generator=g, dtype=torch.float64) + 0.5)
        if kind == "sbox":
            out["v"] = 2 * torch.rand(B, N, 1, generator=g, dtype=torch.float64) - 1  # Solver.cpp:796
    return out
