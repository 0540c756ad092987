import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
