"""INTEGRATION.md, Option B, executed as written (-m gpu): the ctypes patch a reference maintainer would paste into qcqp.py is
extracted from the document, pointed at the built library and run -- its QPFn2 must return what the oracle returns (and the
very bits of diffqcqp_amd.qcqp.QPFn2, which makes the same two C-ABI calls).  A signature change that the document does not
follow fails here."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import make_problem

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _snippet():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = [b for b in blocks if "class QPFn2(Function)" in b]
    assert len(code) == 1, "the Option B patch is one python block"
    return code[0]


def test_option_b_patch_runs_as_documented(oracle):
    from diffqcqp_amd import build, _capi
    from diffqcqp_amd.qcqp import QPFn2 as Product
    build.build()
    src = _snippet().replace('ctypes.CDLL("libdiffqcqp_hip.so")', "ctypes.CDLL(%r)" % _capi.LIB_PATH)
    ns = {"Function": torch.autograd.Function}
    exec(compile(src, "INTEGRATION.md:option-b", "exec"), ns)
    Patched = ns["QPFn2"]
    for structure in ("diag", "dense"):
        d = make_problem("qp", 700, 8, 6100, structure)
        outs = []
        for F in (Patched, Product):
            P = d["P"].cuda().requires_grad_(True)
            q = d["q"].cuda().requires_grad_(True)
            x = F.apply(P, q, torch.zeros_like(q), 1e-7, 1000)
            (x * d["grad_x"].cuda()).sum().backward()
            outs.append((x.detach().cpu().numpy(), P.grad.cpu().numpy(), q.grad.cpu().numpy()))
        for a, b in zip(*outs):
            assert np.array_equal(a, b)                       # the same two launches: the same bits
        xo, _ = oracle.qp_fwd_batch(d["P"].numpy(), d["q"].numpy(), 1e-7, 1000)
        gP, gq, _ = oracle.qp_bwd_batch(d["P"].numpy(), d["q"].numpy(), xo, d["grad_x"].numpy())
        assert np.abs(outs[0][0] - xo).max() <= 1e-6
        assert np.abs(outs[0][2] - gq).max() <= 1e-6 * max(1.0, np.abs(gq).max())
        assert np.abs(outs[0][1] - gP).max() <= 1e-6 * max(1.0, np.abs(gP).max())
