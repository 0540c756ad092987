"""rocprofv3 target: dense 8x8, B=65536 backward through AUTO and DENSE (which kernels, how long)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops
B = 65536
d = {k: v.cuda() for k, v in make_problem("qcqp", B, 8, 7, structure="dense").items()}
g = torch.randn(B, 8, 1, dtype=torch.float64, device="cuda")
cq, cc = ops.diag_cache(d["q"]), ops.diag_cache(d["q"])
xq = ops.qp_forward(d["P"], d["q"], 1e-7, 1000, cache=cq)
xc = ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, cache=cc)
for lay in (0, 1):
    for _ in range(5):
        ops.qp_backward(d["P"], d["q"], xq, g, layout=lay, cache=cq if lay == 0 else None)
        ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], xc, g, layout=lay, cache=cc if lay == 0 else None)
torch.cuda.synchronize()
