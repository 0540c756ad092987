import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import _capi, ops
N, B = int(sys.argv[1]), int(sys.argv[2])
d = {k: v.cuda() for k, v in make_problem("qp", B, N, 1000 + N).items()}
pd = torch.diagonal(d["P"], dim1=1, dim2=2).contiguous()
_capi.set_option("auto_fallback", 0)
for lpp in [int(a) for a in sys.argv[3:]]:
    _capi.set_option("fwd_lpp", lpp)
    for _ in range(5):
        ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=0)
        ops.qp_forward(pd, d["q"], 1e-7, 1000, layout=2)
torch.cuda.synchronize()
