import sys, os, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
def t(fn,n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b)*1e3/n
for st in ("diag","dense"):
  for B in (256, 4096, 16384, 32768, 65536):
    d={k:v.cuda() for k,v in make_problem("qcqp",B,8,7,structure=st).items()}
    g=torch.randn(B,8,1,dtype=torch.float64,device="cuda")
    cq, cc = ops.diag_cache(d["q"]), ops.diag_cache(d["q"])
    xq=ops.qp_forward(d["P"],d["q"],1e-7,1000,cache=cq); xc=ops.qcqp_forward(d["P"],d["q"],d["l_n"],d["mu"],1e-7,1000,cache=cc)
    row=[]
    for fo in (-1,1,0):
        _capi.set_option("fuse_fallback",fo)
        row.append("fuse %2d: qp %.0f qcqp %.0f"%(fo,t(lambda: ops.qp_backward(d["P"],d["q"],xq,g,cache=cq)), t(lambda: ops.qcqp_backward(d["P"],d["q"],d["l_n"],d["mu"],xc,g,cache=cc))))
    _capi.set_option("fuse_fallback",-1)
    print(st,B," | ".join(row))
# DQQ_P_DENSE reference point and a 1-in-1000 dense batch at the bench's batch size
B = 65536
d = {k: v.cuda() for k, v in make_problem("qcqp", B, 8, 7, structure="dense").items()}
g = torch.randn(B, 8, 1, dtype=torch.float64, device="cuda")
xq = ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=1)
xc = ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=1)
print("dense 65536 DQQ_P_DENSE: qp fwd %.0f bwd %.0f | qcqp fwd %.0f bwd %.0f" % (
    t(lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000, layout=1)), t(lambda: ops.qp_backward(d["P"], d["q"], xq, g, layout=1)),
    t(lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=1)),
    t(lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], xc, g, layout=1))))
print("dense 65536 DQQ_P_AUTO : qp fwd %.0f bwd %.0f | qcqp fwd %.0f bwd %.0f" % (
    t(lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000)), t(lambda: ops.qp_backward(d["P"], d["q"], xq, g)),
    t(lambda: ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000)),
    t(lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], xc, g))))
dd = {k: v.cuda() for k, v in make_problem("qcqp", B, 8, 7, structure="diag").items()}
sel = (torch.arange(B, device="cuda") % 1000 == 1).view(B, 1, 1)
Pm = torch.where(sel, d["P"], dd["P"]).contiguous()
xq = ops.qp_forward(Pm, dd["q"], 1e-7, 1000)
xc = ops.qcqp_forward(Pm, dd["q"], dd["l_n"], dd["mu"], 1e-7, 1000)
for fo in (-1, 1):
    _capi.set_option("fuse_fallback", fo)
    print("1-in-1000 dense 65536 AUTO fuse %2d: qp fwd %.0f bwd %.0f | qcqp fwd %.0f bwd %.0f" % (
        fo, t(lambda: ops.qp_forward(Pm, dd["q"], 1e-7, 1000)), t(lambda: ops.qp_backward(Pm, dd["q"], xq, g)),
        t(lambda: ops.qcqp_forward(Pm, dd["q"], dd["l_n"], dd["mu"], 1e-7, 1000)),
        t(lambda: ops.qcqp_backward(Pm, dd["q"], dd["l_n"], dd["mu"], xc, g))))
_capi.set_option("fuse_fallback", -1)
