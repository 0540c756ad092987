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
