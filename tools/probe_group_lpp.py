import sys, os, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b)*1e3/n
for B in (4096, 32768, 65536, 131072):
    d={k:v.cuda() for k,v in make_problem("qcqp",B,8,7,structure="dense").items()}
    x=torch.empty(B,8,1,dtype=torch.float64,device="cuda")
    row=[]
    for lpp in (0,2,4):
        _capi.set_option("fwd_lpp", lpp); _capi.set_option("fuse_fallback", 1)
        row.append("lpp %d: qp %.0f qcqp %.0f"%(lpp,t(lambda: ops.qp_forward(d["P"],d["q"],1e-7,1000,layout=0,out=x)), t(lambda: ops.qcqp_forward(d["P"],d["q"],d["l_n"],d["mu"],1e-7,1000,layout=0,out=x))))
    print(B," | ".join(row))
