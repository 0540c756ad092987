import sys, os, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b)*1e3/n
for B in (64, 512, 4096, 16384, 32768, 65536, 131072):
    d={k:v.cuda() for k,v in make_problem("qcqp",B,8,7,structure="dense").items()}
    x=torch.empty(B,8,1,dtype=torch.float64,device="cuda")
    row=[]
    for nm,lay,ld in (("DENSE(lane kernel)",1,0),("DENSE(routed)",1,1),("AUTO",0,1)):
        _capi.set_option("lane_dense", 1)
        # lane_dense=0 would also change the dense kernel; emulate 'unrouted' through fuse_fallback=0
        _capi.set_option("fuse_fallback", 0 if nm.startswith("DENSE(lane") else -1)
        row.append("%s qp %.0f qcqp %.0f"%(nm,t(lambda: ops.qp_forward(d["P"],d["q"],1e-7,1000,layout=lay,out=x)), t(lambda: ops.qcqp_forward(d["P"],d["q"],d["l_n"],d["mu"],1e-7,1000,layout=lay,out=x))))
    print(B," | ".join(row))
