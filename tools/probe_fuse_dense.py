import sys, os, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from conftest import make_problem
from diffqcqp_amd import ops
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b)*1e3/n
for B in (4096,65536):
    d={k:v.cuda() for k,v in make_problem("qcqp",B,8,7,structure="dense").items()}
    x=torch.empty(B,8,1,dtype=torch.float64,device="cuda")
    print(B,"AUTO dense qp %.1f us  qcqp %.1f us"%(t(lambda: ops.qp_forward(d["P"],d["q"],1e-7,1000,layout=0,out=x)), t(lambda: ops.qcqp_forward(d["P"],d["q"],d["l_n"],d["mu"],1e-7,1000,layout=0,out=x))))
    # mixed: one dense problem every 1000
    P=torch.diag_embed(torch.diagonal(d["P"],dim1=1,dim2=2)).contiguous(); P[::1000]=d["P"][::1000]
    print(B,"AUTO mixed qp %.1f us  qcqp %.1f us"%(t(lambda: ops.qp_forward(P,d["q"],1e-7,1000,layout=0,out=x)), t(lambda: ops.qcqp_forward(P,d["q"],d["l_n"],d["mu"],1e-7,1000,layout=0,out=x))))
from diffqcqp_amd import _capi
for B in (4096, 65536):
    for N in (4, 8, 16):
        d={k:v.cuda() for k,v in make_problem("qcqp",B,N,7,structure="dense").items()}
        x=torch.empty(B,N,1,dtype=torch.float64,device="cuda")
        row=[]
        for name,lay,fo in (("DENSE",1,-1),("AUTO fuse",0,1),("AUTO list",0,0)):
            _capi.set_option("fuse_fallback",fo)
            row.append("%s qp %.0f qcqp %.0f"%(name,t(lambda: ops.qp_forward(d["P"],d["q"],1e-7,1000,layout=lay,out=x)), t(lambda: ops.qcqp_forward(d["P"],d["q"],d["l_n"],d["mu"],1e-7,1000,layout=lay,out=x))))
        print(B,N," | ".join(row))
print("backward")
for B in (4096, 65536):
    for N in (4, 8):
        d={k:v.cuda() for k,v in make_problem("qcqp",B,N,7,structure="dense").items()}
        g=torch.randn(B,N,1,dtype=torch.float64,device="cuda")
        xq=ops.qp_forward(d["P"],d["q"],1e-7,1000,layout=1); xc=ops.qcqp_forward(d["P"],d["q"],d["l_n"],d["mu"],1e-7,1000,layout=1)
        row=[]
        for name,lay,fo in (("DENSE",1,-1),("AUTO fuse",0,1),("AUTO list",0,0)):
            _capi.set_option("fuse_fallback",fo)
            row.append("%s qp %.0f qcqp %.0f"%(name,t(lambda: ops.qp_backward(d["P"],d["q"],xq,g,layout=lay)), t(lambda: ops.qcqp_backward(d["P"],d["q"],d["l_n"],d["mu"],xc,g,layout=lay))))
        print(B,N," | ".join(row))
