import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
def t(fn,n=5):
    fn(); torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b)/n
for N in (32, 40, 48, 64):
    for B in (4096, 16384):
        d={k:v.cuda() for k,v in make_problem("qcqp",B,N,7,structure="dense").items()}
        x=ops.qcqp_forward(d["P"],d["q"],d["l_n"],d["mu"],1e-7,1000,layout=1)
        tf=t(lambda: ops.qcqp_forward(d["P"],d["q"],d["l_n"],d["mu"],1e-7,1000,layout=1))
        tb=t(lambda: ops.qcqp_backward(d["P"],d["q"],d["l_n"],d["mu"],x,d["grad_x"],layout=1))
        print("QCQP dense N=%d B=%d: forward %.3f ms backward %.3f ms"%(N,B,tf,tb), flush=True)
