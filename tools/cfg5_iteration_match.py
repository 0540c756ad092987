import sys, numpy as np, torch
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,"tests"))
from diffqcqp_amd import ops
from oracle import oracle as O
B,N=65536,64
gen=torch.Generator(device="cuda").manual_seed(1005)
S=torch.rand(B,N,N,generator=gen,dtype=torch.float64,device="cuda")
P=torch.bmm(S,S.transpose(1,2))/N; del S
P.diagonal(dim1=1,dim2=2).add_(0.1)
q=2*torch.rand(B,N,1,generator=gen,dtype=torch.float64,device="cuda")-1
x,it=ops.qp_forward(P,q,1e-7,1000,return_iters=True)
idx=torch.arange(0,B,64,device="cuda")
xo,ito=O.qp_fwd_batch(P[idx].cpu().numpy(),q[idx].cpu().numpy(),1e-7,1000,nthreads=64)
ith=it[idx].cpu().numpy()
print("sample",len(ito),"iteration counts equal",(ith==ito).mean(),"max |diff|",np.abs(ith.astype(int)-ito).max(),"max |dx|",np.abs(x[idx].cpu().numpy()-xo).max(), "mean its", ito.mean())
