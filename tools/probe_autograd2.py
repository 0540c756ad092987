"""Developer probe: where the Python-side time of QPFn2.apply + backward goes (B=65536, N=8)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops
from diffqcqp_amd.qcqp import QPFn2
B, N = 65536, 8
d = {k: v.cuda() for k, v in make_problem("qp", B, N, 1002).items()}
P = d["P"].clone().requires_grad_(True); q = d["q"].clone().requires_grad_(True)
ws = torch.zeros_like(q); g = d["grad_x"]

class Noop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, P, q):
        ctx.save_for_backward(P, q)
        return q.detach().clone()
    @staticmethod
    def backward(ctx, gl):
        P, q = ctx.saved_tensors
        return None, gl

def wall(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6

def cpu_only(fn, n=200):
    """host time per call without waiting for the GPU (launch-side cost)"""
    for _ in range(20): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    el = (time.perf_counter() - t) / n * 1e6
    torch.cuda.synchronize()
    return el

print("noop fwd            host %.1f us" % cpu_only(lambda: Noop.apply(P, q)))
def noop_fb():
    y = Noop.apply(P, q); y.backward(g); q.grad = None
print("noop fwd+bwd        wall %.1f us" % wall(noop_fb))
print("QPFn2 fwd           host %.1f us  wall %.1f us" % (cpu_only(lambda: QPFn2.apply(P, q, ws, 1e-7, 1000)), wall(lambda: QPFn2.apply(P, q, ws, 1e-7, 1000))))
def qp_fb():
    x = QPFn2.apply(P, q, ws, 1e-7, 1000); x.backward(g); P.grad = None; q.grad = None
print("QPFn2 fwd+bwd       wall %.1f us" % wall(qp_fb))
print("ops fwd             host %.1f us" % cpu_only(lambda: ops.qp_forward(d["P"], d["q"], 1e-7, 1000)))
x = ops.qp_forward(d["P"], d["q"], 1e-7, 1000)
print("ops bwd             host %.1f us" % cpu_only(lambda: ops.qp_backward(d["P"], d["q"], x, g)))
print("torch.empty(B,N,N)  host %.1f us" % cpu_only(lambda: torch.empty((B, N, N), dtype=torch.float64, device="cuda")))
