"""Developer probe: overhead of the Python layers above the C ABI at the bench shape (B=65536, N=8)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops
from diffqcqp_amd.qcqp import QPFn2, QCQPFn2
B, N = 65536, 8
d = {k: v.cuda() for k, v in make_problem("qcqp", B, N, 1002).items()}
P = d["P"].clone().requires_grad_(True); q = d["q"].clone().requires_grad_(True)
ln = d["l_n"].clone().requires_grad_(True); mu = d["mu"].clone().requires_grad_(True)
ws = torch.zeros_like(q); g = d["grad_x"]
def wall(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
def qp_auto():
    x = QPFn2.apply(P, q, ws, 1e-7, 1000); x.backward(g)
    P.grad = None; q.grad = None
def qcqp_auto():
    x = QCQPFn2.apply(P, q, ln, mu, ws, 1e-7, 1000); x.backward(g)
    P.grad = None; q.grad = None; ln.grad = None; mu.grad = None
def qp_ops():
    x = ops.qp_forward(d["P"], d["q"], 1e-7, 1000); ops.qp_backward(d["P"], d["q"], x, g)
def qcqp_ops():
    x = ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000); ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, g)
print("QP   fwd+bwd: autograd %.1f us   ops %.1f us" % (wall(qp_auto), wall(qp_ops)))
print("QCQP fwd+bwd: autograd %.1f us   ops %.1f us" % (wall(qcqp_auto), wall(qcqp_ops)))
