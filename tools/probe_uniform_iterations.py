"""Upper bound on what perfect lane utilisation could buy the N = 8 forward: the same batch solved with eps = 0
and max_iter = the natural MEAN iteration count -- every problem then runs exactly that many iterations, a wave's lanes are
never idle, and the total iteration work equals the natural run's.  usage: python tools/probe_uniform_iterations.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffqcqp_amd import ops

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps)
    return sorted(ts)[1] * 1e6

for B in (65536, 262144, 1048576):
    g = torch.Generator(device="cuda").manual_seed(1002)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64, device="cuda")
    P = torch.diag_embed(r(B, 8) + 0.1).contiguous(); q = 2 * r(B, 8, 1) - 1
    ln, mu = r(B, 4, 1), r(B, 4, 1)
    x = torch.empty(B, 8, 1, dtype=torch.float64, device="cuda")
    for kind in ("qp", "qcqp"):
        f = (lambda e, m: ops.qp_forward(P, q, e, m, out=x)) if kind == "qp" else (lambda e, m: ops.qcqp_forward(P, q, ln, mu, e, m, out=x))
        fi = (lambda e, m: ops.qp_forward(P, q, e, m, return_iters=True)[1]) if kind == "qp" else (lambda e, m: ops.qcqp_forward(P, q, ln, mu, e, m, return_iters=True)[1])
        it = fi(1e-7, 1000).double()
        mean = float(it.mean()); tmax = float(it.view(-1, 32).max(dim=1).values.mean())
        t_nat = timed(lambda: f(1e-7, 1000))
        m = int(round(mean))
        t_uni = timed(lambda: f(0.0, m))
        t_max = timed(lambda: f(0.0, int(round(tmax))))
        print("B %8d %-5s natural %7.1f us (mean its %.2f, tile-max mean %.2f) | every problem exactly %d its: %7.1f us (%.2f of natural) | exactly %d its: %7.1f us"
              % (B, kind, t_nat, mean, tmax, m, t_uni, t_uni / t_nat, int(round(tmax)), t_max), flush=True)
