"""Second re-spread of the N = 8 forward (fwd_respread2: the last survivors of a tile onto eight lanes per problem):
timing per threshold on the bench distribution, the stress variant p ~ U(0,1), and the reference's figure workload
P = diag(exp(U(-10,10))) at the bench's eps / max_iter and at the reference's own; bit identity against threshold 0."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from diffqcqp_amd import _capi, ops
B, N = 65536, 8
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n
g = torch.Generator(device="cuda").manual_seed(1002)
r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64, device="cuda")
q, l_n, mu = 2 * r(B, N, 1) - 1, r(B, N // 2, 1), r(B, N // 2, 1)
cases = {"bench p~U(.1,1.1)": (torch.diag_embed(r(B, N) + 0.1).contiguous(), 1e-7, 1000, 50),
         "stress p~U(0,1)": (torch.diag_embed(r(B, N)).contiguous(), 1e-7, 1000, 20),
         "figure exp(U(-10,10)) eps 1e-7/1000": (torch.diag_embed(torch.exp(r(B, N) * 20 - 10)).contiguous(), 1e-7, 1000, 10),
         "figure exp(U(-10,10)) eps 1e-10/1e6": (torch.diag_embed(torch.exp(r(B, N) * 20 - 10)).contiguous(), 1e-10, 1000000, 3)}
xo = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
for name, (P, eps, mi, reps) in cases.items():
    for kind in ("qp", "qcqp"):
        run = (lambda **kw: ops.qp_forward(P, q, eps, mi, **kw)) if kind == "qp" else (lambda **kw: ops.qcqp_forward(P, q, l_n, mu, eps, mi, **kw))
        ref, row = None, []
        for at2 in (0, 2, 4, 6, 8):
            _capi.set_option("fwd_respread2", at2)
            x, it = run(return_iters=True)
            if ref is None: ref = (x.clone(), it.clone())
            same = torch.equal(x, ref[0]) and torch.equal(it, ref[1])
            row.append("%d: %.1f%s" % (at2, t(lambda: run(out=xo), reps), "" if same else " DIFFERS"))
        print("%-38s %-5s max it %6d | us at fwd_respread2 = %s" % (name, kind, int(ref[1].max()), "  ".join(row)), flush=True)
_capi.set_option("fwd_respread2", 8)
