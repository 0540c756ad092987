"""Developer probe: N = 8 forward on two lanes per problem, tail of each tile re-spread onto four lanes per problem
(option fwd_respread = number of live problems per wave at which they move; 0 = never): timing and bit-identity."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import _capi, ops
N = 8
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n
def run(kind, d, eps, mi, **kw):
    if kind == "qp": return ops.qp_forward(d["P"], d["q"], eps, mi, **kw)
    return ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], eps, mi, **kw)
for B in (65536, 65536 - 19, 40000, 131072):
    for kind in ("qp", "qcqp"):
        for seed in (1002, 7):
            d = {k: v.cuda() for k, v in make_problem(kind, B, N, seed).items()}
            xo = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
            ref = None
            row = []
            for at in (0, 4, 8, 12, 16):
                _capi.set_option("fwd_respread", at)
                x, it = run(kind, d, 1e-7, 1000, return_iters=True)
                if ref is None: ref = (x.clone(), it.clone())
                same = torch.equal(x, ref[0]) and torch.equal(it, ref[1])
                row.append("%d: %.1f us%s" % (at, t(lambda: run(kind, d, 1e-7, 1000, out=xo)), "" if same else " DIFFERS (%g, %d its)" % ((x - ref[0]).abs().max().item(), (it != ref[1]).sum().item())))
            print(kind, "B", B, "seed", seed, " | ".join(row), flush=True)
# iteration budgets that cut the solve short, NaN problems, heavy tails
B = 65536
for kind in ("qp", "qcqp"):
    d = {k: v.cuda() for k, v in make_problem(kind, B, N, 5).items()}
    d["P"][::97] *= -1.0            # non-convex: NaN / failure signalling
    for mi in (1, 2, 17, 18, 25):
        outs = []
        for at in (0, 16):
            _capi.set_option("fwd_respread", at)
            x, it = run(kind, d, 1e-7, mi, return_iters=True)
            outs.append((x, it))
        nan_same = torch.equal(torch.isnan(outs[0][0]), torch.isnan(outs[1][0]))
        eq = torch.equal(torch.nan_to_num(outs[0][0], nan=7.0), torch.nan_to_num(outs[1][0], nan=7.0)) and torch.equal(outs[0][1], outs[1][1])
        print(kind, "max_iter", mi, "identical" if (eq and nan_same) else "DIFFERS", "NaN problems", int(torch.isnan(outs[0][0]).any(1).sum()))
g = torch.Generator().manual_seed(3)
p = torch.exp(20 * torch.rand(B, N, generator=g, dtype=torch.float64) - 10)
for kind in ("qp", "qcqp"):
    d = {k: v.cuda() for k, v in make_problem(kind, B, N, 5).items()}
    d["P"] = torch.diag_embed(p).cuda()
    xo = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
    outs, row = [], []
    for at in (0, 16):
        _capi.set_option("fwd_respread", at)
        outs.append(run(kind, d, 1e-7, 1000, return_iters=True))
        row.append("%d: %.1f us" % (at, t(lambda: run(kind, d, 1e-7, 1000, out=xo), 10)))
    print("heavy tail", kind, " | ".join(row), "identical" if torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) else "DIFFERS")
