"""One-off stress of the diagonal fast paths (AUTO layout, diagonal P in (B,N,N)) against the oracle."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import make_problem
from diffqcqp_amd import ops
from oracle import oracle as O
B0 = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for kind in ("qp", "qcqp", "box", "sbox"):
    for N in (2, 4, 8, 16, 32, 64):
        B = B0 if N <= 16 else B0 // 8
        d = make_problem(kind, B, N, 5000 + N); g = {k: v.cuda() for k, v in d.items()}
        P, q, gx = d["P"].numpy(), d["q"].numpy(), d["grad_x"].numpy()
        if kind == "qp":
            xo, ito = O.qp_fwd_batch(P, q, 1e-7, 1000, nthreads=64)
            xh, ith = ops.qp_forward(g["P"], g["q"], 1e-7, 1000, return_iters=True)
            ref = O.qp_bwd_batch(P, q, xo, gx, nthreads=64)
            out = ops.qp_backward(g["P"], g["q"], torch.from_numpy(xo).cuda(), g["grad_x"], return_steps=True)
            gr, st, rr, sr = out[:2], out[2], ref[:2], ref[2]
        elif kind == "qcqp":
            ln, mu = d["l_n"].numpy(), d["mu"].numpy()
            xo, ito = O.qcqp_fwd_batch(P, q, ln, mu, 1e-7, 1000, nthreads=64)
            xh, ith = ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, return_iters=True)
            ref = O.qcqp_bwd_batch(P, q, ln, mu, xo, gx, nthreads=64)
            out = ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], torch.from_numpy(xo).cuda(), g["grad_x"], return_steps=True)
            gr, st, rr, sr = out[:4], out[4], ref[:4], ref[4]
        else:
            lo, hi = d["l_min"].numpy(), d["l_max"].numpy()
            v = d["v"].numpy() if kind == "sbox" else None
            xo, ito = O.boxqp_fwd_batch(P, q, lo, hi, 1e-7, 1000, v=v, nthreads=64)
            xh, ith = ops.boxqp_forward(g["P"], g["q"], g["l_min"], g["l_max"], 1e-7, 1000, v=g.get("v"), return_iters=True)
            gr = None
            if kind == "box":
                ref = O.boxqp_bwd_batch(P, q, lo, hi, xo, gx, nthreads=64)
                out = ops.boxqp_backward(g["P"], g["q"], g["l_min"], g["l_max"], torch.from_numpy(xo).cuda(), g["grad_x"], return_steps=True)
                gr, st, rr, sr = out[:4], out[4], ref[:4], ref[5]
        line = f"{kind:5s} N={N:2d} B={B:6d} fwd max|dx| {np.abs(xh.cpu().numpy() - xo).max():.1e} iters equal {(ith.cpu().numpy() == ito).mean():.5f}"
        if gr is not None:
            exact = all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(gr, rr)) and np.array_equal(st.cpu().numpy(), sr)
            line += f" | bwd bit-exact {exact}"
        print(line, flush=True)
