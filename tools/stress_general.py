"""One-off stress of the general (dense P) kernels against the oracle: larger batches than tests/ use.
Prints, per (kind, N): max |x - x_oracle|, share of identical iteration counts, and for the backward the
share of identical refinement exits and the max relative gradient error where they agree."""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import make_problem
from diffqcqp_amd import ops
from oracle import oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
worst = 0.0
for kind in ("qp", "qcqp", "box", "sbox"):
    for N in (2, 4, 6, 8, 10, 12, 14, 16, 32, 64):
        Bn = B if N <= 16 else max(B // 16, 64)
        d = make_problem(kind, Bn, N, 4000 + N, "dense")
        g = {k: v.cuda() for k, v in d.items()}
        P, q = d["P"].numpy(), d["q"].numpy()
        if kind == "qp":
            xo, ito = O.qp_fwd_batch(P, q, 1e-7, 1000, nthreads=64)
            xh, ith = ops.qp_forward(g["P"], g["q"], 1e-7, 1000, layout=1, return_iters=True)
        elif kind == "qcqp":
            xo, ito = O.qcqp_fwd_batch(P, q, d["l_n"].numpy(), d["mu"].numpy(), 1e-7, 1000, nthreads=64)
            xh, ith = ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, layout=1, return_iters=True)
        else:
            v = d["v"].numpy() if kind == "sbox" else None
            xo, ito = O.boxqp_fwd_batch(P, q, d["l_min"].numpy(), d["l_max"].numpy(), 1e-7, 1000, v=v, nthreads=64)
            xh, ith = ops.boxqp_forward(g["P"], g["q"], g["l_min"], g["l_max"], 1e-7, 1000, v=g.get("v"), layout=1,
                                        return_iters=True)
        err = np.abs(xh.cpu().numpy() - xo).max()
        worst = max(worst, err)
        line = f"{kind:5s} N={N:2d} B={Bn:5d}  fwd max|dx| {err:.2e}  iters equal {(ith.cpu().numpy() == ito).mean():.4f}"
        xg = torch.from_numpy(xo).cuda()
        gx = d["grad_x"].numpy()
        try:
            if kind == "qp":
                ref = O.qp_bwd_batch(P, q, xo, gx, nthreads=64)
                out = ops.qp_backward(g["P"], g["q"], xg, g["grad_x"], layout=1, return_steps=True)
                grads, st, gref, sref = out[:2], out[2].cpu().numpy(), ref[:2], ref[2]
            elif kind == "qcqp":
                ref = O.qcqp_bwd_batch(P, q, d["l_n"].numpy(), d["mu"].numpy(), xo, gx, nthreads=64)
                out = ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], xg, g["grad_x"], layout=1, return_steps=True)
                grads, st, gref, sref = out[:4], out[4].cpu().numpy(), ref[:4], ref[4]
            elif kind == "box":
                ref = O.boxqp_bwd_batch(P, q, d["l_min"].numpy(), d["l_max"].numpy(), xo, gx, nthreads=64)
                out = ops.boxqp_backward(g["P"], g["q"], g["l_min"], g["l_max"], xg, g["grad_x"], layout=1, return_steps=True)
                grads, st, gref, sref = out[:4], out[4].cpu().numpy()[:, 1], ref[:4], ref[5][:, 1]
            else:
                raise ValueError("no backward")
            same = st == sref
            rel = 0.0
            for a, b in zip(grads, gref):
                a, b = a.cpu().numpy()[same], b[same]
                sc = np.maximum(1.0, np.abs(b).reshape(b.shape[0], -1).max(1)).reshape((-1,) + (1,) * (b.ndim - 1))
                rel = max(rel, (np.abs(a - b) / sc).max())
            line += f"  | bwd exits equal {same.mean():.4f}  max rel err {rel:.2e}"
        except ValueError as e:
            line += f"  | bwd: {str(e)[:40]}"
        print(line, flush=True)
print("worst forward error", worst)
