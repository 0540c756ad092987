"""GPU box: the reference's own hard-coded matrices (tests/golden/reference_inputs.py) and seeded rank-deficient
dense batches through every route of the HIP path (AUTO / DENSE; B = 1, 33, 4096 replicas of the literal cases),
forward and backward, against the oracle: iteration counts, refinement step counts, max error per output.
Usage: python tools/probe_illcond.py [quick]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import reference_inputs as R  # noqa: E402
from diffqcqp_amd import ops, _capi  # noqa: E402
from oracle import oracle as O  # noqa: E402

O.build()
NT = min(32, os.cpu_count() or 1)


def relerr(a, b):
    a = a.detach().cpu().numpy()
    if not np.isfinite(a).all():
        return float("nan")
    s = np.maximum(1.0, np.abs(b).reshape(b.shape[0], -1).max(1)).reshape((-1,) + (1,) * (b.ndim - 1))
    return float((np.abs(a - b) / s).max())


def run(name, kind, d, eps, max_iter, layouts=(0, 1)):
    P, q, g = d["P"], d["q"], d["grad_x"]
    if kind == "qp":
        xo, ito = O.qp_fwd_batch(P, q, eps, max_iter, nthreads=NT)
        ref = O.qp_bwd_batch(P, q, xo, g, nthreads=NT)
    else:
        xo, ito = O.qcqp_fwd_batch(P, q, d["l_n"], d["mu"], eps, max_iter, nthreads=NT)
        ref = O.qcqp_bwd_batch(P, q, d["l_n"], d["mu"], xo, g, nthreads=NT)
    *gref, sref = ref
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
    xs = torch.from_numpy(xo).cuda()
    for lay in layouts:
        if kind == "qp":
            xh, ith = ops.qp_forward(t["P"], t["q"], eps, max_iter, layout=lay, return_iters=True)
            *gh, sh = ops.qp_backward(t["P"], t["q"], xs, t["grad_x"], layout=lay, return_steps=True)
            *ge, se = ops.qp_backward(t["P"], t["q"], xh, t["grad_x"], layout=lay, return_steps=True)
        else:
            xh, ith = ops.qcqp_forward(t["P"], t["q"], t["l_n"], t["mu"], eps, max_iter, layout=lay, return_iters=True)
            *gh, sh = ops.qcqp_backward(t["P"], t["q"], t["l_n"], t["mu"], xs, t["grad_x"], layout=lay,
                                        return_steps=True)
            *ge, se = ops.qcqp_backward(t["P"], t["q"], t["l_n"], t["mu"], xh, t["grad_x"], layout=lay,
                                        return_steps=True)
        torch.cuda.synchronize()
        itm = float((ith.cpu().numpy() == ito).mean())
        sm = float((sh.cpu().numpy() == sref).mean())
        same = se.cpu().numpy() == sref
        e2e = [relerr(a[torch.from_numpy(same).cuda()], b[same]) if same.any() else 0.0 for a, b in zip(ge, gref)]
        print("%-28s %-4s B=%-5d N=%-2d %s | x err %.1e (scale %.1e) iters== %.4f (oracle mean %.1f max %d) | "
              "bwd same-x: steps== %.4f %s max rel err %s | e2e steps== %.4f err %s" % (
                  name, kind, q.shape[0], q.shape[1], "AUTO " if lay == 0 else "DENSE",
                  relerr(xh, xo), np.abs(xo).max(), itm, ito.mean(), ito.max(), sm,
                  np.bincount(sref).tolist(), ["%.1e" % relerr(a, b) for a, b in zip(gh, gref)], same.mean(),
                  ["%.1e" % e for e in e2e]), flush=True)


def literal(P, q, rad=None, B=1, seed=0):
    n = q.size
    rng = np.random.default_rng(seed)
    d = {"P": np.broadcast_to(P, (B, n, n)).copy(), "q": np.broadcast_to(q.reshape(n, 1), (B, n, 1)).copy(),
         "grad_x": rng.standard_normal((B, n, 1))}
    if rad is not None:
        d["l_n"] = np.broadcast_to(rad.reshape(-1, 1), (B, n // 2, 1)).copy()
        d["mu"] = np.ones((B, n // 2, 1))
    return d


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    print(_capi.version())
    for kv in os.environ.get("DQQ_OPTS", "").split(","):
        if "=" in kv:
            k, v = kv.split("=")
            _capi.set_option(k, int(v))
            print("option", k, "=", v)
    if len(sys.argv) > 1 and sys.argv[1] == "qcqpbwd":  # QCQP backward 16 < N <= 32: wave kernel vs reference-order kernel
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import make_problem
        for fam in ("dense", "lowrank", "duprows"):
            for N in ((24, 32) if len(sys.argv) < 3 else tuple(int(v) for v in sys.argv[2].split(","))):
                B = 4096 if N <= 32 else 1024
                if fam == "dense":
                    d = {k: v.numpy() for k, v in make_problem("qcqp", B, N, 7100 + N, "dense").items()}
                else:
                    d = {k: v.numpy() for k, v in R.rank_deficient("qcqp", B, N, 7000 + N, fam).items()}
                run(fam, "qcqp", d, 1e-7, 1000, layouts=(1,))
                t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
                xh = ops.qcqp_forward(t["P"], t["q"], t["l_n"], t["mu"], 1e-7, 1000, layout=1)
                ms = timed(lambda: ops.qcqp_backward(t["P"], t["q"], t["l_n"], t["mu"], xh, t["grad_x"], layout=1))
                print("   backward %.3f ms per %d problems" % (ms, B), flush=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "qpbwd":   # only the QP backward of the rank-deficient families, timed
        for fam in ("lowrank", "duprows", "dense"):
            for N in (32, 64):
                B = 4096
                if fam == "dense":
                    sys.path.insert(0, os.path.join(ROOT, "tests"))
                    from conftest import make_problem
                    d = {k: v.numpy() for k, v in make_problem("qp", B, N, 7100 + N, "dense").items()}
                else:
                    d = {k: v.numpy() for k, v in R.rank_deficient("qp", B, N, 7000 + N, fam).items()}
                run(fam, "qp", d, 1e-7, 1000, layouts=(1,))
                t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
                xh = ops.qp_forward(t["P"], t["q"], 1e-7, 1000, layout=1)
                ms = timed(lambda: ops.qp_backward(t["P"], t["q"], xh, t["grad_x"], layout=1))
                print("   backward %.3f ms per %d problems" % (ms, B), flush=True)
        return
    for B in ((1, 33) if quick else (1, 33, 4096)):
        P, q, ln = R.m2_singular()
        run("m2_singular", "qp", literal(P, q, None, B), 1e-10, 1000)
        run("m2_singular max_iter=1", "qp", literal(P, q, None, B), 1e-10, 1)
        run("m2_singular", "qcqp", literal(P, q, ln, B), 1e-10, 1000)
        P, q, ln = R.m2_first()
        run("m2_first", "qp", literal(P, q, None, B), 1e-10, 1000)
        run("m2_first", "qcqp", literal(P, q, ln, B), 1e-10, 1000)
        P, q = R.g2_product()
        run("g2_product", "qp", literal(P, q, None, B), 1e-10, 1000)
        run("g2_product r=.1", "qcqp", literal(P, q, np.full(6, 0.1), B), 1e-10, 1000)
        run("g2_product r=1e4", "qcqp", literal(P, q, np.full(6, 1e4), B), 1e-10, 1000)
        P, q, rad = R.g_blockdiag()
        run("g_blockdiag", "qp", literal(P, q, None, B), 1e-10, 1000)
        for r in rad:
            run("g_blockdiag r0=%.3g" % r[0], "qcqp", literal(P, q, r, B), 1e-10, 1000)
        P, q, ln = R.g4_delassus()
        for s in (1.0, -1.0):
            run("g4_delassus q*%+d" % s, "qp", literal(P, s * q, None, B), 1e-10, 1000)
            run("g4_delassus q*%+d" % s, "qcqp", literal(P, s * q, ln, B), 1e-10, 1000)
    for fam in ("lowrank", "duprows", "psd_eps"):
        for N in (8, 32, 64):
            for kind in ("qp", "qcqp"):
                B = 256 if quick else (4096 if N < 64 else 2048)
                d = {k: v.numpy() for k, v in R.rank_deficient(kind, B, N, 7000 + N, fam).items()}
                run(fam, kind, d, 1e-7, 1000)


if __name__ == "__main__":
    main()
