"""Randomised parity of the forward routes against the oracle: random kind, N (small set by default, `big` as third
argument: 2..70 incl. odd sizes), batch size, lanes per problem, structure (diag / dense / mixed / non-symmetric), layout, fused or work-list fallback, compaction, eps and
iteration budget.  usage: python tools/fuzz_small.py [trials] [seed] [big]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import _capi, ops
from oracle import oracle as O

TUNING = _capi.tuning_build()   # the developer build (-DDQQ_TUNING): the kernel-selection knobs exist; the shipped library has none


def apply_opts(opts):
    """Knobs -> the library (developer build only; on the shipped build the draw still happens, so that a seed names the same
    trials on both).  The two former NUMERICS knobs are the per-call flag DQQ_F_REFERENCE_ORDER now: returned for the layout."""
    ref = _capi.F_REFERENCE_ORDER if (opts.get("dense_wave64", 1) == 0 or opts.get("wave_qcqp_bwd", 1) == 0) else 0
    if TUNING:
        for k, v in opts.items():
            if k not in ("dense_wave64", "wave_qcqp_bwd", "fwd_compact"): _capi.set_option(k, v)   # (fwd_compact: removed in round 5; still drawn so that a seed names the same trials)
    return ref


trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
LPP = {2: [1], 4: [1, 2], 8: [1, 2, 4], 16: [2, 4, 8], 32: [4, 8, 16], 64: [8, 16, 32]}
BIG = len(sys.argv) > 3
SIZES = [2, 4, 8] if not BIG else [2, 3, 4, 5, 6, 8, 10, 12, 16, 17, 20, 24, 32, 33, 40, 48, 56, 64, 70]
worst, bad = 0.0, 0
for t in range(trials):
    kind = rng.choice(["qp", "qcqp", "box", "sbox"])
    N = int(rng.choice(SIZES))
    if kind == "qcqp" and N % 2: N += 1
    B = int(rng.choice([1, 3, 31, 32, 33, 64, 100, 257, 1000, 4097]))
    if N > 16: B = min(B, 100)
    if N > 8: B = min(B, 1000)
    structure = str(rng.choice(["diag", "dense", "mixed", "nonsym"]))
    fast = N in LPP
    layout = int(rng.choice([0, 0, 1])) if structure != "diag" else int(rng.choice([0, 2]) if fast else 0)
    if structure == "mixed": layout = 0
    eps = float(rng.choice([1e-7, 1e-7, 1e-10, 1e-5]))
    max_iter = int(rng.choice([1000, 1000, 1, 7, 16, 40]))
    opts = {"fwd_lpp": int(rng.choice([0] + LPP.get(N, []))), "dense_wave64": int(rng.choice([0, 1])),
            "lane_dense": int(rng.choice([0, 1])), "small_fwd": int(rng.choice([0, 1])), "fuse_fallback": int(rng.choice([-1, 0, 1])),
            "fwd_compact": int(rng.choice([0, 1])), "wpb": int(rng.choice([0, 1, 4])),
            "lane_defer": int(rng.choice([0, 1, 3, 7])), "fwd_respread": int(rng.choice([0, 5, 16])),
            "fwd_respread2": int(rng.choice([0, 3, 8])), "fwd_respread2_from": int(rng.choice([0, 0, 12, 48]))}
    d = make_problem(kind, B, N, 9000 + t, "dense" if structure == "nonsym" else structure)
    if structure == "nonsym":
        g = torch.Generator().manual_seed(t)
        d["P"] = (d["P"] + torch.triu(torch.rand(B, N, N, generator=g, dtype=torch.float64), diagonal=1) * 0.05).contiguous()
    P, q = d["P"].numpy(), d["q"].numpy()
    if kind == "qp":
        xo, ito = O.qp_fwd_batch(P, q, eps, max_iter, nthreads=16)
    elif kind == "qcqp":
        xo, ito = O.qcqp_fwd_batch(P, q, d["l_n"].numpy(), d["mu"].numpy(), eps, max_iter, nthreads=16)
    else:
        v = d["v"].numpy() if kind == "sbox" else None
        xo, ito = O.boxqp_fwd_batch(P, q, d["l_min"].numpy(), d["l_max"].numpy(), eps, max_iter, v=v, nthreads=16)
    ref_flag = apply_opts(opts)
    g = {k: v.cuda() for k, v in d.items()}
    Pin = g["P"] if layout != 2 else torch.diagonal(g["P"], dim1=1, dim2=2).contiguous()
    if kind == "qp":
        xh, ith = ops.qp_forward(Pin, g["q"], eps, max_iter, layout=layout | ref_flag, return_iters=True)
    elif kind == "qcqp":
        xh, ith = ops.qcqp_forward(Pin, g["q"], g["l_n"], g["mu"], eps, max_iter, layout=layout | ref_flag, return_iters=True)
    else:
        xh, ith = ops.boxqp_forward(Pin, g["q"], g["l_min"], g["l_max"], eps, max_iter, v=g.get("v"), layout=layout | ref_flag,
                                    return_iters=True)
    err = float(np.abs(xh.cpu().numpy() - xo).max())
    same = float((ith.cpu().numpy() == ito).mean())
    worst = max(worst, err)
    ok = err <= 1e-6 and same >= (0.97 if B >= 100 else 0.0) and np.isfinite(xh.cpu().numpy()).all()
    if not ok:
        bad += 1
        print("FAIL", t, kind, N, B, structure, layout, eps, max_iter, opts, "err %.2e iters equal %.4f" % (err, same), flush=True)
apply_opts({"fwd_lpp": 0, "fuse_fallback": -1, "fwd_compact": 0, "wpb": 0, "dense_wave64": 1, "lane_dense": 1,
            "small_fwd": 1, "lane_defer": 0, "fwd_respread": 16, "fwd_respread2": 8, "fwd_respread2_from": 48})
print("%d trials, %d failures, worst |dx| %.2e" % (trials, bad, worst))
