"""Randomised soak of the segmented work-list (N = 32 / 64 through DQQ_P_AUTO): random batch sizes, dense-problem
patterns, workgroup shapes; the queued problems must carry the bits of the same batch declared DQQ_P_DENSE, the
header must be zero afterwards.   python tools/soak_segmented.py [trials] [seed]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
HDR = 32 + 3 * 32 * 32
bad = 0
for t in range(trials):
    kind = str(rng.choice(["qp", "qcqp"])); N = int(rng.choice([32, 64]))
    B = int(rng.integers(1, 20000 if N == 32 else 6000))
    dd, dg = make_problem(kind, B, N, 100 + t, "dense"), make_problem(kind, B, N, 100 + t, "diag")
    mode = int(rng.integers(4))
    idx = torch.arange(B)
    if mode == 0: sel = torch.from_numpy(rng.random(B) < rng.choice([0.001, 0.02, 0.3]))
    elif mode == 1: sel = ((idx // 16) % 32 == int(rng.integers(32)))
    elif mode == 2: sel = torch.ones(B, dtype=torch.bool)
    else: sel = idx >= int(rng.integers(B))
    d = dict(dd); d["P"] = torch.where(sel.view(B, 1, 1), dd["P"], dg["P"]).contiguous()
    g = {k: v.cuda() for k, v in d.items()}
    wpb_draw = int(rng.choice([0, 1]))
    if _capi.tuning_build(): _capi.set_option("wpb", wpb_draw)   # (developer build only: csrc/tuning.h)
    def fwd(lay):
        if kind == "qp": return ops.qp_forward(g["P"], g["q"], 1e-7, 1000, layout=lay, return_iters=True)
        return ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, layout=lay, return_iters=True)
    def bwd(lay, x, cache=None):
        if kind == "qp": return ops.qp_backward(g["P"], g["q"], x, g["grad_x"], layout=lay, return_steps=True, cache=cache)
        return ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], x, g["grad_x"], layout=lay, return_steps=True, cache=cache)
    xd, itd = fwd(1); gd = bwd(1, xd)
    cache = ops.diag_cache(g["q"]) if rng.integers(2) else None
    if cache is not None:
        if kind == "qp": xa, ita = ops.qp_forward(g["P"], g["q"], 1e-7, 1000, return_iters=True, cache=cache)
        else: xa, ita = ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, return_iters=True, cache=cache)
    else:
        xa, ita = fwd(0)
    ga = bwd(0, xd, cache)
    torch.cuda.synchronize()
    nd = sel.numpy()
    ok = all(int(ws[:HDR].abs().sum()) == 0 for ws in ops._workspaces.values())
    ok = ok and np.array_equal(xa.cpu().numpy()[nd], xd.cpu().numpy()[nd]) and np.array_equal(ita.cpu().numpy()[nd], itd.cpu().numpy()[nd])
    for a, b in zip(ga, gd):
        ok = ok and np.array_equal(a.cpu().numpy()[nd], b.cpu().numpy()[nd])
    ok = ok and float((xa - xd).abs().max()) <= 1e-9 and bool(torch.isfinite(xa).all())
    if not ok:
        bad += 1
        print("FAIL", t, kind, N, B, mode, int(sel.sum()), flush=True)
if _capi.tuning_build(): _capi.set_option("wpb", 0)
print("%d trials, %d failures" % (trials, bad))
