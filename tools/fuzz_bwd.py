"""Randomised parity of the backward routes against the oracle (x from the oracle on both sides): random kind, N, batch
size, structure, layout, fused / work-list fallback, with and without the forward's diagonal hand-off.
usage: python tools/fuzz_bwd.py [trials] [seed] [big | lane]   (lane: only the batches the lane-per-problem backward takes --
N <= 8, QP / QCQP, B >= 16384 -- declared dense or through DQQ_P_AUTO with the report word of _capi.py poked)"""
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
            if k not in ("dense_wave64", "wave_qcqp_bwd"): _capi.set_option(k, v)
    return ref


trials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad, worst, lane_list = 0, 0.0, 0
_capi.set_option("lane_list_drains", 0)
_capi.set_option("bwd_whole_batches", 0)
for t in range(trials):
    kind = str(rng.choice(["qp", "qcqp", "box"]))
    LANE = len(sys.argv) > 3 and sys.argv[3] == "lane"
    if LANE: kind = str(rng.choice(["qp", "qcqp"]))
    N = int(rng.choice([2, 4, 8]) if LANE else rng.choice([2, 4, 8, 16, 32] if len(sys.argv) <= 3 else [2, 3, 5, 6, 8, 10, 12, 16, 17, 20, 22, 24, 32, 33, 40, 44, 48, 56, 64, 70]))
    if kind == "qcqp" and N % 2: N += 1
    B = int(rng.choice([1, 5, 16, 17, 64, 130, 1000, 2049]))
    if N <= 8 and kind != "box" and (LANE or rng.integers(8) == 0):
        B = int(rng.choice([16384, 16421, 24576, 24700, 33001]))   # the lane-per-problem backward (bwd_lane_dense.hip: DQQ_P_DENSE, or the drain of a long list)
    if N >= 32: B = min(B, 130 if N == 32 else 17)
    if N > 16 and kind == 'box': B = min(B, 17)
    structure = str(rng.choice(["diag", "dense", "mixed", "nonsym"]))
    layout = 0 if structure in ("diag", "mixed") else int(rng.choice([0, 1]))
    if N not in (2, 4, 8, 16, 32, 64) and structure == "diag": structure = "dense"
    opts = {"fuse_fallback": int(rng.choice([-1, 0, 1])), "wpb": int(rng.choice([0, 1, 4])),
            "small_bwd": int(rng.choice([0, 1])), "dense_teams": int(rng.choice([0, 1])),
            "dense_wave64": int(rng.choice([0, 1])), "wave_qcqp_bwd": int(rng.choice([0, 1])), "lane_bwd": int(rng.choice([0, 1, 1]))}
    use_cache = bool(rng.integers(2)) and layout == 0
    d = make_problem(kind, B, N, 7000 + t, "dense" if structure == "nonsym" else structure)
    if structure == "nonsym":
        gg = torch.Generator().manual_seed(t)
        d["P"] = (d["P"] + torch.triu(torch.rand(B, N, N, generator=gg, dtype=torch.float64), diagonal=1) * 0.05).contiguous()
    P, q, gx = d["P"].numpy(), d["q"].numpy(), d["grad_x"].numpy()
    ref_flag = apply_opts(opts)
    if B >= 16384 and layout == 0 and kind != "box":
        # the report word (_capi.py), set to anything: a "long list" sends the lane-per-problem kernel (LIST) after
        # whatever list this batch has -- full, every other tile, empty; a hint must never change a result
        _capi.enable_feedback(True)
        word = (int(rng.integers(4)) << 62) | (B << 32) | int(rng.choice([0, 30000, B, B]))   # (streak, B, entries): launch.h
        _capi._feedback[(0 if kind == "qp" else 1) * 4 + N // 2 - 1] = word - (1 << 64) if word >= (1 << 63) else word
        lane_list += 1
    g = {k: v.cuda() for k, v in d.items()}
    cache = ops.diag_cache(g["q"]) if use_cache else None
    if kind == "qp":
        xo, _ = O.qp_fwd_batch(P, q, 1e-7, 1000, nthreads=16)
        ref = O.qp_bwd_batch(P, q, xo, gx, nthreads=16)
        if use_cache: ops.qp_forward(g["P"], g["q"], 1e-7, 1000, cache=cache)
        out = ops.qp_backward(g["P"], g["q"], torch.from_numpy(xo).cuda(), g["grad_x"], layout=layout | ref_flag, return_steps=True, cache=cache)
        grads, st, gref, sref = out[:2], out[2].cpu().numpy(), ref[:2], ref[2]
    elif kind == "qcqp":
        xo, _ = O.qcqp_fwd_batch(P, q, d["l_n"].numpy(), d["mu"].numpy(), 1e-7, 1000, nthreads=16)
        ref = O.qcqp_bwd_batch(P, q, d["l_n"].numpy(), d["mu"].numpy(), xo, gx, nthreads=16)
        if use_cache: ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, cache=cache)
        out = ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], torch.from_numpy(xo).cuda(), g["grad_x"], layout=layout | ref_flag, return_steps=True, cache=cache)
        grads, st, gref, sref = out[:4], out[4].cpu().numpy(), ref[:4], ref[4]
    else:
        xo, _ = O.boxqp_fwd_batch(P, q, d["l_min"].numpy(), d["l_max"].numpy(), 1e-7, 1000, nthreads=16)
        ref = O.boxqp_bwd_batch(P, q, d["l_min"].numpy(), d["l_max"].numpy(), xo, gx, nthreads=16)
        if use_cache: ops.boxqp_forward(g["P"], g["q"], g["l_min"], g["l_max"], 1e-7, 1000, cache=cache)
        out = ops.boxqp_backward(g["P"], g["q"], g["l_min"], g["l_max"], torch.from_numpy(xo).cuda(), g["grad_x"], layout=layout | ref_flag, return_steps=True, cache=cache)
        grads, st, gref, sref = out[:4], out[4].cpu().numpy()[:, 1], ref[:4], ref[5][:, 1]
    same = st == sref
    rel = 0.0
    # the QCQP's contact gradients through the matrix-core kernels (16 < N <= 64, wave_qcqp_bwd = 1): the evaluation-order
    # noise of the reference's own formulas is up to 8.6e-6 there (tests/test_gpu_parity.py: REASSOC_TOL); judged at 2e-5
    reassoc = kind == "qcqp" and 16 < N <= 64 and ref_flag == 0 and structure != "diag"
    for k, (a, b) in enumerate(zip(grads, gref)):
        a, b = a.cpu().numpy()[same], b[same]
        if a.size:
            sc = np.maximum(1.0, np.abs(b).reshape(b.shape[0], -1).max(1)).reshape((-1,) + (1,) * (b.ndim - 1))
            r = float((np.abs(a - b) / sc).max())
            rel = max(rel, r / 20 if (reassoc and k >= 2) else r)
    finite = all(np.isfinite(a.cpu().numpy()).all() for a in grads)
    worst = max(worst, rel)
    ok = rel <= 1e-6 and finite and (same.mean() >= 0.9 or B < 64)
    if not ok:
        bad += 1
        print("FAIL", t, kind, N, B, structure, layout, opts, "cache", use_cache, "rel %.2e exits equal %.3f finite %s" % (rel, same.mean(), finite), flush=True)
apply_opts({"fuse_fallback": -1, "wpb": 0, "small_bwd": 1, "dense_teams": 1, "dense_wave64": 1, "wave_qcqp_bwd": 1, "lane_bwd": 1})
print("%d trials, %d failures, worst rel err %.2e  (feedback word poked in %d trials: %d drains by the lane kernel, %d whole batches)"
      % (trials, bad, worst, lane_list, _capi.get_option("lane_list_drains"), _capi.get_option("bwd_whole_batches")))
