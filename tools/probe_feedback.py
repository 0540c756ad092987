"""Backward of a dense 8 x 8 batch through DQQ_P_AUTO with and without the feedback buffer (dqq_set_feedback): the drain launch
by the team kernel vs by the lane-per-problem kernel (LIST); and what the word costs the headline's empty drain."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import ops, _capi
def t(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) * 1e3 / n
for kind, N in (("qcqp", 8), ("qp", 8), ("qcqp", 6), ("qcqp", 4)):
    for B in (32768, 65536, 131072):
        for structure in ("dense", "diag"):
            d = {k: v.cuda() for k, v in make_problem(kind, B, N, 4250, structure).items()}
            if kind == "qp":
                x, it = ops.qp_forward(d["P"], d["q"], 1e-7, 1000, return_iters=True)
                out = [torch.empty_like(d["P"]), torch.empty_like(d["q"])]
                run = lambda: ops.qp_backward(d["P"], d["q"], x, d["grad_x"], out=out)
                rund = lambda: ops.qp_backward(d["P"], d["q"], x, d["grad_x"], out=out, layout=1)
            else:
                x, it = ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000, return_iters=True)
                out = [torch.empty_like(d["P"]), torch.empty_like(d["q"]), torch.empty_like(d["l_n"]), torch.empty_like(d["mu"])]
                run = lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, d["grad_x"], out=out)
                rund = lambda: ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x, d["grad_x"], out=out, layout=1)
            _capi.enable_feedback(False); ops._feedback_tried = True
            off = t(run)
            _capi.enable_feedback(True); _capi.set_option("lane_list_drains", 0); _capi.set_option("bwd_whole_batches", 0)
            _capi.set_option("bwd_skip_classify", 0)
            on = t(run)
            n = _capi.get_option("lane_list_drains")
            _capi.set_option("bwd_skip_classify", 1)
            whole = t(run)
            nw = _capi.get_option("bwd_whole_batches")
            dd = t(rund) if structure == "dense" else float("nan")
            print("%-5s N=%d B=%6d %-5s  AUTO backward: no feedback %.1f us, feedback %.1f us (%d lane drains of 45), whole batch %.1f us (%d of 45)   declared dense %.1f"
                  % (kind, N, B, structure, off, on, n, whole, nw, dd), flush=True)
