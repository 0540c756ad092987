"""Wave timeline of fwd_diag_kernel at the headline shape (debug library, tools/ubench/build_timeline.sh).

Per wave: entry, P consumed, inputs loaded, loop start, loop end, stores issued (100 MHz constant clock, 10 ns), the
largest iteration count among its problems and their sum.  Prints where the launch's microseconds go.
usage: python tools/probe_timeline.py [qp|qcqp] [B] [N] [layout 0|2] [fwd_compact 0|1]
"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_problem
from diffqcqp_amd import _capi
_capi.LIB_PATH = os.path.join(ROOT, "tools", "ubench", "bin", os.environ.get("DQQ_TL_LIB", "libdqq_timeline.so"))
from diffqcqp_amd import ops

kind = sys.argv[1] if len(sys.argv) > 1 else "qcqp"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
N = int(sys.argv[3]) if len(sys.argv) > 3 else 8
layout = int(sys.argv[4]) if len(sys.argv) > 4 else 0
compact = int(sys.argv[5]) if len(sys.argv) > 5 else 1
d = {k: v.cuda() for k, v in make_problem(kind, B, N, 1002).items()}
Pin = d["P"] if layout == 0 else torch.diagonal(d["P"], dim1=1, dim2=2).contiguous()
x = torch.empty(B, N, 1, dtype=torch.float64, device="cuda")
handle = _capi.lib()
_capi.set_option("fwd_compact", compact)
fn = handle.dqq_debug_set_timeline; fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
NW = 1 << 16
buf = torch.zeros(NW, 32, dtype=torch.int64, device="cuda")

def run():
    if kind == "qp": ops.qp_forward(Pin, d["q"], 1e-7, 1000, layout=layout, out=x)
    else: ops.qcqp_forward(Pin, d["q"], d["l_n"], d["mu"], 1e-7, 1000, layout=layout, out=x)
for _ in range(5): run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); run(); b.record(); b.synchronize()
print("un-instrumented launch (events): %.1f us" % (a.elapsed_time(b) * 1e3))
assert fn(buf.data_ptr()) == 0
for rep in range(3):
    buf.zero_(); torch.cuda.synchronize()
    a.record(); run(); b.record(); b.synchronize()
    t = buf.cpu().numpy()
    t = t[t[:, 0] != 0]
    us = lambda v: v * 0.01
    s0 = t[:, 0].min(); end = t[:, 5].max()
    print("rep %d: %d waves, instrumented launch %.1f us (events), first entry -> last exit %.2f us" %
          (rep, len(t), a.elapsed_time(b) * 1e3, us(end - s0)))
q = lambda v: "min %.2f  p10 %.2f  med %.2f  p90 %.2f  max %.2f" % tuple(us(np.percentile(v, [0, 10, 50, 90, 100])))
print("entry after first entry      ", q(t[:, 0] - s0))
print("P consumed after entry       ", q(t[:, 1] - t[:, 0]))
print("P consumed after first entry ", q(t[:, 1] - s0))
print("other inputs loaded          ", q(t[:, 2] - t[:, 1]))
print("prologue (power it., pow)    ", q(t[:, 3] - t[:, 2]))
print("ADMM loop                    ", q(t[:, 4] - t[:, 3]))
print("stores                       ", q(t[:, 5] - t[:, 4]))
print("exit after first entry       ", q(t[:, 5] - s0))
mx, sm = t[:, 6], t[:, 7]
print("wave max iterations: mean %.2f max %d; mean of problems %.2f" % (mx.mean(), mx.max(), sm.sum() / B))
loop = us(t[:, 4] - t[:, 3])
print("loop us per iteration of the wave (loop / max iterations): med %.3f p10 %.3f p90 %.3f" %
      tuple(np.percentile(loop / np.maximum(mx, 1), [50, 10, 90])))
for c in range(8):
    a, b, e = t[:, 8 + 3 * c], t[:, 9 + 3 * c], t[:, 10 + 3 * c]
    m = (a != 0) & (e != 0)
    if m.sum() == 0: break
    prev = t[:, 3] if c == 0 else t[:, 10 + 3 * (c - 1)]
    print("checkpoint %d (%4d waves): segment %s | write-out %s | count exchange + barrier %s" % (
        c, m.sum(), "med %.2f p90 %.2f" % tuple(us(np.percentile((a - prev)[m], [50, 90]))),
        "med %.2f p90 %.2f" % tuple(us(np.percentile((b - a)[m], [50, 90]))),
        "med %.2f p90 %.2f max %.2f" % tuple(us(np.percentile((e - b)[m], [50, 90, 100])))))
# waves still inside the loop over time
for tt in range(0, int(us(end - s0)) + 2, 2):
    c = s0 + tt * 100
    inl = ((t[:, 3] <= c) & (t[:, 4] > c)).sum(); pre = (t[:, 3] > c).sum(); done = (t[:, 4] <= c).sum()
    print("t=%2d us: before loop %4d  in loop %4d  past loop %4d" % (tt, pre, inl, done))
