"""A/B of two builds of the C ABI on ONE box (boxes of the pool differ by several per cent): the four launches of the headline
step (B = 65536, N = 8, diagonal P through DQQ_P_AUTO, with the forward's diagonal hand-off), each timed as a run of 100
back-to-back calls between two HIP events, alternating A, B, A, B ... five times; medians.  Only the four entry points whose
signatures every build shares are bound (raw ctypes).   usage: python tools/ab_libs.py libA.so libB.so"""
import ctypes, os, sys
import torch

F64 = torch.float64
vp, i32, i64, dbl, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_size_t


def bind(path):
    L = ctypes.CDLL(path)
    L.dqq_workspace_bytes.argtypes, L.dqq_workspace_bytes.restype = [i64], sz
    L.dqq_qp_fwd_f64.argtypes = [vp, vp, vp, i64, i32, dbl, dbl, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    L.has_report = hasattr(L, "dqq_hint_flags")   # builds since the stateless hint protocol: one more pointer (`report`) in the backward
    rep = [vp] if L.has_report else []
    L.dqq_qp_bwd_f64.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, dbl, i32, vp, vp, vp] + rep + [vp, sz, vp]
    L.dqq_qcqp_fwd_f64.argtypes = [vp, vp, vp, vp, vp, i64, i32, dbl, dbl, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    L.dqq_qcqp_bwd_f64.argtypes = [vp] * 12 + [i64, i32, dbl, i32, vp, vp, vp] + rep + [vp, sz, vp]
    return L


def main():
    import torch  # (torch first: its libamdhip64 must be the process's HIP runtime)
    libs = [bind(os.path.abspath(p)) for p in sys.argv[1:3]]
    B, N = 65536, 8
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(1000)
    r = lambda *s: torch.rand(*s, generator=g, dtype=F64, device=dev)
    P = torch.diag_embed(r(B, N) + 0.1).contiguous(); q = 2 * r(B, N, 1) - 1
    ln, mu = r(B, 4, 1), r(B, 4, 1)
    gx = torch.randn(B, N, 1, generator=g, dtype=F64, device=dev)
    x, gP, gq = torch.empty(B, N, 1, dtype=F64, device=dev), torch.empty(B, N, N, dtype=F64, device=dev), torch.empty(B, N, 1, dtype=F64, device=dev)
    gl, gm = torch.empty(B, 4, 1, dtype=F64, device=dev), torch.empty(B, 4, 1, dtype=F64, device=dev)
    pd, fl = torch.empty(B, N, dtype=F64, device=dev), torch.empty(B, dtype=torch.uint8, device=dev)
    p = lambda t: t.data_ptr()
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    outs = []
    for li, L in enumerate(libs):
        rep = (None,) if L.has_report else ()
        wsb = L.dqq_workspace_bytes(B)
        ws = torch.zeros((wsb + 3) // 4, dtype=torch.int32, device=dev)
        calls = {
            "qp_fwd": lambda L=L, ws=ws, wsb=wsb: L.dqq_qp_fwd_f64(p(P), p(q), p(x), B, N, 1e-7, 1e-7, 1000, 1, 0, None, p(pd), p(fl), p(ws), wsb, s),
            "qp_bwd": lambda L=L, ws=ws, wsb=wsb, rep=rep: L.dqq_qp_bwd_f64(p(P), p(q), p(x), p(gx), p(gP), p(gq), B, N, 1e-10, 0, None, p(pd), p(fl), *rep, p(ws), wsb, s),
            "qcqp_fwd": lambda L=L, ws=ws, wsb=wsb: L.dqq_qcqp_fwd_f64(p(P), p(q), p(ln), p(mu), p(x), B, N, 1e-7, 1e-7, 1000, 1, 0, None, p(pd), p(fl), p(ws), wsb, s),
            "qcqp_bwd": lambda L=L, ws=ws, wsb=wsb, rep=rep: L.dqq_qcqp_bwd_f64(p(P), p(q), p(ln), p(mu), p(x), p(gx), p(gP), p(gq), p(gl), p(gm), None, None, B, N, 1e-10, 0, None, p(pd), p(fl), *rep, p(ws), wsb, s),
        }
        res[li] = {k: [] for k in calls}
        res[li]["_calls"] = calls
    for rep in range(5):
        for li in (0, 1):
            for name, fn in res[li]["_calls"].items():
                if name.endswith("bwd"):
                    res[li]["_calls"][name.replace("bwd", "fwd")]()   # x and the hand-off of THIS family
                for _ in range(5): assert fn() == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100): fn()
                e1.record(); torch.cuda.synchronize()
                res[li][name].append(e0.elapsed_time(e1) * 10.0)   # us per call
    for name in ("qp_fwd", "qp_bwd", "qcqp_fwd", "qcqp_bwd"):
        a, b = sorted(res[0][name])[2], sorted(res[1][name])[2]
        print("%-9s A %7.2f us   B %7.2f us   B/A %.3f   (A runs %s | B runs %s)" % (name, a, b, b / a, ["%.1f" % v for v in res[0][name]], ["%.1f" % v for v in res[1][name]]))
    print("sum       A %7.2f us   B %7.2f us" % (sum(sorted(res[0][n])[2] for n in ("qp_fwd", "qp_bwd", "qcqp_fwd", "qcqp_bwd")),
                                                 sum(sorted(res[1][n])[2] for n in ("qp_fwd", "qp_bwd", "qcqp_fwd", "qcqp_bwd"))))


if __name__ == "__main__":
    main()
