"""-m gpu: the C ABI never allocates, so a forward + backward pair can be captured into a HIP graph and replayed
(VERDICT r2 #11); the scratch of the global-memory kernels is the caller's (dqq_scratch_bytes); the workspace cache of
the torch layer is reused, not re-allocated, from call to call."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import make_problem, knob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from diffqcqp_amd import build, ops as _ops, _capi
    build.build()
    _capi.lib()
    return _ops


def _run(ops, kind, t, layout):
    if kind == "qp":
        x = ops.qp_forward(t["P"], t["q"], 1e-7, 1000, layout=layout, out=t["x"])
        ops.qp_backward(t["P"], t["q"], x, t["grad_x"], layout=layout, out=(t["gP"], t["gq"]))
    else:
        x = ops.qcqp_forward(t["P"], t["q"], t["l_n"], t["mu"], 1e-7, 1000, layout=layout, out=t["x"])
        ops.qcqp_backward(t["P"], t["q"], t["l_n"], t["mu"], x, t["grad_x"], layout=layout,
                          out=(t["gP"], t["gq"], t["gl"], t["gm"]))


@pytest.mark.parametrize("kind,N,B,structure,layout", [
    ("qp", 8, 4096, "diag", 0), ("qcqp", 8, 4096, "diag", 0),      # fast path + (empty) work-list launch
    ("qcqp", 8, 2051, "mixed", 0),                                # work-list in use
    ("qp", 64, 96, "dense", 0), ("qp", 40, 64, "dense", 1),       # register-resident kernels (round 2: hipMallocAsync for 48 < N < 64)
    ("qp", 56, 64, "dense", 1),
    ("qp", 70, 24, "dense", 1), ("qcqp", 48, 32, "dense", 0),     # global-memory kernels on the caller's scratch
])
def test_forward_backward_captured_into_a_graph_and_replayed(oracle, ops, kind, N, B, structure, layout):
    d1, d2 = make_problem(kind, B, N, 8100 + N, structure), make_problem(kind, B, N, 8200 + N, structure)
    keys = [k for k in ("P", "q", "grad_x", "l_n", "mu") if k in d1]
    t = {k: d1[k].cuda().clone() for k in keys}
    nc = N // 2
    e = lambda *shape: torch.empty(*shape, device="cuda", dtype=torch.float64)
    t.update(x=e(B, N, 1), gP=e(B, N, N), gq=e(B, N, 1), gl=e(B, nc, 1), gm=e(B, nc, 1))
    outs = ("x", "gP", "gq") + (("gl", "gm") if kind == "qcqp" else ())
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):          # warm-up on the capture stream: workspace allocation, function attributes
        _run(ops, kind, t, layout)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    eager1 = {k: t[k].clone() for k in outs}
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        _run(ops, kind, t, layout)
    for k in outs:
        t[k].zero_()
    graph.replay()
    torch.cuda.synchronize()
    for k in outs:
        assert torch.equal(t[k], eager1[k]), "replay differs from the eager call (%s)" % k
    for k in keys:                      # new inputs in the captured buffers
        t[k].copy_(d2[k])
    graph.replay()
    torch.cuda.synchronize()
    replayed = {k: t[k].clone() for k in outs}
    with torch.cuda.stream(s):
        _run(ops, kind, t, layout)
    torch.cuda.synchronize()
    for k in outs:
        assert torch.equal(t[k], replayed[k]), "second replay differs from the eager call on the new inputs (%s)" % k
    if kind == "qp":
        xo, _ = oracle.qp_fwd_batch(d2["P"].numpy(), d2["q"].numpy(), 1e-7, 1000, nthreads=8)
    else:
        xo, _ = oracle.qcqp_fwd_batch(d2["P"].numpy(), d2["q"].numpy(), d2["l_n"].numpy(), d2["mu"].numpy(), 1e-7, 1000,
                                      nthreads=8)
    assert np.abs(replayed["x"].cpu().numpy() - xo).max() <= 1e-6


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
def test_a_captured_call_ignores_the_feedback_word(ops, kind):
    """the caller-side hints (dqq_hint_flags, _capi.py): the word may move a call to another lane layout or kernel -- never inside a stream capture, where the
    route must be the one the arguments determine (a graph is replayed on batches the word knows nothing about)."""
    from diffqcqp_amd import _capi
    N, B = 8, 57344 + 2048
    d = make_problem(kind, B, N, 8300, "dense")
    keys = [k for k in ("P", "q", "grad_x", "l_n", "mu") if k in d]
    t = {k: d[k].cuda().clone() for k in keys}
    e = lambda *shape: torch.empty(*shape, device="cuda", dtype=torch.float64)
    t.update(x=e(B, N, 1), gP=e(B, N, N), gq=e(B, N, 1), gl=e(B, N // 2, 1), gm=e(B, N // 2, 1))
    outs = ("x", "gP", "gq") + (("gl", "gm") if kind == "qcqp" else ())
    slot = (0 if kind == "qp" else 1) * 4 + N // 2 - 1
    was_on = _capi._feedback is not None
    _capi.enable_feedback(True)
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(4):                      # the word learns: all non-diagonal, three times running
                _run(ops, kind, t, 0)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        assert _capi.feedback_words()[slot] == (B, B) and _capi.feedback_streaks()[slot] >= 2
        eager = {k: t[k].clone() for k in outs}
        for name in ("fwd_feedback_routes", "bwd_whole_batches", "lane_list_drains"):
            knob(name, 0)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            _run(ops, kind, t, 0)
        assert [_capi.get_option(n) for n in ("fwd_feedback_routes", "bwd_whole_batches", "lane_list_drains")] == [0, 0, 0]
        for k in outs:
            t[k].zero_()
        graph.replay()
        torch.cuda.synchronize()
        for k in outs:
            assert torch.equal(t[k], eager[k])      # the hinted routes and the captured ones: the same bits
        with torch.cuda.stream(s):
            _run(ops, kind, t, 0)                   # outside the capture the word is followed again
        torch.cuda.synchronize()
        assert _capi.get_option("fwd_feedback_routes") == 1 and _capi.get_option("bwd_whole_batches") == 1
    finally:
        _capi.enable_feedback(was_on)


@pytest.mark.parametrize("kind", ["qp", "qcqp"])
def test_a_graph_captured_with_explicit_hint_flags_replays_the_same_bits_on_any_data(ops, kind):
    """VERDICT r5 #3 (ii).  A captured graph gets no hints from the report word -- but a caller who KNOWS what the graph will
    be replayed on may say so: `layout = P_AUTO | F_EXPECT_DENSE` is an argument, and arguments are captured.  The flag
    selects routes of identical results: the graph captured on dense data replays the un-flagged eager call's bits on dense
    data AND on diagonal data put into the captured buffers afterwards (there the flag only costs time)."""
    from diffqcqp_amd import _capi
    N, B = 8, 57344 + 1024
    dd, dg = make_problem(kind, B, N, 8400, "dense"), make_problem(kind, B, N, 8401, "diag")
    keys = [k for k in ("P", "q", "grad_x", "l_n", "mu") if k in dd]
    t = {k: dd[k].cuda().clone() for k in keys}
    e = lambda *shape: torch.empty(*shape, device="cuda", dtype=torch.float64)
    t.update(x=e(B, N, 1), gP=e(B, N, N), gq=e(B, N, 1), gl=e(B, N // 2, 1), gm=e(B, N // 2, 1))
    outs = ("x", "gP", "gq") + (("gl", "gm") if kind == "qcqp" else ())
    flagged = _capi.P_AUTO | _capi.F_EXPECT_DENSE
    was_on = _capi._feedback is not None
    _capi.enable_feedback(False)                    # nothing below comes from a report word
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            _run(ops, kind, t, 0)                   # the argument-determined routes, eager: the reference bits (dense data)
        torch.cuda.synchronize()
        eager_dense = {k: t[k].clone() for k in outs}
        for name in ("fwd_feedback_routes", "bwd_whole_batches", "lane_list_drains"):
            knob(name, 0)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            _run(ops, kind, t, flagged)
        # the capture holds the hinted routes: one lane per problem forward, whole-batch lane-per-problem backward
        assert _capi.get_option("fwd_feedback_routes") == 1 and _capi.get_option("bwd_whole_batches") == 1
        for k in outs:
            t[k].zero_()
        graph.replay()
        torch.cuda.synchronize()
        for k in outs:
            assert torch.equal(t[k], eager_dense[k]), "flagged graph differs from the un-flagged eager call on dense data (%s)" % k
        for k in keys:                              # diagonal data in the captured buffers: the flag is wrong now
            t[k].copy_(dg[k].cuda())
        graph.replay()
        torch.cuda.synchronize()
        replayed = {k: t[k].clone() for k in outs}
        with torch.cuda.stream(s):
            _run(ops, kind, t, 0)
        torch.cuda.synchronize()
        for k in outs:
            assert torch.equal(t[k], replayed[k]), "flagged graph differs from the un-flagged eager call on diagonal data (%s)" % k
    finally:
        _capi.enable_feedback(was_on)


def test_output_buffers_of_the_wrong_type_are_refused(ops):
    d = make_problem("qp", 16, 8, 8500)
    P, q = d["P"].cuda(), d["q"].cuda()
    with pytest.raises(ValueError):
        ops.qp_forward(P, q, 1e-7, 1000, out=torch.empty(16, 8, 1, device="cuda", dtype=torch.float32))
    with pytest.raises(ValueError):
        ops.qp_backward(P, q, q, q, out=(torch.empty(16, 8, 8, device="cuda", dtype=torch.float64),
                                         torch.empty(16, 8, device="cuda", dtype=torch.float64)))


def test_scratch_is_the_callers(ops):
    from diffqcqp_amd import _capi
    L = _capi.lib()
    for kind in (0, 1, 2, 3):
        for pas in (0, 1):
            for N in (2, 8, 21, 32, 64):
                if (kind, pas) == (2, 1) and N > 21:
                    assert L.dqq_scratch_bytes(kind, pas, N, 1000, 0) > 0
                else:   # QCQP backward up to N = 64 included: the register-resident kernels use no scratch
                    assert L.dqq_scratch_bytes(kind, pas, N, 1000, 0) == 0, (kind, pas, N)
    # dqq_scratch_bytes follows the route: with the reference-order flag in p_layout, the QCQP backward of 42 < N <= 64
    # reaches the global-memory kernel and the call demands (and uses) its scratch
    REF = _capi.F_REFERENCE_ORDER
    assert L.dqq_max_n(2, 0) == 64 and L.dqq_max_n(2, REF) == 42
    assert L.dqq_scratch_bytes(1, 1, 64, 1000, REF) > 0 and L.dqq_scratch_bytes(1, 1, 42, 1000, REF) == 0
    assert ops.workspace_bytes(1000, 1, 1, 64, REF) == L.dqq_workspace_bytes(1000) + L.dqq_scratch_bytes(1, 1, 64, 1000, REF)
    assert ops.workspace_bytes(1000, 1, 1, 64) == L.dqq_workspace_bytes(1000)
    need = L.dqq_scratch_bytes(0, 0, 70, 24, 0)
    assert need > 0 and L.dqq_scratch_bytes(0, 0, 70, 24, 0) == need and L.dqq_scratch_bytes(0, 0, 70, 0, 0) == 0
    d = make_problem("qp", 24, 70, 8300, "dense")
    P, q = d["P"].cuda(), d["q"].cuda()
    x = torch.empty(24, 70, 1, device="cuda", dtype=torch.float64)
    stream = torch.cuda.current_stream().cuda_stream
    small = torch.zeros(L.dqq_workspace_bytes(24) // 4, dtype=torch.int32, device="cuda")   # work-list only
    for layout in (0, 1):
        rc = L.dqq_qp_fwd_f64(P.data_ptr(), q.data_ptr(), x.data_ptr(), 24, 70, 1e-7, 1e-7, 1000, 1, layout, None, None,
                              None, small.data_ptr(), small.numel() * 4, stream)
        assert rc == -5, "a workspace without the scratch must be refused (DQQ_E_WORKSPACE), got %d" % rc
    rc = L.dqq_qp_fwd_f64(P.data_ptr(), q.data_ptr(), x.data_ptr(), 24, 70, 1e-7, 1e-7, 1000, 1, 1, None, None, None,
                          None, 0, stream)
    assert rc == -5
    big = torch.zeros((L.dqq_workspace_bytes(24) + need) // 4, dtype=torch.int32, device="cuda")
    rc = L.dqq_qp_fwd_f64(P.data_ptr(), q.data_ptr(), x.data_ptr(), 24, 70, 1e-7, 1e-7, 1000, 1, 1, None, None, None,
                          big.data_ptr(), big.numel() * 4, stream)
    torch.cuda.synchronize()
    assert rc == 0 and torch.isfinite(x).all()
    # DQQ_P_DENSE at a size the register kernels hold still needs no workspace at all
    d8 = make_problem("qp", 100, 8, 8301, "dense")
    x8 = torch.empty(100, 8, 1, device="cuda", dtype=torch.float64)
    rc = L.dqq_qp_fwd_f64(d8["P"].cuda().data_ptr(), d8["q"].cuda().data_ptr(), x8.data_ptr(), 100, 8, 1e-7, 1e-7, 1000, 1,
                          1, None, None, None, None, 0, stream)
    torch.cuda.synchronize()
    assert rc == 0


def test_workspace_cache_is_reused_and_bounded(ops):
    """`numel < B + 64` re-allocated (and zero-filled) the workspace on every call for 94 % of the
    batch sizes; and one tensor per stream was kept for ever."""
    dev = torch.device("cuda", 0)
    for B in (1, 60, 61, 2051, 4096, 65536):
        a = ops._workspace(dev, B)
        b = ops._workspace(dev, B)
        assert a.data_ptr() == b.data_ptr(), "workspace re-allocated for the same batch size B=%d" % B
    big = ops._workspace(dev, 65536)
    assert ops._workspace(dev, 100).data_ptr() == big.data_ptr()      # a larger one serves smaller batches
    streams = [torch.cuda.Stream() for _ in range(ops._MAX_WORKSPACES + 6)]
    for st in streams:
        ops._workspace(dev, 128, st.cuda_stream)
    assert len(ops._workspaces) <= ops._MAX_WORKSPACES
    d = make_problem("qp", 2051, 8, 8302)
    P, q = d["P"].cuda(), d["q"].cuda()
    x1 = ops.qp_forward(P, q, 1e-7, 1000)
    p1 = ops._workspace(dev, 2051).data_ptr()
    x2 = ops.qp_forward(P, q, 1e-7, 1000)
    assert ops._workspace(dev, 2051).data_ptr() == p1 and torch.equal(x1, x2)


def test_workspace_lifetime_under_graph_capture(ops):
    """a captured graph bakes in the workspace pointer.  (1) A caller-owned `workspace=` is used as given.
    (2) A cached workspace that was handed out during a capture is never replaced in place: when a larger call arrives
    later the cache moves on to a new tensor and the captured one stays alive, so a replay cannot scribble over memory
    the allocator has given to someone else.  (3) A first call inside a capture is refused (it would allocate + memset)."""
    dev = torch.device("cuda", 0)
    d = make_problem("qcqp", 300, 8, 8700, "mixed")
    P, q, ln, mu = (d[k].cuda() for k in ("P", "q", "l_n", "mu"))
    own = ops.make_workspace(dev, 300, 1, 0, 8)
    x_own = ops.qcqp_forward(P, q, ln, mu, 1e-7, 1000, workspace=own)
    x_ref = ops.qcqp_forward(P, q, ln, mu, 1e-7, 1000)
    torch.cuda.synchronize()
    assert torch.equal(x_own, x_ref) and int(own[:2].abs().sum()) == 0
    with pytest.raises(ValueError):
        ops.qcqp_forward(P, q, ln, mu, 1e-7, 1000, workspace=own[:8])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    x = torch.empty(300, 8, 1, device="cuda", dtype=torch.float64)
    fresh = torch.cuda.Stream()
    graph0 = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="stream capture"):
        with torch.cuda.graph(graph0, stream=fresh):
            ops.qcqp_forward(P, q, ln, mu, 1e-7, 1000, out=x)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        ops.qcqp_forward(P, q, ln, mu, 1e-7, 1000, out=x)      # warm-up: the cache entry of this stream
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        ops.qcqp_forward(P, q, ln, mu, 1e-7, 1000, out=x)
    captured = ops._workspaces[(0, s.cuda_stream)]
    assert any(w is captured for w in ops._pinned)
    with torch.cuda.stream(s):                                  # a larger batch on the same stream: the cache grows ...
        dbig = make_problem("qp", 200000, 8, 8701)
        ops.qp_forward(dbig["P"].cuda(), dbig["q"].cuda(), 1e-7, 1000)
    torch.cuda.synchronize()
    assert ops._workspaces[(0, s.cuda_stream)] is not captured   # ... into a NEW tensor
    assert any(w is captured for w in ops._pinned)               # the captured one is still alive
    x.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(x, x_ref) and int(captured[:2].abs().sum()) == 0


def test_functions_take_the_layout_from_the_module_default(oracle, ops):
    """QPFn2 / QCQPFn2 keep the reference's signatures (qcqp.py:24, 144); a caller who knows P is dense selects the
    general kernels through the module-level default."""
    from diffqcqp_amd import qcqp
    d = make_problem("qcqp", 512, 8, 8400, "dense")
    args = [d[k].cuda() for k in ("P", "q", "l_n", "mu")]
    ws = torch.zeros(512, 8, 1, device="cuda", dtype=torch.float64)
    res = {}
    assert qcqp.get_default_layout() == "auto"
    for lay in ("auto", "dense", "auto_expect_dense"):
        prev = qcqp.set_default_layout(lay)
        try:
            a = [t.clone().requires_grad_(True) for t in args]
            x = qcqp.QCQPFn2.apply(*a, ws, 1e-7, 1000)
            (x * d["grad_x"].cuda()).sum().backward()
            res[lay] = [x.detach()] + [t.grad for t in a]
        finally:
            qcqp.set_default_layout(prev)
    assert qcqp.get_default_layout() == "auto"
    for u, v in zip(res["auto"], res["dense"]):
        assert torch.allclose(u, v, rtol=1e-9, atol=1e-12)
    for u, v in zip(res["auto"], res["auto_expect_dense"]):     # the caller's explicit hint flag: routes of identical results
        assert torch.equal(u, v)
    xo, _ = oracle.qcqp_fwd_batch(d["P"].numpy(), d["q"].numpy(), d["l_n"].numpy(), d["mu"].numpy(), 1e-7, 1000, nthreads=8)
    assert np.abs(res["dense"][0].cpu().numpy() - xo).max() <= 1e-6
    with pytest.raises(ValueError):
        qcqp.set_default_layout("diag")


def test_ctypes_and_pybind11_bindings_agree(ops):
    """The C ABI has two Python faces (diffqcqp_amd/_capi.py): the pybind11 module `_dqq` and ctypes.  Same symbols,
    same arguments, same numbers; the pybind11 call is the cheaper one (it matters at B = 1)."""
    import time
    from diffqcqp_amd import _capi
    d = make_problem("qcqp", 777, 8, 8600)
    P, q, ln, mu = (d[k].cuda() for k in ("P", "q", "l_n", "mu"))
    ws = ops._workspace(torch.device("cuda", 0), 777)
    stream = torch.cuda.current_stream().cuda_stream
    out, cost = {}, {}
    for name, L in (("ctypes", _capi.ctypes_lib()), ("pybind11", _capi.pybind_lib())):
        assert L is not None, name
        x = torch.empty(777, 8, 1, device="cuda", dtype=torch.float64)
        it = torch.empty(777, device="cuda", dtype=torch.int32)
        args = (P.data_ptr(), q.data_ptr(), ln.data_ptr(), mu.data_ptr(), x.data_ptr(), 777, 8, 1e-7, 1e-7, 1000, 1, 0,
                it.data_ptr(), None, None, ws.data_ptr(), ws.numel() * 4, stream)
        assert L.dqq_qcqp_fwd_f64(*args) == 0
        torch.cuda.synchronize()
        out[name] = (x.clone(), it.clone())
        bad = (P.data_ptr(), q.data_ptr(), ln.data_ptr(), mu.data_ptr(), x.data_ptr(), -1, 8, 1e-7, 1e-7, 1000, 1, 0,
               None, None, None, None, 0, None)
        t0 = time.perf_counter()
        for _ in range(2000):
            L.dqq_qcqp_fwd_f64(*bad)          # argument check only: returns DQQ_E_BAD_SIZE before any launch
        cost[name] = (time.perf_counter() - t0) / 2000
        assert L.dqq_qcqp_fwd_f64(*bad) == -2
    assert torch.equal(out["ctypes"][0], out["pybind11"][0]) and torch.equal(out["ctypes"][1], out["pybind11"][1])
    assert _capi.binding() in ("pybind11", "ctypes")
    print("call overhead: ctypes %.2f us, pybind11 %.2f us" % (cost["ctypes"] * 1e6, cost["pybind11"] * 1e6))
    assert cost["pybind11"] < cost["ctypes"]
