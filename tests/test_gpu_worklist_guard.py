"""Work-list hygiene (csrc/launch.h): a DQQ_P_AUTO call on a workspace whose header was POISONED either
repairs it or reports it -- it never reads or writes out of bounds, never "solves" a problem that does not exist and never
silently leaves one unsolved.  Every poisoned call is followed by a second call on the same workspace, which must be clean.

Header layout (csrc/launch.h): [0] count, [1] exit ticket, [2] pick-up, [12] dirty, [32 + 32 g] sub-tickets, [1056 + 32 g]
segment counts (N >= 32), [2080 + 32 g] segment pick-ups, entries from [3104]; capacity 32 * (B / 32 + 512) slots (plain list) /
B / 32 + 512 per segment."""
import numpy as np
import pytest
import torch

from conftest import make_problem

pytestmark = pytest.mark.gpu

COUNT, TICKET, NEXT, DIRTY, SUB, SEGC, SEGN, ENTRIES = 0, 1, 2, 12, 32, 1056, 2080, 3104


@pytest.fixture(scope="module")
def ops():
    from diffqcqp_amd import build, ops as _ops, _capi
    build.build()
    _capi.lib()
    return _ops


def _inputs(kind, B, N, seed):
    """The first quarter of the batch has a dense P, the rest a diagonal one: most tiles never touch the work-list."""
    d, dd = make_problem(kind, B, N, seed, "diag"), make_problem(kind, B, N, seed, "dense")
    d["P"][: B // 4] = dd["P"][: B // 4]
    return {k: v.contiguous().cuda() for k, v in d.items()}


def _run(ops, kind, g, ws, x=None):
    """forward (x is None) or backward through DQQ_P_AUTO on the caller's workspace -> list of output tensors
    (the optional diagnostics with them: iteration counts / refinement steps -- int tensors, -1 on a problem that was not solved)"""
    if x is None:
        if kind == "qp":
            return list(ops.qp_forward(g["P"], g["q"], 1e-7, 1000, workspace=ws, return_iters=True))
        return list(ops.qcqp_forward(g["P"], g["q"], g["l_n"], g["mu"], 1e-7, 1000, workspace=ws, return_iters=True))
    if kind == "qp":
        return list(ops.qp_backward(g["P"], g["q"], x, g["grad_x"], workspace=ws, return_steps=True))
    return list(ops.qcqp_backward(g["P"], g["q"], g["l_n"], g["mu"], x, g["grad_x"], workspace=ws, return_steps=True))


def _header_is_idle(ws):
    h = ws[:ENTRIES].cpu().numpy().copy()
    h[DIRTY] = 0
    h[4:12] = 0     # (the feedback shadow words and the per-problem note are not part of the protocol's invariant)
    return not h.any()


def _poisons(B, N):
    cap = 32 * (B // 32 + 512)
    seg = N >= 32

    def tickets(ws):
        ws[TICKET], ws[NEXT] = 3, 7
        for g in range(32):
            ws[SUB + 32 * g], ws[SEGN + 32 * g] = 5, 9

    def stale_entries(ws):
        if seg:
            ws[COUNT] = 1
            ws[SEGC + 32 * 3] = 4
            base = ENTRIES + 3 * (B // 32 + 512)
        else:
            ws[COUNT] = 4
            base = ENTRIES
        ws[base: base + 4] = torch.tensor([B + 10, -3, 2 ** 30, 1], dtype=torch.int32, device=ws.device)

    def overflowing_count(ws):
        if seg:
            ws[COUNT] = 1
            for g in range(32):
                ws[SEGC + 32 * g] = B // 32 + 512 + 5
        else:
            ws[COUNT] = cap + 5

    def all_ones(ws):
        ws[:ENTRIES] = -1

    def garbage(ws):
        gen = torch.Generator().manual_seed(7)
        ws[:ENTRIES] = torch.randint(-2 ** 31, 2 ** 31 - 1, (ENTRIES,), generator=gen, dtype=torch.int64).to(torch.int32).cuda()

    return {"tickets": (tickets, False), "stale_entries": (stale_entries, True), "overflowing_count": (overflowing_count, True),
            "all_ones": (all_ones, True), "garbage": (garbage, True)}


@pytest.mark.parametrize("poison", ["tickets", "stale_entries", "overflowing_count", "all_ones", "garbage"])
@pytest.mark.parametrize("kind,N,B,pas", [("qp", 32, 3000, "fwd"), ("qcqp", 32, 3000, "bwd"), ("qp", 16, 5000, "fwd"),
                                          ("qcqp", 8, 40000, "bwd"), ("qp", 8, 9000, "bwd"), ("qp", 64, 700, "bwd")])
def test_a_poisoned_header_is_repaired_or_reported(ops, kind, N, B, pas, poison):
    from diffqcqp_amd import _capi
    g = _inputs(kind, B, N, 5100 + N)
    clean = ops.make_workspace(g["q"].device, B, 1 if kind == "qcqp" else 0, 1 if pas == "bwd" else 0, N)
    x = _run(ops, kind, g, clean)[0]
    ref = _run(ops, kind, g, clean, x if pas == "bwd" else None)
    torch.cuda.synchronize()
    assert not _capi.workspace_status(clean) and _header_is_idle(clean)

    ws = ops.make_workspace(g["q"].device, B, 1 if kind == "qcqp" else 0, 1 if pas == "bwd" else 0, N)
    fn, must_be_dirty = _poisons(B, N)[poison]
    fn(ws)
    torch.cuda.synchronize()
    out = _run(ops, kind, g, ws, x if pas == "bwd" else None)
    torch.cuda.synchronize()
    dirty = _capi.workspace_status(ws)
    assert dirty or not must_be_dirty, "an inconsistent header went unnoticed"
    unsolved = torch.zeros(B, dtype=torch.bool, device=x.device)
    ints_flagged = torch.zeros(B, dtype=torch.bool, device=x.device)
    for o, r in zip(out, ref):
        if o is None:
            continue
        o2, r2 = o.reshape(B, -1), r.reshape(B, -1)
        if not o.is_floating_point():     # iteration counts / refinement steps: the clean call's, or the sentinel -1
            flagged = (o2 == -1).any(dim=1)
            assert bool(((o2 == r2).all(dim=1) | flagged).float().mean() > 0.99), (poison, "a diagnostic is neither the clean call's nor -1")
            assert bool((o2[~flagged] > 0).all()), (poison, "garbage in a diagnostic of a solved problem")
            ints_flagged = flagged
            continue
        nan = torch.isnan(o2).any(dim=1)
        unsolved |= nan
        # every problem: the clean call's answer (to rounding: a problem queued once more is solved by the general kernel),
        # or NaN -- reported, never silently wrong
        good = ((o2 - r2).abs() <= 1e-9 * (1.0 + r2.abs())).all(dim=1)
        assert bool((good | nan).all()), (poison, "a problem is neither solved nor reported")
        assert bool((torch.isnan(o2) | (o2 == r2))[~nan].float().mean() > 0.99)   # and bit-identical on (nearly) all of them
    if not dirty:
        assert not bool(unsolved.any()), "NaN outputs without the dirty word"
    #  the diagnostics of a problem reported as failed are sentinels, never what torch.empty left there
    assert bool((ints_flagged == unsolved).all()), (poison, "sentinel -1 exactly on the problems whose outputs are NaN")
    assert bool((~unsolved).float().mean() > 0.5)      # the diagonal tiles never depend on the list
    # every word of the protocol is sane again (the drain re-zeroed it, or the fast kernel / the clamp repaired it): the next
    # call on the SAME workspace, without a reset, is clean.  (Words outside the protocol keep what the poison wrote.)
    if poison not in ("all_ones", "garbage"):
        assert _header_is_idle(ws), poison
    out2 = _run(ops, kind, g, ws, x if pas == "bwd" else None)
    torch.cuda.synchronize()
    for o, r in zip(out2, ref):
        if o is not None:
            assert torch.equal(o, r), (poison, "the call after the poisoned one")
    assert _capi.workspace_status(ws) == dirty      # sticky until the reset
    _capi.workspace_reset(ws)
    torch.cuda.synchronize()
    assert not _capi.workspace_status(ws) and _header_is_idle(ws)


def test_reset_and_status_round_trip(ops):
    from diffqcqp_amd import _capi
    ws = ops.make_workspace(torch.device("cuda", 0), 1000)
    assert not _capi.workspace_status(ws)
    ws[DIRTY] = 1
    ws[COUNT] = 17
    assert _capi.workspace_status(ws)
    _capi.workspace_reset(ws)
    torch.cuda.synchronize()
    assert not _capi.workspace_status(ws) and int(ws[COUNT]) == 0
