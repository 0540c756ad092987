"""world_size-2 (and 3) gloo tests of the batch-sharding layer on CPU tensors:
shard -> per-rank solve -> all-gather must reproduce the single-process result.
The per-rank solve here is the oracle (test infrastructure); on the GPU box the
same `parallel` functions wrap the HIP ops over RCCL (bench.py --gpus N)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, q_out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import make_problem
    from diffqcqp_amd import parallel
    from oracle import oracle as O
    d = make_problem("qp", B, 8, 99)
    lo, hi = parallel.shard_bounds(B, rank, world)

    def solve(P, q):
        x, _ = O.qp_fwd_batch(P.numpy(), q.numpy(), 1e-7, 1000)
        return torch.from_numpy(x)

    x_full = parallel.solve_sharded(solve, (d["P"], d["q"]), B)
    x_local = parallel.solve_sharded(solve, (d["P"], d["q"]), B, gather=False)
    assert x_local.shape[0] == hi - lo
    assert torch.equal(x_full[lo:hi], x_local)
    assert torch.equal(parallel.shard(d["q"]), d["q"][lo:hi])
    if rank == 0:
        q_out.put(x_full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _worker_n32(rank, world, port, B, q_out):
    """BASELINE configs[3] shapes (N=32 diagonal-P QP, forward + backward): x and grad_q are gathered, grad_P stays
    sharded with P."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import make_problem
    from diffqcqp_amd import parallel
    from oracle import oracle as O
    d = make_problem("qp", B, 32, 1004)
    lo, hi = parallel.shard_bounds(B, rank, world)
    P, q, g = (parallel.shard(d[k]) for k in ("P", "q", "grad_x"))
    x, _ = O.qp_fwd_batch(P.numpy(), q.numpy(), 1e-7, 1000)
    gP, gq, _ = O.qp_bwd_batch(P.numpy(), q.numpy(), x, g.numpy())
    x_full = parallel.gather_batch(torch.from_numpy(x), B)
    gq_full, work = parallel.gather_batch(torch.from_numpy(gq), B, async_op=True)
    if work is not None:
        work.wait()
    assert gP.shape == (hi - lo, 32, 32)                       # grad_P: this rank's slice only
    assert torch.equal(x_full[lo:hi], torch.from_numpy(x))
    q_out.put((rank, x_full.numpy() if rank == 0 else None, gq_full.numpy() if rank == 0 else None, gP))
    dist.barrier()
    dist.destroy_process_group()


def _worker_async_out(rank, world, port, B, q_out):
    """gather_batch with caller-owned result and staging buffers, asynchronously, equal and ragged shards (VERDICT r3 #6,
    #9: no allocation per call; the ragged path must honour async_op -- the collective is in flight when the call returns,
    the result is complete after work.wait())."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffqcqp_amd import parallel
    lo, hi = parallel.shard_bounds(B, rank, world)
    full = torch.arange(B * 6, dtype=torch.float64).view(B, 3, 2)
    out = torch.full((B, 3, 2), -1.0, dtype=torch.float64)
    rows = parallel.gather_scratch_rows(B, world)
    assert (rows == 0) == (B % world == 0)
    scratch = torch.empty((rows, 3, 2), dtype=torch.float64) if rows else None
    ok = True
    for rep in range(3):   # the same buffers again and again, as a hot loop would
        x_local = (full[lo:hi] + rep).clone()
        res, work = parallel.gather_batch(x_local, B, async_op=True, out=out, scratch=scratch)
        ok &= res is out and work is not None and hasattr(work, "wait") and hasattr(work, "is_completed")
        work.wait()
        ok &= bool(torch.equal(out, full + rep))
        ok &= bool(work.wait())          # idempotent
    # a caller that POLLS instead of waiting: once is_completed() says so, `out` holds the gathered batch -- on the
    # ragged path too, where the trim of the padded blocks has to have run by then
    import time
    out.fill_(-1.0)
    res, work = parallel.gather_batch((full[lo:hi] + 7).clone(), B, async_op=True, out=out, scratch=scratch)
    t0 = time.time()
    while not work.is_completed() and time.time() - t0 < 60:
        time.sleep(0.001)
    ok &= bool(work.is_completed()) and bool(torch.equal(out, full + 7))
    # a scratch of another dtype is refused up front, not inside the collective
    if rows:
        try:
            parallel.gather_batch(full[lo:hi].clone(), B, out=out, scratch=scratch.float())
            ok = False
        except AssertionError:
            pass
    res = parallel.gather_batch(full[lo:hi].clone(), B, out=out, scratch=scratch)   # synchronous, same buffers
    ok &= res is out and bool(torch.equal(out, full))
    q_out.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 12), (2, 13), (3, 10)])
def test_gather_batch_caller_buffers_async(world, B):
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + world * 13 + B
    procs = [ctx.Process(target=_worker_async_out, args=(r, world, port, B, q_out)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q_out.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok in got), got


@pytest.mark.parametrize("world,B", [(2, 40), (2, 33)])
def test_config4_shapes_forward_backward_sharded(oracle, world, B):
    from conftest import make_problem
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world * 11 + B
    procs = [ctx.Process(target=_worker_n32, args=(r, world, port, B, q_out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q_out.get(timeout=180) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    d = make_problem("qp", B, 32, 1004)
    x, _ = oracle.qp_fwd_batch(d["P"].numpy(), d["q"].numpy(), 1e-7, 1000)
    gP, gq, _ = oracle.qp_bwd_batch(d["P"].numpy(), d["q"].numpy(), x, d["grad_x"].numpy())
    assert np.array_equal(got[0][1], x) and np.array_equal(got[0][2], gq)
    assert np.array_equal(np.concatenate([t[3] for t in got], axis=0), gP)


@pytest.mark.parametrize("world,B", [(2, 64), (2, 37), (3, 10)])
def test_shard_solve_gather_matches_single_process(oracle, world, B):
    from conftest import make_problem
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world * 7 + B
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q_out)) for r in range(world)]
    for p in procs:
        p.start()
    got = q_out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    d = make_problem("qp", B, 8, 99)
    ref, _ = oracle.qp_fwd_batch(d["P"].numpy(), d["q"].numpy(), 1e-7, 1000)
    assert np.array_equal(got, ref)


def test_shard_bounds_cover_batch():
    from diffqcqp_amd.parallel import shard_bounds
    for B in (0, 1, 7, 8, 65536, 262144 + 3):
        for world in (1, 2, 4, 8):
            edges = [shard_bounds(B, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == B
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
