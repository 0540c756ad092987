"""Batch sharding over the GPUs of one node (one process per GPU, RCCL over xGMI).

The problems of a batch are independent (nothing in the ADMM solve or in the
backward couples two problems), so the path shards with NO data-path
collective: rank r owns the contiguous slice `shard_slice(B, r, world)` of every
(B, ...) tensor, solves it with the single-GPU kernels, and keeps its slice of
every gradient (grad_P stays sharded with P: it is N^2 doubles per problem).
The one collective is optional and comes after the solve: `gather_batch`
all-gathers the (B_local, N, 1) solution so every rank holds the full x.

`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on the
"gloo" backend with CPU tensors (tests/test_parallel_gloo.py).
"""
import torch
import torch.distributed as dist


def shard_bounds(B, rank, world):
    """[lo, hi) of rank's contiguous slice; the first B % world ranks get one extra problem."""
    base, rem = divmod(int(B), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_slice(B, rank, world):
    lo, hi = shard_bounds(B, rank, world)
    return slice(lo, hi)


def shard(t, rank=None, world=None):
    """This rank's slice of a full-batch tensor (a view)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return t[shard_slice(t.shape[0], rank, world)]


def gather_batch(x_local, B_total, group=None, async_op=False):
    """All-gather per-rank slices (B_r, ...) into the full (B_total, ...) tensor on every rank.

    Equal shards use one all_gather_into_tensor (a single RCCL all-gather);
    ragged shards are padded to the largest shard and trimmed afterwards.
    Returns the tensor, or (tensor, work) when async_op is set.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(B_total, r, world)[1] - shard_bounds(B_total, r, world)[0] for r in range(world)]
    assert x_local.shape[0] == sizes[rank], "local shard has %d rows, expected %d" % (x_local.shape[0], sizes[rank])
    tail = tuple(x_local.shape[1:])
    x_local = x_local.contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((B_total,) + tail, dtype=x_local.dtype, device=x_local.device)
        work = dist.all_gather_into_tensor(out, x_local, group=group, async_op=async_op)
        return (out, work) if async_op else out
    m = max(sizes)
    padded = torch.zeros((m,) + tail, dtype=x_local.dtype, device=x_local.device)
    padded[: sizes[rank]] = x_local
    buf = torch.empty((world * m,) + tail, dtype=x_local.dtype, device=x_local.device)
    work = dist.all_gather_into_tensor(buf, padded, group=group, async_op=async_op)
    if async_op:
        work.wait()
    out = torch.cat([buf[r * m: r * m + sizes[r]] for r in range(world)], dim=0)
    return (out, None) if async_op else out


def solve_sharded(solve_fn, full_inputs, B_total, gather=True, group=None):
    """Run `solve_fn(*local_inputs) -> x_local` on this rank's slice of each full-batch
    input and (optionally) all-gather the result.  `full_inputs` are tensors whose
    first dimension is the batch; ranks that already hold only their slice should
    call `solve_fn` + `gather_batch` directly."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    sl = shard_slice(B_total, rank, world)
    x_local = solve_fn(*[t[sl] for t in full_inputs])
    if not gather:
        return x_local
    return gather_batch(x_local, B_total, group=group)
