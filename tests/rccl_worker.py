"""Worker of tests/test_gpu_rccl.py: one rank of a torch.distributed.run launch (one process per GPU, backend "nccl" =
RCCL).  argv: B, verdict path.  Every rank solves the WHOLE batch on its own GPU first (the single-GPU answer), then its
slice through parallel.solve_sharded / gather_batch over the HIP ops; rank 0 writes the verdict after an all-reduce of the
per-rank comparisons (a mismatch on ANY rank fails)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    B, out_path = int(sys.argv[1]), sys.argv[2]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from conftest import make_problem
    from diffqcqp_amd import ops, parallel
    ok = {}
    lo, hi = parallel.shard_bounds(B, rank, world)

    # ---- QP, N = 32 (the configs[3] shape), diagonal P in the dense layout
    d = {k: v.to(dev) for k, v in make_problem("qp", B, 32, 1004).items()}
    x1 = ops.qp_forward(d["P"], d["q"], 1e-7, 1000)
    gP1, gq1 = ops.qp_backward(d["P"], d["q"], x1, d["grad_x"])
    x_all = parallel.solve_sharded(lambda P, q: ops.qp_forward(P.contiguous(), q.contiguous(), 1e-7, 1000),
                                   (d["P"], d["q"]), B)
    ok["qp_x_equal"] = bool(torch.equal(x_all, x1))
    sl = slice(lo, hi)
    gP, gq = ops.qp_backward(d["P"][sl].contiguous(), d["q"][sl].contiguous(), x_all[sl].contiguous(),
                             d["grad_x"][sl].contiguous())
    ok["qp_grad_P_equal"] = bool(torch.equal(gP, gP1[sl]))                       # grad_P stays sharded with P
    ok["qp_grad_q_equal"] = bool(torch.equal(parallel.gather_batch(gq, B), gq1))  # (grad_q is cheap to gather)

    # ---- QCQP, N = 8
    d = {k: v.to(dev) for k, v in make_problem("qcqp", B, 8, 1003).items()}
    x1 = ops.qcqp_forward(d["P"], d["q"], d["l_n"], d["mu"], 1e-7, 1000)
    g1 = ops.qcqp_backward(d["P"], d["q"], d["l_n"], d["mu"], x1, d["grad_x"])
    x_all = parallel.solve_sharded(
        lambda P, q, ln, mu: ops.qcqp_forward(P.contiguous(), q.contiguous(), ln.contiguous(), mu.contiguous(), 1e-7, 1000),
        (d["P"], d["q"], d["l_n"], d["mu"]), B)
    ok["qcqp_x_equal"] = bool(torch.equal(x_all, x1))
    gs = ops.qcqp_backward(*(d[k][sl].contiguous() for k in ("P", "q", "l_n", "mu")), x_all[sl].contiguous(),
                           d["grad_x"][sl].contiguous())
    ok["qcqp_grads_equal"] = all(bool(torch.equal(a, b[sl])) for a, b in zip(gs, g1))

    # ---- the asynchronous (possibly ragged) gather into caller-owned buffers, as bench.py's step issues it
    rows = parallel.gather_scratch_rows(B, world)
    out = torch.empty_like(x1)
    scratch = torch.empty((rows,) + tuple(x1.shape[1:]), dtype=x1.dtype, device=dev) if rows else None
    res, work = parallel.gather_batch(x1[sl].contiguous(), B, async_op=True, out=out, scratch=scratch)
    work.wait()
    torch.cuda.synchronize()
    ok["async_ragged_equal"] = bool(torch.equal(res, x1))

    flags = torch.tensor([1 if ok[k] else 0 for k in sorted(ok)], dtype=torch.int32, device=dev)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    if rank == 0:
        v = {k: bool(f) for k, f in zip(sorted(ok), flags.tolist())}
        v.update({"world": world, "backend": dist.get_backend(), "B": B})
        with open(out_path, "w") as f:
            json.dump(v, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
