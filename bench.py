#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched QP/QCQP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--graph] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): QP+QCQP solves/sec (fwd+bwd) per GPU; achieved HBM GB/s
vs roofline.  One STEP is one pass of the hot path over one batch of synthetic
input per GPU:
    QP   forward + backward on B=65536, N=8, diagonal P in the (B,8,8) layout   (configs[1] + its backward)
    QCQP forward + backward on the same P, q plus l_n, mu                       (configs[2])
= 2*B forward+backward solves per GPU per step.  Inputs are resident in HBM when
the timed region starts; outputs are written to preallocated device buffers.  The
two families are independent problems: by default their launch chains go to two
HIP streams (QP fwd -> QP bwd | QCQP fwd -> QCQP bwd), so the HBM-bound backward
of one overlaps the FP64-VALU-bound forward of the other (--streams 1: one stream;
its rate is reported as `single_stream`).
N > 1: one process per GPU, every rank solves its own shard of B problems (weak
scaling, no data-path collective); the single collective is the final RCCL
all-gather of the last solution x, inside the timed region.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      dominant kernel: algorithmic bytes per launch / its mean duration (HIP events) vs 8 TB/s
  cpu_baseline  the oracle (our C port of the reference algorithm) on this box's host cores
  kernels       per-launch breakdown (mean microseconds, algorithmic GB/s) of the four launches of a step
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_PER_GPU = 65536
N = 8
EPS, MAX_ITER, MU_PROX = 1e-7, 1000, 1e-7
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md chip table (6290 GB/s measured copy)
# algorithmic bytes per problem, SURVEY.md 8(d), N=8, float64, dense (B,N,N) P layout
ALGO_BYTES = {"qp_fwd": 640, "qp_bwd": 1280, "qcqp_fwd": 704, "qcqp_bwd": 1408}


def make_inputs(rank, dev):
    """p~U(0.1,1.1) -> diag_embed, q~U(-1,1), l_n,mu~U(0,1), grad~N(0,1); CPU generator."""
    g = torch.Generator().manual_seed(1002 + 7919 * rank)
    f64 = torch.float64
    p = torch.rand(B_PER_GPU, N, generator=g, dtype=f64) + 0.1
    t = {
        "P": torch.diag_embed(p),
        "q": 2 * torch.rand(B_PER_GPU, N, 1, generator=g, dtype=f64) - 1,
        "l_n": torch.rand(B_PER_GPU, N // 2, 1, generator=g, dtype=f64),
        "mu": torch.rand(B_PER_GPU, N // 2, 1, generator=g, dtype=f64),
        "g_qp": torch.randn(B_PER_GPU, N, 1, generator=g, dtype=f64),
        "g_qcqp": torch.randn(B_PER_GPU, N, 1, generator=g, dtype=f64),
    }
    return t, {k: v.to(dev).contiguous() for k, v in t.items()}


class Plan:
    """The four launches of a step with every pointer resolved once (no allocation in the loop)."""

    def __init__(self, d, dev):
        from diffqcqp_amd import _capi, ops
        self.lib = _capi.lib()
        self.d = d
        e = lambda *s: torch.empty(s, dtype=torch.float64, device=dev)
        B = B_PER_GPU
        self.x_qp, self.x_qcqp = e(B, N, 1), e(B, N, 1)
        self.gP_qp, self.gq_qp = e(B, N, N), e(B, N, 1)
        self.gP_qc, self.gq_qc, self.gl_qc, self.gm_qc = e(B, N, N), e(B, N, 1), e(B, N // 2, 1), e(B, N // 2, 1)
        self.cache_qp, self.cache_qc = ops.diag_cache(d["q"]), ops.diag_cache(d["q"])
        self.ws = ops._workspace(dev, B)
        self.wsb = self.ws.numel() * 4
        self.names = ["qp_fwd", "qp_bwd", "qcqp_fwd", "qcqp_bwd"]

    def launch(self, which, stream):
        d, L, p = self.d, self.lib, (lambda t: t.data_ptr())
        B = B_PER_GPU
        if which == 0:
            rc = L.dqq_qp_fwd_f64(p(d["P"]), p(d["q"]), p(self.x_qp), B, N, EPS, MU_PROX, MAX_ITER, 1, 0, None,
                                  p(self.cache_qp[0]), p(self.cache_qp[1]), p(self.ws), self.wsb, stream)
        elif which == 1:
            rc = L.dqq_qp_bwd_f64(p(d["P"]), p(d["q"]), p(self.x_qp), p(d["g_qp"]), p(self.gP_qp), p(self.gq_qp), B, N,
                                  1e-10, 0, None, p(self.cache_qp[0]), p(self.cache_qp[1]), p(self.ws), self.wsb, stream)
        elif which == 2:
            rc = L.dqq_qcqp_fwd_f64(p(d["P"]), p(d["q"]), p(d["l_n"]), p(d["mu"]), p(self.x_qcqp), B, N, EPS, MU_PROX,
                                    MAX_ITER, 1, 0, None, p(self.cache_qc[0]), p(self.cache_qc[1]), p(self.ws), self.wsb,
                                    stream)
        else:
            rc = L.dqq_qcqp_bwd_f64(p(d["P"]), p(d["q"]), p(d["l_n"]), p(d["mu"]), p(self.x_qcqp), p(d["g_qcqp"]),
                                    p(self.gP_qc), p(self.gq_qc), p(self.gl_qc), p(self.gm_qc), None, None, B, N, 1e-10,
                                    0, None, p(self.cache_qc[0]), p(self.cache_qc[1]), p(self.ws), self.wsb, stream)
        if rc != 0:
            raise RuntimeError("launch %s failed with %d" % (self.names[which], rc))

    def step(self, stream):
        for w in range(4):
            self.launch(w, stream)

    def step2(self, stream_a, stream_b):
        """The QP chain on stream_a and the (independent) QCQP chain on stream_b, each with its own work-list."""
        ws = self.ws
        self.launch(0, stream_a)
        self.ws = self.ws_side
        self.launch(2, stream_b)
        self.ws = ws
        self.launch(1, stream_a)
        self.ws = self.ws_side
        self.launch(3, stream_b)
        self.ws = ws


def check_against_oracle(plan, host, nsample=2048):
    """Parity spot-check of what was just timed (rank 0): HIP vs oracle on the first nsample problems."""
    from oracle import oracle as O
    s = slice(0, nsample)
    h = {k: v[s].numpy() for k, v in host.items()}
    xo, _ = O.qp_fwd_batch(h["P"], h["q"], EPS, MAX_ITER, MU_PROX, nthreads=O.max_threads())
    xq, _ = O.qcqp_fwd_batch(h["P"], h["q"], h["l_n"], h["mu"], EPS, MAX_ITER, MU_PROX, nthreads=O.max_threads())
    nt = O.max_threads()
    gq = O.qp_bwd_batch(h["P"], h["q"], xo, h["g_qp"], nthreads=nt)[1]
    ref = O.qcqp_bwd_batch(h["P"], h["q"], h["l_n"], h["mu"], xq, h["g_qcqp"], nthreads=nt)
    # QCQP backward: the reference's refinement exit (1 or 3 Tikhonov steps) is decided by rounding noise
    # (Solver.cpp:32-41), so end to end it is compared where the exits agree; with identical x it is bit-exact.
    from diffqcqp_amd import ops
    dv = plan.d
    st = ops.qcqp_backward(dv["P"][s], dv["q"][s], dv["l_n"][s], dv["mu"][s], plan.x_qcqp[s], dv["g_qcqp"][s],
                           need=(False, True, False, False), return_steps=True)[-1].cpu().numpy()
    same = st == ref[-1]
    same_x = ops.qcqp_backward(dv["P"][s], dv["q"][s], dv["l_n"][s], dv["mu"][s], torch.from_numpy(xq).to(dv["q"].device),
                               dv["g_qcqp"][s], need=(False, True, False, False))[1].cpu().numpy()
    gqc = ref[1]
    dqc = (plan.gq_qc[s].cpu().numpy() - gqc)
    scale = np.maximum(1.0, np.abs(gqc).max(axis=(1, 2)))
    err = {
        "x_qp": float((plan.x_qp[s].cpu() - torch.from_numpy(xo)).abs().max()),
        "x_qcqp": float((plan.x_qcqp[s].cpu() - torch.from_numpy(xq)).abs().max()),
        "grad_q_qp": float((plan.gq_qp[s].cpu() - torch.from_numpy(gq)).abs().max()),
        "grad_q_qcqp_same_x_bit_exact": bool(np.array_equal(same_x, gqc)),
        "grad_q_qcqp_rel_where_refinement_exit_agrees": float((np.abs(dqc).max(axis=(1, 2)) / scale)[same].max()),
        "qcqp_refinement_exit_flip_rate": float(1.0 - same.mean()),
    }
    return err


def cpu_baseline(host):
    """The oracle timed on the host cores over the SAME workload (one full step = 2*B fwd+bwd solves)."""
    from oracle import oracle as O
    h = {k: v.numpy() for k, v in host.items()}

    def one_pass(nt, nb):
        s = slice(0, nb)
        t0 = time.perf_counter()
        x, _ = O.qp_fwd_batch(h["P"][s], h["q"][s], EPS, MAX_ITER, MU_PROX, nthreads=nt)
        O.qp_bwd_batch(h["P"][s], h["q"][s], x, h["g_qp"][s], nthreads=nt)
        xq, _ = O.qcqp_fwd_batch(h["P"][s], h["q"][s], h["l_n"][s], h["mu"][s], EPS, MAX_ITER, MU_PROX, nthreads=nt)
        O.qcqp_bwd_batch(h["P"][s], h["q"][s], h["l_n"][s], h["mu"][s], xq, h["g_qcqp"][s], nthreads=nt)
        return 2 * nb / (time.perf_counter() - t0)

    cores = O.max_threads()
    one_pass(cores, 4096)  # spin up the OpenMP team
    best_all = max(one_pass(cores, B_PER_GPU) for _ in range(3))
    one_t = one_pass(1, 16384)
    return {
        "value": best_all, "unit": "solves/s", "cores": cores, "kind": "port",
        "sample": "oracle/diffqcqp_oracle.c (dense C port of the reference algorithm, OpenMP over the batch), "
                  "best of 3 passes over the full step workload (B=65536 QP + B=65536 QCQP, fwd+bwd)",
        "single_thread_value": one_t,
        "single_thread_sample": "same, 1 thread, first 16384 problems of each family",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph")
    ap.add_argument("--streams", type=int, default=2, choices=(1, 2),
                    help="2 (default): the QP chain and the QCQP chain of a step -- independent problems -- are enqueued "
                         "on two HIP streams, so the HBM-bound backward of one overlaps the VALU-bound forward of the "
                         "other; 1: everything on one stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()

    # Everything the native libraries print on stdout (RCCL prints its version banner there) goes to
    # stderr; the one JSON line is written to the real stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs the GPU (the hot path is a HIP kernel; there is no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    # DQQ_BENCH_FORCE_DIST=1 exercises the RCCL path (init, barrier, all-gather) with a single rank
    use_dist = world > 1 or os.environ.get("DQQ_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL

    from diffqcqp_amd import build, _capi, parallel
    if rank == 0:
        build.build()
    if use_dist:
        dist.barrier()
    _capi.lib()

    host, d = make_inputs(rank, dev)
    plan = Plan(d, dev)
    stream = torch.cuda.current_stream()
    sh = stream.cuda_stream

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (also warms RCCL's all-gather)
    for _ in range(max(args.warmup, 1)):
        plan.step(sh)
    if use_dist:
        parallel.gather_batch(plan.x_qcqp, B_PER_GPU * world)
    graph = None
    if args.graph:
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            plan.step(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for _ in range(3):
            graph.replay()

    side = None
    if args.streams == 2:
        from diffqcqp_amd import ops as _ops
        side = torch.cuda.Stream()
        plan.ws_side = _ops._workspace(dev, B_PER_GPU, side.cuda_stream)  # one work-list per stream
        for _ in range(3):
            plan.step2(sh, side.cuda_stream)
        torch.cuda.synchronize()

    # ---- timed region: EXACTLY K steps (+ the final gather when sharded)
    barrier()
    t0 = time.perf_counter()
    if graph is not None:
        for _ in range(args.steps):
            graph.replay()
    elif side is not None:
        for _ in range(args.steps):
            plan.step2(sh, side.cuda_stream)
        side.synchronize()
    else:
        for _ in range(args.steps):
            plan.step(sh)
    gather_ms = None
    if use_dist:
        torch.cuda.synchronize()
        tg = time.perf_counter()
        x_all = parallel.gather_batch(plan.x_qcqp, B_PER_GPU * world)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert x_all.shape[0] == B_PER_GPU * world
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- context (outside the contract's timed region): the same K steps strictly on one stream
    single_ms = None
    if side is not None and graph is None:
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            plan.step(sh)
        barrier()
        single_ms = (time.perf_counter() - t1) / args.steps * 1e3

    # ---- roofline pass: HIP events around every launch of the step, on the launch stream.  The dense
    # fallback launch of the AUTO layout is switched off here so that each bracket holds exactly one kernel
    # (the inputs are diagonal by construction, so the fallback kernel is an empty launch anyway).
    _capi.set_option("auto_fallback", 0)
    nrep = min(max(args.steps, 20), 200)
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
          for _ in range(nrep)]
    torch.cuda.synchronize()
    for r in range(nrep):
        for w in range(4):
            ev[r][w][0].record(stream)
            plan.launch(w, sh)
            ev[r][w][1].record(stream)
    torch.cuda.synchronize()
    _capi.set_option("auto_fallback", 1)
    kernels = {}
    for w, name in enumerate(plan.names):
        ts = sorted(ev[r][w][0].elapsed_time(ev[r][w][1]) for r in range(nrep))
        mean_ms = sum(ts) / len(ts)
        kernels[name] = {
            "mean_us": mean_ms * 1e3, "median_us": ts[len(ts) // 2] * 1e3,
            "algo_bytes_per_launch": ALGO_BYTES[name] * B_PER_GPU,
            "algo_GBps": ALGO_BYTES[name] * B_PER_GPU / (mean_ms * 1e-3) / 1e9,
        }
    dom = max(kernels, key=lambda k: kernels[k]["mean_us"])
    traffic, valu = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path)).get(dom, {})
            traffic = pmc.get("hbm_bytes_per_launch")
            if "SQ_INSTS_VALU" in pmc:
                # second ceiling of the same kernel: FP64 VALU issue.  Wave-level VALU instructions of one launch
                # (SQ_INSTS_VALU, rocprofv3 PMC pass) spread over the chip's 1024 SIMDs, at the measured issue
                # floor of a wave64 fp64 instruction (tools/ubench/fp64_issue.hip: 2.08 ns; 4 cycles at 2.4 GHz
                # would be 1.67 ns).  Almost every VALU instruction of this kernel is fp64.
                floor_us = pmc["SQ_INSTS_VALU"] / 1024.0 * 2.08e-3
                valu = {"valu_insts_per_launch": pmc["SQ_INSTS_VALU"], "valu_insts_per_wave": pmc.get("valu_insts_per_wave"),
                        "issue_floor_ns_per_wave_inst": 2.08, "floor_us": floor_us,
                        "frac": floor_us / kernels[dom]["mean_us"]}
        except Exception:
            traffic, valu = None, None
    roofline = {
        "bound": "hbm", "kernel": dom, "achieved": kernels[dom]["algo_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": kernels[dom]["algo_GBps"] / HBM_PEAK_GBS, "traffic": traffic,
        "frac_vs_measured_copy_bw_6290": kernels[dom]["algo_GBps"] / 6290.0,
        "step_algo_GBps": sum(ALGO_BYTES.values()) * B_PER_GPU / (sum(k["mean_us"] for k in kernels.values()) * 1e-6) / 1e9,
        "step_algo_GBps_as_timed": sum(ALGO_BYTES.values()) * B_PER_GPU / (elapsed / args.steps) / 1e9,
        "fp64_valu_issue": valu,
    }

    if rank != 0:
        dist.barrier()
        dist.destroy_process_group()
        return

    solves = 2 * B_PER_GPU * world * args.steps
    out = {
        "metric": "QP+QCQP solves/sec (fwd+bwd)", "value": solves / elapsed, "unit": "solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "per GPU and step: B=65536 N=8 diagonal-P (dense (B,8,8) layout) QP forward+backward "
                        "[BASELINE configs[1] + backward] and B=65536 N=8 QCQP forward+backward [configs[2]]; "
                        "eps=1e-7 max_iter=1000 mu_prox=1e-7; value counts one forward+backward as one solve",
            "B_per_gpu": B_PER_GPU, "N": N, "p_layout": "auto (off-diagonals verified in-kernel)",
            "launch": ("hip graph replay" if graph is not None else "eager, 4 C-ABI calls per step")
                      + (", QP and QCQP chains on two streams" if args.streams == 2 else ""),
            "sharding": "batch shards, no data-path collective; final all-gather of x" if world > 1 else "single GPU",
        },
        "roofline": roofline,
        "kernels": kernels,
        "per_gpu_value": solves / elapsed / world,
    }
    if gather_ms is not None:
        out["final_allgather_ms"] = gather_ms
    if single_ms is not None:
        out["single_stream"] = {"ms_per_step": single_ms, "value_this_rank": 2 * B_PER_GPU / (single_ms * 1e-3),
                                "note": "same K steps with all four launches on one stream (what one problem family "
                                        "alone sees); not the headline"}
    if not args.no_check:
        out["parity_max_abs_err_vs_oracle_first_2048"] = check_against_oracle(plan, host)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(host)
        out["gpu_over_cpu_all_cores"] = out["value"] / out["cpu_baseline"]["value"]
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
