#!/usr/bin/env python3
"""Where does a short timed region of the headline step lose time against a long one?  (the driver
runs `--steps 20 --warmup 5`, the builder's profiles `--steps 100`.)  For K in a list: R regions of exactly K steps,
bracketed like bench.py's; per K the median ms/step, the host's enqueue time (perf_counter around the K x 4 C-ABI
calls, no synchronise) and the fit  T(K) = F + s*K.   Usage: probe_region_overhead.py [--threads] [--streams 2]"""
import argparse
import json
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="5,10,20,50,100,200")
    ap.add_argument("--repeats", type=int, default=9)
    ap.add_argument("--threads", action="store_true", help="enqueue each stream's chain from its own host thread")
    ap.add_argument("--balance", action="store_true", help="alternate the chains between the two streams step by step")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from diffqcqp_amd import build
    build.build()
    chains = [bench.Chain("qp", 65536, 8, "diag", True, dev, 1000), bench.Chain("qcqp", 65536, 8, "diag", True, dev, 1031)]
    main_s, side = torch.cuda.current_stream(), torch.cuda.Stream()
    streams = [main_s.cuda_stream, side.cuda_stream]
    ctr = [0]

    def step():
        a, b = (streams[0], streams[1]) if not (args.balance and ctr[0] & 1) else (streams[1], streams[0])
        ctr[0] += 1
        chains[0].launch(0, a)
        chains[1].launch(0, b)
        chains[0].launch(1, a)
        chains[1].launch(1, b)

    def chain_k(c, s, k):
        for _ in range(k):
            c.launch(0, s)
            c.launch(1, s)

    def region(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if args.threads:
            th = threading.Thread(target=chain_k, args=(chains[1], streams[1], k))
            th.start()
            chain_k(chains[0], streams[0], k)
            th.join()
        else:
            for _ in range(k):
                step()
        t1 = time.perf_counter()
        side.synchronize()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return t2 - t0, t1 - t0

    for _ in range(30):
        step()
    torch.cuda.synchronize()
    ks = [int(k) for k in args.ks.split(",")]
    rows = []
    for k in ks:
        rs = sorted(region(k) for _ in range(args.repeats))
        tot, enq = rs[len(rs) // 2]
        rows.append({"K": k, "region_us": tot * 1e6, "us_per_step": tot / k * 1e6, "host_enqueue_us_per_step": enq / k * 1e6,
                     "min_us_per_step": rs[0][0] / k * 1e6})
    A = np.array([[1.0, r["K"]] for r in rows])
    y = np.array([r["region_us"] for r in rows])
    F, s = np.linalg.lstsq(A, y, rcond=None)[0]
    out = {"rows": rows, "fit_fixed_us": F, "fit_us_per_step": s, "threads": args.threads, "balance": args.balance,
           "env": {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG", "HSA_ENABLE_INTERRUPT")}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
