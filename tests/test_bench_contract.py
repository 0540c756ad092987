"""bench.py's bookkeeping, checked without a GPU: the algorithmic bytes per problem are SURVEY.md 8(d)'s, the
workloads are BASELINE.json's configs, and the driver's command line parses (the run itself needs the GPU)."""
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_are_the_surveys():
    b = _bench()
    # QP fwd N^2 w + 2 N w, bwd 2 N^2 w + 4 N w; QCQP + 2 nc w / + 4 nc w
    assert b.algo_bytes("qp", 8, "fwd") == 640 and b.algo_bytes("qp", 8, "bwd") == 1280
    assert b.algo_bytes("qcqp", 8, "fwd") == 704 and b.algo_bytes("qcqp", 8, "bwd") == 1408
    assert b.algo_bytes("qp", 32, "fwd") == 8704 and b.algo_bytes("qp", 32, "bwd") == 17408
    assert b.algo_bytes("qp", 64, "fwd") == 33792 and b.algo_bytes("qp", 64, "bwd") == 67584
    # the diagonal hand-off moves fewer bytes than the algorithmic figure in the backward, a few more in the forward
    assert b.moved_bytes("qp", 8, "bwd", True) == 1280 - 512 + 64 + 1
    assert b.moved_bytes("qp", 8, "fwd", True) == 640 + 64 + 1
    assert b.moved_bytes("qcqp", 8, "bwd", False) == 1408


def test_workloads_are_the_baseline_configs():
    b = _bench()
    cfg = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert len(cfg) == 5
    w = b.WORKLOADS
    assert w[2][1] == [("qp", 8, "diag", False)] and w[2][2] == 65536            # configs[1]
    assert w[3][1] == [("qcqp", 8, "diag", True)] and w[3][2] == 65536           # configs[2]
    assert w[4][1] == [("qp", 32, "diag", True)] and w[4][2] == 262144 and w[4][3] == "strong"   # configs[3]
    assert w[5][1] == [("qp", 64, "dense", True)] and w[5][2] == 65536           # configs[4]
    assert w[0][1] == [("qp", 8, "diag", True), ("qcqp", 8, "diag", True)]       # the headline: configs[1] + bwd, configs[2]
    assert b.HBM_PEAK_GBS == 8000.0 and (b.EPS, b.MAX_ITER, b.MU_PROX) == (1e-7, 1000, 1e-7)


def test_bench_refuses_to_run_without_a_gpu_and_never_falls_back():
    """There is no CPU path: on a box without a GPU the bench must stop with a message, not print a number."""
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs the GPU" in (r.stderr + r.stdout)
    assert not any(line.startswith("{") for line in r.stdout.splitlines())
