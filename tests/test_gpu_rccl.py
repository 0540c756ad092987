"""Two ranks over RCCL on a box with two visible GPUs (-m gpu): lights up by itself, skips on a one-GPU box.

SURVEY.md 8(e): contiguous batch split, every rank solves its slice with the single-GPU HIP kernels, ONE all-gather of x,
grad_P stays sharded with P.  The round's GPU boxes have one GPU, so until an N > 1 box runs this file the multi-GPU path
is covered by the gloo tests (tests/test_parallel_gloo.py) and by the RCCL branch with one rank
(tests/test_gpu_bench_line.py); nothing here needs editing the day two GPUs are visible.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible GPUs (RCCL over xGMI)")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_worker.py")


def _torchrun(nproc, port, *cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                           "--master-addr", "127.0.0.1", "--master-port", str(port), *cmd],
                          capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


@two_gpus
def test_bench_line_on_two_ranks(tmp_path):
    """The driver's own scaling command at N = 2: one short JSON line, rccl_world 2, the headline workload on every rank;
    the gather and strong-scaling figures are scalars of the line, their sub-records are in the details file."""
    det = str(tmp_path / "details.json")
    r = _torchrun(2, 29611, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--repeats", "2",
                  "--no-cpu-baseline", "--details", det)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1 and len(lines[0].encode()) <= 6000, r.stdout[-2000:]
    d, full = json.loads(lines[0]), json.load(open(det))
    assert d["n_gpus"] == 2 and d["config"]["rccl_world"] == 2 and d["scaling"] == "weak"
    assert d["config"]["B_total"] == 2 * 131072
    assert d["config"]["with_gather_ms_per_step"] > 0 and d["config"]["strong_cfg4_ms_per_step"] > 0
    assert full["with_gather"]["rccl_world"] == 2 and full["with_gather"]["ms_per_step"] > 0
    assert full["strong_config4"]["without_gather"]["rccl_world"] == 2
    assert d["value"] > 0 and abs(d["value"] - d["config"]["B_total"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


@two_gpus
@pytest.mark.parametrize("B", [4096, 4097])
def test_solve_sharded_over_two_ranks_equals_one_gpu_bit_for_bit(B, tmp_path):
    """parallel.solve_sharded over the HIP ops on two ranks (equal and ragged split): the gathered x equals the single-GPU
    x bit for bit on both ranks; the grad_P / grad_q shards concatenate to the single-GPU gradients bit for bit (QP N = 32,
    the configs[3] shape, and QCQP N = 8)."""
    out = str(tmp_path / "verdict.json")
    r = _torchrun(2, 29612 + (B % 2), WORKER, str(B), out)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    v = json.load(open(out))
    assert v["world"] == 2 and v["backend"] == "nccl"
    for k in ("qp_x_equal", "qp_grad_P_equal", "qp_grad_q_equal", "qcqp_x_equal", "qcqp_grads_equal", "async_ragged_equal"):
        assert v[k] is True, (k, v)


def test_the_worker_itself_with_one_rank(tmp_path):
    """The same worker under torch.distributed.run with ONE rank (RCCL initialised, every collective the identity): runs
    on the one-GPU boxes of this round, so that first contact with two GPUs is not the worker's first execution."""
    out = str(tmp_path / "verdict.json")
    r = _torchrun(1, 29615, WORKER, "1025", out)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    v = json.load(open(out))
    assert v["world"] == 1 and v["backend"] == "nccl"
    for k in ("qp_x_equal", "qp_grad_P_equal", "qp_grad_q_equal", "qcqp_x_equal", "qcqp_grads_equal", "async_ragged_equal"):
        assert v[k] is True, (k, v)
