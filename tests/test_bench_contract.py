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


def _canned_full(b, extra_cfg=0, junk=False):
    """A full record as measure() + main() assemble it for the default one-GPU run (values of a round-5 run), optionally
    with more scalars than the line may carry and with hostile values (NaN, inf, very long strings, nested junk)."""
    big = "x" * 5000 if junk else "B=65536 N=8 diagonal-P QP forward+backward and QCQP forward+backward; eps=1e-7"
    nan = float("nan") if junk else 0.0566
    cfg = {"workload": big, "baseline_config": "2'+3 (headline)", "B_total": 131072, "B_this_rank": [65536, 65536],
           "N": [8, 8], "buffers": "step k works on set k mod 5", "p_layout": big, "launch": big, "sharding": big, "rccl_world": 1}
    for k in b.CONFIG_KEYS:
        cfg.setdefault(k, 0.123456789012345)
    for i in range(extra_cfg):
        cfg["extra_scalar_%d" % i] = 1.0 / (i + 1)
    rl = {"bound": "hbm", "kernel": "qcqp_fwd", "achieved": 1631.123456789, "peak": 8000.0, "unit": "GB/s",
          "frac": 0.2038904320987, "traffic": 50638868.6, "traffic_source": big, "kernel_us_qcqp_fwd": 28.2812345,
          "fp64": {"nested": {"deep": [1, 2, 3]}}, "live_pmc": {"a": {"b": 1}}, "timing": big}
    for k in b.ROOFLINE_KEYS:
        rl.setdefault(k, float("inf") if junk else 0.33333333333)
    for i in range(extra_cfg):
        rl["more_%d" % i] = float(i)
    return {"metric": "QP+QCQP solves/sec (fwd+bwd)", "value": 2.315e9, "unit": "solves/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": nan, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": cfg, "roofline": rl,
            "cpu_baseline": {"value": 7.1e6, "unit": "solves/s", "cores": 128, "kind": "port", "sample": big,
                             "single_thread_value": 1.8e5, "single_thread_sample": big, "python_loop_value": 2.7e4,
                             "python_loop_sample": big, "reference_published": big},
            "kernels": {"qp_fwd": {"mean_us": 23.3}}, "per_config": {"config_%d" % i: {"junk": [big] * 3} for i in range(2, 6)},
            "environment": {"rocm_smi": {big[:50]: big}}}


def _strict(line):
    def fail(c):
        raise AssertionError("non-finite constant %s in the line" % c)
    return json.loads(line, parse_constant=fail)


def test_contract_line_is_one_short_strict_json_line():
    """round 5's 24 KB line was unreadable to the driver.  Whatever the full record holds, the ONE stdout line
    is at most 6000 bytes (it must also fit the driver's 8081-character tail whole), strict JSON (no NaN / Infinity), not
    nested below config / roofline / cpu_baseline, `config` = workload + <= 20 scalars, `roofline` <= 24 scalars."""
    b = _bench()
    assert b.LINE_LIMIT == 6000
    for extra, junk in ((0, False), (200, False), (200, True)):
        line = b.contract_line(_canned_full(b, extra, junk))
        assert "\n" not in line and len(line.encode()) <= 6000, len(line)
        d = _strict(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in d, k
        assert "workload" in d["config"] and len(d["config"]) <= 21 and len(d["roofline"]) <= 24 and len(d["cpu_baseline"]) <= 7
        for sub in ("config", "roofline", "cpu_baseline"):
            assert not any(isinstance(v, (dict, list)) for v in d[sub].values()), sub
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in d["roofline"]
        assert set(d["cpu_baseline"]) <= {"value", "unit", "cores", "kind", "sample", "single_thread_value", "python_loop_value"}
        assert d["details"] == "bench_details.json"
    # the verdict's named scalars are inside the kept ones
    d = _strict(b.contract_line(_canned_full(b)))
    for k in ("cold_ms_per_step", "cold_value", "hot_ms_per_step", "hot_value", "qp_pair_moved_frac", "qp_pair_large_moved_frac",
              "qp_pair_solves_per_s", "qp_pair_hot_ms_per_step", "traffic_over_algorithmic", "valu_busy_frac", "kernel_us"):
        assert k in d["roofline"], k
    for k in ("cfg2_ms_per_step", "cfg3_ms_per_step", "cfg4_ms_per_step", "cfg4_moved_frac", "cfg5_ms_per_step",
              "dense8_auto_no_hint_ms_per_step", "dense8_dense_ms_per_step", "ref_figure_qp_fwd_ms", "qcqp_grad_exit_flip_rate"):
        assert k in d["config"], k
    assert d["value"] == 2.315e9 and d["roofline"]["kernel_us"] == 28.2812   # (6 significant digits below the top level)


def test_contract_line_of_a_committed_full_record():
    """The same on a full record a GPU run of the final tree wrote (tests/golden/bench_details_sample.json)."""
    b = _bench()
    path = os.path.join(ROOT, "tests", "golden", "bench_details_sample.json")
    if not os.path.exists(path):
        import pytest
        pytest.skip("no committed sample")
    full = json.load(open(path))
    line = b.contract_line(full)
    assert len(line.encode()) <= 6000
    d = _strict(line)
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
    assert abs(d["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5
    assert d["roofline"]["qp_pair_large_moved_frac"] > 0.4          # north_star's 40 % where the chip is filled
    assert all(v is None or not isinstance(v, float) or v <= 1.0 for k, v in d["roofline"].items() if k.endswith("_frac")
               and "algorithmic" not in k)


def test_emit_writes_the_details_file_and_keeps_stderr_unlike_a_contract_line(tmp_path, capfd):
    b = _bench()
    full = _canned_full(b, 50, True)
    r, w = os.pipe()
    b.emit(full, w, str(tmp_path / "d.json"))
    os.close(w)
    line = os.read(r, 1 << 16).decode()
    os.close(r)
    assert line.endswith("\n") and line.count("\n") == 1 and len(line) <= 6001
    _strict(line)
    det = _strict(open(tmp_path / "d.json").read())
    assert "per_config" in det and "environment" in det and det["ms_per_step"] is None
    err = capfd.readouterr().err
    assert err.startswith("[bench details] ") and not any(l.startswith("{") for l in err.splitlines())


def test_inputs_follow_the_surveys_recipe():
    """float64 inputs generated on the CPU with torch.Generator().manual_seed(1000 + cfg), then copied to the
    device -- cfg 2: p ~ U(0.1, 1.1) -> diag_embed, q ~ U(-1, 1); cfg 3: + l_n, mu ~ U(0, 1), grad_l ~ N(0, 1)."""
    import torch
    b = _bench()
    assert [b.workload_seed(0, 0), b.workload_seed(0, 1)] == [1002, 1003]
    assert [b.workload_seed(c, 0) for c in (2, 3, 4, 5, 8, 9, 10)] == [1002, 1003, 1004, 1005, 1002, 1002, 1004]
    c = b.Chain("qcqp", 16, 8, "diag", True, torch.device("cpu"), b.workload_seed(3, 0))
    g = torch.Generator().manual_seed(1003)
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    t = c.sets[0]
    assert torch.equal(t["P"], torch.diag_embed(r(16, 8) + 0.1)) and torch.equal(t["q"], 2 * r(16, 8, 1) - 1)
    assert torch.equal(t["l_n"], r(16, 4, 1)) and torch.equal(t["mu"], r(16, 4, 1))
    assert torch.equal(t["g"], torch.randn(16, 8, 1, generator=g, dtype=torch.float64))
    c.add_sets(2)                       # rotating buffers: distinct data
    assert len(c.sets) == 3 and not torch.equal(c.sets[1]["q"], t["q"]) and not torch.equal(c.sets[2]["q"], c.sets[1]["q"])
    d = b.Chain("qp", 4, 64, "dense", False, torch.device("cpu"), b.workload_seed(5, 0)).sets[0]["P"]
    S = torch.rand(4, 64, 64, generator=torch.Generator().manual_seed(1005), dtype=torch.float64)
    assert torch.allclose(d, torch.bmm(S, S.transpose(1, 2)) / 64 + 0.1 * torch.eye(64, dtype=torch.float64), rtol=0, atol=1e-14)


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
