"""The one JSON line bench.py prints (the driver's contract), on a short run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strict(line):
    def fail(c):
        raise AssertionError("non-finite constant %s in the line" % c)
    return json.loads(line, parse_constant=fail)


def _run(*flags, env=None, tmp=None, hot_only=True):
    """-> (contract line, full record of bench_details.json)."""
    import tempfile
    det = os.path.join(tmp or tempfile.mkdtemp(prefix="dqq_bench_", dir="/tmp"), "details.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                        "--repeats", "2", "--details", det, *(("--hot-only",) if hot_only else ()), *flags],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    assert len(lines[0].encode()) <= 6000, "the contract line is at most 6000 bytes (VERDICT r5 #1)"
    assert not any(l.startswith("{") for l in r.stderr.splitlines()), "nothing on stderr looks like a contract line"
    assert "[bench details] " in r.stderr
    return _strict(lines[0]), _strict(open(det).read())


def test_headline_line_carries_the_contract_keys():
    d, full = _run("--no-per-config")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["unit"] == "solves/s" and d["higher_is_better"] is True and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = the units all ranks processed / the timed region
    assert abs(d["value"] - d["config"]["B_total"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    rl = d["roofline"]
    assert rl["bound"] in ("hbm", "mfma") and rl["unit"] == "GB/s" and rl["peak"] == 8000.0
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-5 and 0 < rl["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["value"] > 100 * cb["value"]   # (a reported baseline, not the target; but a GPU path slower than that is broken)
    assert full["value"] == d["value"] and "kernels" in full and "environment" in full


def test_a_config_line_checks_itself_against_the_oracle():
    d, full = _run("--config", "3", "--no-cpu-baseline")
    assert "configs[2]" in d["config"]["workload"]
    err = full.get("parity_max_abs_err_vs_oracle_sample")
    assert err is not None and err["qcqp"]["x"] < 1e-6
    assert d["config"]["qcqp_grad_exit_flip_rate"] == pytest.approx(err["qcqp"]["refinement_exit_flip_rate"], rel=1e-5, abs=1e-12)


def test_details_name_host_enqueue_and_every_kernel():
    """What tells a host-bound step from a GPU-bound one (host enqueue time, the one-stream step, the per-kernel durations,
    the fixed cost of a timed region) is in the full record; no fraction without `algorithmic` in its name exceeds the
    physical bound."""
    d, full = _run("--no-per-config", "--no-cpu-baseline")
    rl = full["roofline"]
    for k in ("host_enqueue_us_per_step", "single_stream_ms_per_step", "kernels_sum_us", "kernel_us_qp_fwd", "kernel_us_qp_bwd",
              "kernel_us_qcqp_fwd", "kernel_us_qcqp_bwd", "region_fixed_us", "us_per_step_steady_state",
              "ms_per_step_long_region", "step_moved_frac", "moved_frac", "step_algorithmic_frac"):
        assert isinstance(rl.get(k), float), k
    # (sanity of the values only: a 3-step run on a shared box is no place for timing relations -- the process gets
    # descheduled for tens of milliseconds now and then)
    assert 0 < rl["host_enqueue_us_per_step"] < 1e5
    assert 0 < rl["step_moved_frac"] < 0.79 and 0 < rl["moved_frac"] < 0.79
    assert d["roofline"]["kernel_us"] == pytest.approx(rl["kernel_us_" + rl["kernel"]], rel=1e-5)


def test_value_is_the_rotating_buffer_step_and_the_hot_step_is_beside_it():
    """the same 185 MB of buffers step after step sit in the 256 MiB Infinity Cache; a training loop presents
    new data every step.  Default: step k works on set k mod nsets (> 768 MiB in all) and THAT is `value`; the hot figure is
    in the line beside it."""
    d, full = _run("--no-per-config", "--no-cpu-baseline", "--no-check", hot_only=False)
    rl = d["roofline"]
    assert full["roofline"]["buffer_sets"] >= 3 and "mod" in d["config"]["buffers"]
    assert rl["cold_ms_per_step"] == pytest.approx(d["ms_per_step"], rel=1e-5) and rl["cold_value"] == pytest.approx(d["value"], rel=1e-5)
    assert rl["hot_ms_per_step"] > 0 and rl["hot_value"] > 0
    assert "hot" in full and full["hot"]["ms_per_step"] == pytest.approx(rl["hot_ms_per_step"], rel=1e-5)
    # one set: the hot step is the value, and there is no cold figure
    d1, _ = _run("--no-per-config", "--no-cpu-baseline", "--no-check")
    assert d1["roofline"]["cold_ms_per_step"] is None and d1["roofline"]["hot_ms_per_step"] == pytest.approx(d1["ms_per_step"], rel=1e-5)


def _dist_env(port):
    env = dict(os.environ, DQQ_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    return env


def test_rccl_branch_with_one_rank():
    """The code the driver's `torch.distributed.run ... bench.py --gpus N` executes -- RCCL init, barriers, the all-gather of
    x into a caller-owned buffer overlapped with the backward, MAX over ranks -- run here with one rank so that first
    contact with a multi-GPU node is not the first execution."""
    d, full = _run("--config", "4", "--no-cpu-baseline", env=_dist_env("29577"))
    assert d["config"]["rccl_world"] == 1 and d["scaling"] == "strong" and d["config"]["B_total"] == 262144
    assert full["without_gather"]["rccl_world"] == 1 and full["without_gather"]["ms_per_step"] > 0
    assert full["gather_after_backward"]["ms_per_step"] > 0
    assert full["without_gather"]["allgather_bytes_per_rank"] == 262144 * 32 * 8
    # the gather is the identity at one rank: the three rates agree within a few percent on a quiet box
    assert d["ms_per_step"] < 2.0 * full["without_gather"]["ms_per_step"]   # (generous: see the note on timing relations above)


def test_distributed_default_line_is_the_headline_with_the_gather():
    """What the driver's scaling command prints (no --config, RCCL initialised): the SAME headline workload as the N = 1 line
    -- so that value(N) / (N value(1)) is a scaling efficiency --, `with_gather` = the same step with the all-gather of both
    families' x, and configs[3] (strong scaling) as the sub-record `strong_config4`; their step times are scalars of the
    line.  One rank here, rotating buffers (the default)."""
    d, full = _run("--no-cpu-baseline", env=_dist_env("29578"), hot_only=False)
    assert d["scaling"] == "weak" and d["config"]["rccl_world"] == 1 and d["config"]["B_total"] == 131072
    assert "weak scaling" in full["config"]["sharding"]
    wg = full["with_gather"]
    assert wg["ms_per_step"] >= 0.5 * d["ms_per_step"] and wg["allgather_bytes_per_rank"] == 2 * 65536 * 8 * 8 and wg["rccl_world"] == 1
    s4 = full["strong_config4"]
    assert s4["without_gather"]["rccl_world"] == 1 and s4["gather_after_backward"]["ms_per_step"] > 0
    cf = d["config"]
    assert cf["strong_cfg4_ms_per_step"] == pytest.approx(s4["ms_per_step"], rel=1e-5)
    assert cf["with_gather_ms_per_step"] == pytest.approx(wg["ms_per_step"], rel=1e-5) and cf["with_gather_value"] > 0
    assert cf["strong_cfg4_without_gather_ms_per_step"] > 0 and cf["strong_cfg4_gather_after_backward_ms_per_step"] > 0


def test_default_line_answers_the_north_star_sentence():
    """The full default run (what the driver runs: rotating buffers, every sub-record, no rocprofv3 sub-process): the line
    carries north_star's target sentence -- N = 8 QP forward+backward solves/s on one GPU with its HBM fraction, at
    B = 65536 and at the chip-filling B = 1048576 --, the cold and the hot step, every BASELINE config's step time and the
    dense 8 x 8 figures; no fraction without `algorithmic` in its name exceeds 6.29 / 8.0 (the measured copy rate over the
    spec peak), no `*_frac` of the full record exceeds 1 unless it says `algorithmic`."""
    d, full = _run(hot_only=False)
    rl, cf = d["roofline"], d["config"]
    for k in ("frac", "traffic", "traffic_source", "qp_pair_ms_per_step", "qp_pair_solves_per_s", "qp_pair_moved_frac",
              "qp_pair_hot_ms_per_step", "qp_pair_large_ms_per_step", "qp_pair_large_moved_frac", "moved_frac",
              "step_moved_frac", "kernel_us", "cold_ms_per_step", "cold_value", "hot_ms_per_step", "hot_value"):
        assert k in rl, (k, list(rl))
    for k in ("cfg2_ms_per_step", "cfg2_moved_frac", "cfg3_ms_per_step", "cfg3_moved_frac", "cfg4_ms_per_step",
              "cfg4_moved_frac", "cfg5_ms_per_step", "cfg5_fp64_frac", "dense8_auto_ms_per_step",
              "dense8_auto_no_hint_ms_per_step", "dense8_dense_ms_per_step", "dense8_auto_no_hint_over_dense",
              "ref_figure_qp_fwd_ms", "qcqp_grad_exit_flip_rate"):
        assert k in cf, (k, list(cf))
    # the sentence itself: >= 1e6 N = 8 QP fwd+bwd solves/s (by three orders of magnitude), with a physical HBM fraction
    assert rl["qp_pair_solves_per_s"] > 1e8 and rl["qp_pair_large_solves_per_s"] > 1e8
    assert abs(rl["qp_pair_solves_per_s"] - 65536 / (rl["qp_pair_ms_per_step"] * 1e-3)) < 1e-4 * rl["qp_pair_solves_per_s"]
    assert abs(rl["qp_pair_moved_frac"] - 1538 * 65536 / (rl["qp_pair_ms_per_step"] * 1e-3) / 8e12) < 1e-4
    assert abs(full["roofline"]["qp_pair_algorithmic_frac"] / full["roofline"]["qp_pair_moved_frac"] - 1920 / 1538) < 1e-9
    assert 0 < rl["qp_pair_hot_ms_per_step"] < 2 * rl["qp_pair_ms_per_step"]
    assert full["qp_pair"]["bytes_per_pair"] == {"moved": 1538, "algorithmic": 1920}
    # physical bound on everything that does not say `algorithmic`
    def walk(m, path=""):
        for k, v in m.items():
            if isinstance(v, dict):
                yield from walk(v, path + k + ".")
            elif isinstance(v, float) and "frac" in k and "algorithmic" not in k:
                yield path + k, v
    fr = list(walk(full))
    assert not [(k, v) for k, v in fr if v > 1.0], "a fraction above 1"
    phys = [(k, v) for k, v in fr if not any(t in k for t in ("fp64", "valu", "flip", "SQ_", "busy"))]
    assert not [(k, v) for k, v in phys if v > 0.79], [(k, v) for k, v in phys if v > 0.79]
    assert 0.3 < cf["cfg4_moved_frac"] < 0.79
    assert rl["traffic"] is None or rl["traffic"] > 0.5 * 704 * 65536
    # the reference's execution model beside the C port
    cb = d["cpu_baseline"]
    assert cb["python_loop_value"] > 0 and cb["python_loop_value"] < cb["single_thread_value"] * 1.5
    fb = full["cpu_baseline"]
    assert "qcqp.py:29-31" in fb["python_loop_sample"] and "1.1e4" in fb["reference_published"]


def test_live_pmc_measures_the_traffic_when_asked():
    """--live-pmc: two rocprofv3 counter passes spawned by the run over its own launches (not part of the default run)."""
    d, full = _run("--no-per-config", "--no-cpu-baseline", "--no-check", "--live-pmc")
    rl = d["roofline"]
    if "live_pmc_error" in full["roofline"]:
        pytest.skip("rocprofv3 counter pass failed on this box: %s" % full["roofline"]["live_pmc_error"][:200])
    assert "live" in rl["traffic_source"] and rl["traffic"] > 0.5 * 704 * 65536
    assert 0.8 < rl["traffic_over_algorithmic"] < 2.0
