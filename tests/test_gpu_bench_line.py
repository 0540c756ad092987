"""The one JSON line bench.py prints (the driver's contract), on a short run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2",
                        "--repeats", "2", "--no-cold", *flags], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    return json.loads(lines[0])


def test_headline_line_carries_the_contract_keys():
    d = _run("--no-per-config")
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
    assert abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12 and 0 < rl["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert d["value"] > 100 * cb["value"]   # (a reported baseline, not the target; but a GPU path slower than that is broken)


def test_a_config_line_checks_itself_against_the_oracle():
    d = _run("--config", "3", "--no-cpu-baseline")
    assert "configs[2]" in d["config"]["workload"]
    err = d.get("parity_max_abs_err_vs_oracle_sample") or d.get("parity")
    assert err is not None
