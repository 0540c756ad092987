"""Bounded runs of the randomised route fuzzers (tools/fuzz_small.py, tools/fuzz_bwd.py): random kind, size, batch,
structure, layout and tuning options against the oracle.  A run of tools/fuzz_small.py found the one routing bug of
round 2 (DQQ_P_DENSE + a since-removed option); these keep the net in place with fixed seeds."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), *args], capture_output=True, text=True,
                       timeout=1200, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " 0 failures" in r.stdout, tail


def test_fuzz_forward_small_sizes():
    _run("fuzz_small.py", "120", "11")


def test_fuzz_forward_all_sizes():
    _run("fuzz_small.py", "120", "12", "big")


def test_fuzz_backward_all_sizes():
    _run("fuzz_bwd.py", "100", "13", "big")
