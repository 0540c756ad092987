"""The tests that drive the ALTERNATIVE kernels -- other lane layouts, queued instead of fused fallback, team instead of lane
kernels, re-spread thresholds ... -- need run-time tuning knobs, which only the developer build of the library has
(csrc/tuning.h, `python -m diffqcqp_amd.build --tuning`); on the shipped library they skip.  This test runs those files once
more in a subprocess bound to the developer build (DQQ_LIB), so that one `pytest -m gpu` covers both builds: nothing in the
knob-driven files may fail there, and nothing may skip for want of a knob."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["test_gpu_parity.py", "test_gpu_respread.py", "test_gpu_group_dense.py", "test_gpu_edge_cases.py",
         "test_gpu_reference_inputs.py", "test_gpu_graph_capture.py", "test_gpu_fuzz.py"]


@pytest.mark.skipif(os.environ.get("DQQ_DEV_SUBPROCESS") == "1", reason="already inside the developer-build run")
def test_knob_driven_files_pass_on_the_developer_build():
    from diffqcqp_amd import build
    lib = build.build_tuning()
    assert os.path.exists(lib)
    env = dict(os.environ, DQQ_LIB=lib, DQQ_DEV_SUBPROCESS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-rs",
                        *[os.path.join(ROOT, "tests", f) for f in FILES]],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"(\d+) passed", r.stdout)
    assert m and int(m.group(1)) >= 400, tail
    assert "needs the developer build" not in r.stdout, tail     # no test skipped for want of a knob
    k = re.search(r"(\d+) skipped", r.stdout)
    assert (int(k.group(1)) if k else 0) <= 6, tail               # (QCQP with odd N: the reference has no such problem)
