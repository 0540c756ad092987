#!/usr/bin/env python3
"""Developer tool: run bench.py with tuning knobs of the C ABI set first.  python tools/bench_opt.py fwd_lpp=1 wpb=4 -- [bench args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
opts, rest = args[:split], args[split + 1:]
from diffqcqp_amd import _capi  # noqa: E402
for o in opts:
    k, v = o.split("=")
    _capi.set_option(k, int(v))
sys.argv = [os.path.join(ROOT, "bench.py")] + rest
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
