#!/bin/bash
# Debug build of the library with the wave timeline compiled in (common.h DQQ_TL): every wave of fwd_diag_kernel
# records the 100 MHz constant clock at entry / P consumed / inputs loaded / loop start / loop end / stores issued.
# Output: tools/ubench/bin/libdqq_timeline.so (git-ignored; travels to the GPU box).  Used by tools/probe_timeline.py.
set -e
cd "$(dirname "$0")/../.."
python - <<'PY'
import os, subprocess
from concurrent.futures import ThreadPoolExecutor
from diffqcqp_amd import build as b
out = "tools/ubench/bin"; objdir = os.path.join(out, "obj_tl"); os.makedirs(objdir, exist_ok=True)
def cc(kv):
    obj = os.path.join(objdir, kv[0].replace(".hip", ".o"))
    subprocess.check_call([b._hipcc()] + b.COMMON + kv[1] + ["-DDQQ_TIMELINE"] + os.environ.get("DQQ_EXTRA", "").split() + ["-I", b.INCLUDE, "-c",
                          os.path.join(b.CSRC, kv[0]), "-o", obj])
    return obj
with ThreadPoolExecutor(4) as ex:
    objs = list(ex.map(cc, b.UNITS.items()))
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
                       os.path.join(out, "libdqq_timeline.so")] + objs)
print(os.path.join(out, "libdqq_timeline.so"))
PY
