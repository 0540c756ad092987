#!/bin/sh
# oracle/ref_build.sh -- builds oracle/_ref/libref.so: the REFERENCE's own qcqplib/Solver.cpp (compiled where it lies
# under /root/reference; nothing is copied) + the extern "C" shim oracle/ref_capi.cpp.  Test infrastructure only.
#
# DORMANT in this image: the reference needs <Eigen/Dense> and there are no Eigen3 headers here (SURVEY.md 8(c),
# oracle/README.md), and the rules forbid stand-in headers -- so without Eigen this script prints why and exits 0
# having built nothing; tests/test_oracle_vs_ref.py then skips.  With Eigen3 headers present:
#     EIGEN3_INCLUDE_DIR=/usr/include/eigen3 sh oracle/ref_build.sh
# No CMake, no pybind11 (the reference's pybind11 submodule directory is empty): two translation units, one g++ line.
# oracle/_ref/ is git-ignored (never in history) but NOT gpurun-ignored (the built .so travels to the GPU box).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
REF=${DQQ_REFERENCE_DIR:-/root/reference}
if [ ! -f "$REF/qcqplib/Solver.cpp" ]; then
    echo "ref_build: $REF/qcqplib/Solver.cpp not found (no reference checkout on this machine): nothing built"; exit 0
fi
INC=""
# where Eigen3 headers could be: the usual prefixes, a conda environment, and Python wheels that ship them (cmeel-eigen /
# eigenpy put them under site-packages/cmeel.prefix/include/eigen3; some wheels under <pkg>/include/eigen3) -- none exist
# in this image today; the day one does, the first build() / pytest run pins the oracle without an edit
PYDIRS=$(python3 - <<'PY' 2>/dev/null
import glob, os, sys
seen = []
for p in sys.path:
    if p and os.path.isdir(p):
        for pat in ("cmeel.prefix/include/eigen3", "*/include/eigen3", "*/*/include/eigen3", "*/include"):
            for d in glob.glob(os.path.join(p, pat)):
                if os.path.isfile(os.path.join(d, "Eigen", "Dense")) and d not in seen:
                    seen.append(d)
print(" ".join(seen))
PY
)
for d in "$EIGEN3_INCLUDE_DIR" /usr/include/eigen3 /usr/local/include/eigen3 /opt/conda/include/eigen3 \
         "${CONDA_PREFIX:+$CONDA_PREFIX/include/eigen3}" /usr/include /usr/local/include $PYDIRS; do
    if [ -n "$d" ] && [ -f "$d/Eigen/Dense" ]; then INC=$d; break; fi
done
if [ -z "$INC" ]; then
    echo "ref_build: no Eigen3 headers (set EIGEN3_INCLUDE_DIR): the reference is unbuildable here, nothing built"; exit 0
fi
mkdir -p "$HERE/_ref"
# the reference's own optimisation level (setup.py:52 CMAKE_BUILD_TYPE=Release => -O3 -DNDEBUG); no -ffast-math
${CXX:-g++} -O3 -DNDEBUG -std=c++14 -fPIC -shared -fvisibility=default -I"$INC" -I"$REF" \
    "$REF/qcqplib/Solver.cpp" "$HERE/ref_capi.cpp" -o "$HERE/_ref/libref.so"
echo "ref_build: built $HERE/_ref/libref.so against Eigen in $INC"
