"""Builds libdiffqcqp_hip.so (the C-ABI library of include/diffqcqp_hip.h) for gfx950.

hipcc cross-compiles without a GPU.  Every translation unit is compiled for
gfx950 only; the forward fast path allows FMA contraction, everything that has
to reproduce the reference's operation order (backward, dense path) is built
with -ffp-contract=off.  Output: diffqcqp_amd/lib/libdiffqcqp_hip.so (in-tree,
git-ignored, shipped to the GPU box with the snapshot).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIBDIR, "libdiffqcqp_hip.so")
PYMOD = os.path.join(LIBDIR, "_dqq.so")   # pybind11 module over the C ABI (csrc/pybind_module.cpp)
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-fvisibility=hidden",
          "-Wall", "-Wno-unused-function"]
# translation unit -> extra flags
UNITS = {
    "fwd_diag.hip": ["-ffp-contract=fast-honor-pragmas"],
    "bwd_diag.hip": ["-ffp-contract=off"],
    "dense.hip": ["-ffp-contract=off"],
    "general_any.hip": ["-ffp-contract=off"],
    "bwd_small.hip": ["-ffp-contract=off"],
    "bwd_lane_dense.hip": ["-ffp-contract=off"],
    "dense_wave64.hip": ["-ffp-contract=fast"],
    "bwd_wave_qcqp.hip": ["-ffp-contract=off"],
    "bwd_wave_qcqp_big.hip": ["-ffp-contract=off"],
    "fwd_lane_dense.hip": ["-ffp-contract=fast"],
    "fwd_small.hip": ["-ffp-contract=fast"],
    "capi.hip": ["-ffp-contract=off", "-fvisibility=default"],
}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _sources():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "diffqcqp_hip.h")]
    return deps


STAMP = os.path.join(LIBDIR, "build_flags.txt")


def source_sha16():
    """sha256 (16 hex digits) over what decides the library's contents: the CODE of csrc/* and include/diffqcqp_hip.h
    (comments and blank lines stripped) and the flags.  A
    counter summary under profiles/ records it (tools/summarize_prof.py); bench.py quotes counter-derived figures only from
    a summary of THIS build."""
    import hashlib
    import re

    def code(path):
        """The file without comments and blank lines: a comment-only edit is the same build."""
        with open(path, "r", errors="replace") as fh:
            text = fh.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        lines = [re.sub(r"//.*$", "", ln).rstrip() for ln in text.split("\n")]
        return "\n".join(ln for ln in lines if ln.strip()).encode()

    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        h.update(f.encode())
        h.update(code(os.path.join(CSRC, f)))
    h.update(code(os.path.join(INCLUDE, "diffqcqp_hip.h")))
    h.update(" ".join(COMMON + [u + " " + " ".join(f) for u, f in sorted(UNITS.items())]).encode())
    return h.hexdigest()[:16]


def _flag_stamp():
    """Everything besides the sources that decides what the library contains (DQQ_EXTRA_FLAGS: developer -D flags)."""
    return " ".join(COMMON) + " | " + os.environ.get("DQQ_EXTRA_FLAGS", "")


def _read_stamp():
    """(flag line, pybind line) of the last build; the second line records whether the pybind11 module was built
    ("pybind11: ok") or why not -- a host without pybind11 must not rebuild the HIP units on every call."""
    try:
        lines = open(STAMP).read().split("\n")
    except OSError:
        return None, None
    return lines[0], (lines[1] if len(lines) > 1 else None)


def needs_build():
    """Does the C-ABI LIBRARY have to be rebuilt?  (The pybind11 module is optional and has its own test,
    `_pybind_stale`: its absence alone never recompiles the HIP units.)"""
    if not os.path.exists(LIB):
        return True
    flags, _ = _read_stamp()
    if flags != _flag_stamp():
        return True   # built with other flags (e.g. an experiment's -D...): never reuse it silently
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _sources() + [os.path.abspath(__file__)])


def _pybind_stale():
    """The module must be (re)built when it is older than what it binds: its source, the header, the library.  A
    recorded failure ("pybind11: unavailable ...") is not retried until one of those changes."""
    deps = [os.path.join(CSRC, "pybind_module.cpp"), os.path.join(INCLUDE, "diffqcqp_hip.h"), LIB]
    newest = max(os.path.getmtime(p) for p in deps)
    if os.path.exists(PYMOD):
        return os.path.getmtime(PYMOD) < newest
    _, note = _read_stamp()
    if note is not None and note.startswith("pybind11: unavailable"):
        try:
            return os.path.getmtime(STAMP) < newest
        except OSError:
            return True
    return True


def _compile(unit, flags, objdir, verbose, more=()):
    obj = os.path.join(objdir, unit.replace(".hip", ".o"))
    extra = os.environ.get("DQQ_EXTRA_FLAGS", "").split() + list(more)  # developer experiments (-D...)
    cmd = [_hipcc()] + COMMON + flags + extra + ["-I", INCLUDE, "-c", os.path.join(CSRC, unit), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False):
    """Compile and link; returns the library path."""
    if not force and not needs_build():
        if _pybind_stale():
            _write_stamp(_build_pybind(verbose))
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    for stale in os.listdir(objdir):   # objects of units that no longer exist must never be linked
        if stale.endswith(".o") and stale[:-2] + ".hip" not in UNITS:
            os.remove(os.path.join(objdir, stale))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda kv: _compile(kv[0], kv[1], objdir, verbose), UNITS.items()))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    _write_stamp(_build_pybind(verbose))
    return LIB


def _write_stamp(pybind_note):
    with open(STAMP, "w") as f:
        f.write(_flag_stamp() + "\n" + pybind_note)


def _build_pybind(verbose=False):
    """The pybind11 module `_dqq` (host-only C++, g++): every C-ABI function under its own name.  Optional: without
    pybind11 headers `_capi.py` binds the same symbols with ctypes.  Returns the line for the stamp file.  The old
    module is deleted BEFORE the attempt: after a C-ABI change a failed rebuild must fall back to ctypes, never to a
    stale module with another argument order."""
    try:
        os.remove(PYMOD)
    except OSError:
        pass
    try:
        import pybind11
        import sysconfig
    except ImportError:
        return "pybind11: unavailable (no pybind11 headers); _capi.py binds the C ABI with ctypes"
    cxx = os.environ.get("CXX", "g++")
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-I", INCLUDE, "-I", pybind11.get_include(),
           "-I", sysconfig.get_paths()["include"], os.path.join(CSRC, "pybind_module.cpp"), "-o", PYMOD,
           "-L", LIBDIR, "-ldiffqcqp_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
    except (OSError, subprocess.CalledProcessError) as e:
        try:
            os.remove(PYMOD)
        except OSError:
            pass
        return "pybind11: unavailable (%s failed: %s); _capi.py binds the C ABI with ctypes" % (cxx, type(e).__name__)
    return "pybind11: ok"


TUNING_LIB = os.path.join(LIBDIR, "tuning", "libdiffqcqp_hip.so")


def build_tuning(force=False, verbose=False):
    """The DEVELOPER build of the same library (-DDQQ_TUNING, csrc/tuning.h: the kernel-selection knobs as run-time options
    behind dqq_set_option) into lib/tuning/ -- never loaded by default: `DQQ_LIB=<path> python ...` selects it (bound with
    ctypes, diffqcqp_amd/_capi.py).  For A/B sweeps (tools/) and for the tests that drive the alternative kernels."""
    if not force and os.path.exists(TUNING_LIB):
        t = os.path.getmtime(TUNING_LIB)
        if not any(os.path.getmtime(p) > t for p in _sources() + [os.path.abspath(__file__)]):
            return TUNING_LIB
    objdir = os.path.join(LIBDIR, "tuning", "obj")
    os.makedirs(objdir, exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda kv: _compile(kv[0], kv[1], objdir, verbose, more=("-DDQQ_TUNING",)), UNITS.items()))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", TUNING_LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TUNING_LIB


if __name__ == "__main__":
    if "--tuning" in sys.argv:
        print(build_tuning(force="--force" in sys.argv, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
