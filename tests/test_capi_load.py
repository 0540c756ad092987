"""CPU checks of the drop-in boundary: libdiffqcqp_hip.so builds for gfx950,
loads, exports every symbol include/diffqcqp_hip.h declares, and validates its
arguments before touching a GPU.  No compute call is made here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=["ctypes", "pybind11"])
def lib(request):
    """Both faces of the C ABI: the ctypes handle and the pybind11 module `_dqq` (csrc/pybind_module.cpp) -- same
    symbols, same argument order, pointers as Python ints."""
    from diffqcqp_amd import build, _capi
    build.build()
    if request.param == "ctypes":
        return _capi.ctypes_lib()
    mod = _capi.pybind_lib()
    assert mod is not None, "the pybind11 module was not built (pybind11 headers are in this image)"
    return mod


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "diffqcqp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dqq_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
    names = _declared_symbols()
    assert {"dqq_qp_fwd_f64", "dqq_qp_bwd_f64", "dqq_qcqp_fwd_f64", "dqq_qcqp_bwd_f64"} <= set(names)
    for n in names:
        assert hasattr(lib, n), "library does not export %s" % n
    from diffqcqp_amd import _capi
    assert set(_capi.SIGNATURES) == set(names), "python binding and header disagree"


def test_version_and_limits(lib):
    assert b"gfx950" in lib.dqq_version()
    assert lib.dqq_max_n(0, 0) == 64 and lib.dqq_max_n(1, 0) == 64 and lib.dqq_max_n(2, 0) == 64 and lib.dqq_max_n(3, 0) == 21
    assert lib.dqq_workspace_bytes(0) >= 16
    assert lib.dqq_workspace_bytes(65536) >= 4 * 65536
    # scratch of the global-memory kernels: the caller's, a function of (kind, pass, N, B); 0 for every BASELINE config
    for kind, pas, N, B in ((0, 0, 8, 65536), (1, 0, 8, 65536), (1, 1, 8, 65536), (0, 0, 32, 262144), (0, 1, 32, 262144),
                            (0, 0, 64, 65536), (0, 1, 64, 65536)):
        assert lib.dqq_scratch_bytes(kind, pas, N, B, 0) == 0
    assert lib.dqq_scratch_bytes(0, 0, 65, 10, 0) > 0 and lib.dqq_scratch_bytes(1, 1, 66, 10, 0) > 0
    assert lib.dqq_scratch_bytes(1, 1, 44, 10, 0) == 0 and lib.dqq_scratch_bytes(1, 1, 64, 10, 0) == 0   # register-resident kernels
    REF = 0x100   # DQQ_F_REFERENCE_ORDER, a per-call flag: the reference-order route needs (and demands) scratch
    assert lib.dqq_scratch_bytes(1, 1, 44, 10, 1 | REF) > 0 and lib.dqq_max_n(2, REF) == 42 and lib.dqq_max_n(2, 0) == 64
    assert lib.dqq_scratch_bytes(2, 1, 22, 10, 0) > 0 and lib.dqq_scratch_bytes(3, 1, 200, 10, 0) == 0
    assert lib.dqq_scratch_bytes(0, 0, 65, 4, 0) * 2 == lib.dqq_scratch_bytes(0, 0, 65, 8, 0)   # a slice per workgroup
    assert lib.dqq_scratch_bytes(0, 0, 65, 10 ** 6, 0) == lib.dqq_scratch_bytes(0, 0, 65, 10 ** 7, 0)  # persistent grid


def test_argument_validation_without_gpu(lib):
    one = 8  # a pointer as a Python int, never dereferenced: the checks fail first
    f = lib.dqq_qp_fwd_f64
    assert f(one, one, one, -1, 8, 1e-7, 1e-7, 10, 1, 0, None, None, None, None, 0, None) == -2
    assert f(one, one, one, 4, 0, 1e-7, 1e-7, 10, 1, 0, None, None, None, None, 0, None) == -2
    assert f(one, one, one, 4, 8, 1e-7, 1e-7, 10, 1, 7, None, None, None, None, 0, None) == -4
    assert f(None, one, one, 4, 8, 1e-7, 1e-7, 10, 1, 0, None, None, None, None, 0, None) == -1
    assert f(one, one, one, 4, 8, 1e-7, 1e-7, 10, 1, 0, None, None, None, None, 0, None) == -5  # AUTO needs a workspace
    assert f(one, one, one, 4, 200, 1e-7, 1e-7, 10, 1, 2, None, None, None, None, 0, None) == -3  # compact diagonal: fast-path sizes only
    assert f(one, one, one, 0, 8, 1e-7, 1e-7, 10, 1, 0, None, None, None, None, 0, None) == 0   # empty batch: no-op
    g = lib.dqq_qcqp_fwd_f64
    assert g(one, one, one, one, one, 4, 7, 1e-7, 1e-7, 10, 1, 0, None, None, None, None, 0, None) == -2  # odd N
    assert lib.dqq_qcqp_bwd_f64(one, one, one, one, one, one, None, None, None, None, None, None, 4, 66, 1e-10, 2,
                                None, None, None, None, None, 0, None) == -3
    assert lib.dqq_qcqp_bwd_f64(one, one, one, one, one, one, None, None, None, None, None, None, 4, 67, 1e-10, 1,
                                None, None, None, None, None, 0, None) == -2  # odd N
    # DQQ_P_DENSE beyond the register / LDS kernels: the scratch is the caller's, nothing is allocated inside
    assert f(one, one, one, 4, 80, 1e-7, 1e-7, 10, 1, 1, None, None, None, None, 0, None) == -5
    assert lib.dqq_set_option(b"no_such_knob", 1) == -6
    # the shipped library has no kernel-selection knobs (csrc/tuning.h): the three route counters are all dqq_set_option knows
    v = ctypes.c_int(7)

    def get(name):   # (rc, value) through either binding
        if isinstance(lib, ctypes.CDLL):
            rc = lib.dqq_get_option(name, ctypes.byref(v))
            return rc, v.value
        return tuple(lib.dqq_get_option(name))
    tuning = get(b"fwd_lpp")[0] == 0
    for name in (b"lane_list_drains", b"bwd_whole_batches", b"fwd_feedback_routes"):
        assert lib.dqq_set_option(name, 0) == 0 and get(name) == (0, 0)
    if not tuning:
        for name in (b"fwd_lpp", b"wpb", b"fuse_fallback", b"auto_fallback", b"dense_wave64", b"wave_qcqp_bwd", b"lane_dense",
                     b"lane_defer", b"fwd_respread", b"fwd_compact", b"dense_teams", b"small_bwd", b"lane_bwd"):
            assert lib.dqq_set_option(name, 1) == -6, name
    # unknown flag bits in p_layout are refused; the reference-order flag is accepted with every layout
    assert f(one, one, one, 4, 8, 1e-7, 1e-7, 10, 1, 0x800, None, None, None, None, 0, None) == -4
    assert f(one, one, one, 4, 8, 1e-7, 1e-7, 10, 1, 0x103, None, None, None, None, 0, None) == -4
    # the work-list hygiene entry points validate their arguments without touching the device
    assert lib.dqq_workspace_reset(None, 0, None) == -1 and lib.dqq_workspace_reset(one, 16, None) == -5
    if isinstance(lib, ctypes.CDLL):
        assert lib.dqq_workspace_status(None, 0, None, None) == -1 and lib.dqq_workspace_status(one, 16, None, ctypes.byref(v)) == -5
    else:
        assert lib.dqq_workspace_status(None, 0, None)[0] == -1 and lib.dqq_workspace_status(one, 16, None)[0] == -5
    # the hint protocol keeps its state with the caller: dqq_hint_flags is a pure function of the caller's report word
    EXPECT_DENSE, EXPECT_LONG = 0x200, 0x400
    B = 65536
    word = lambda count, streak=0, single=0: (streak << 62) | (B << 32) | (single << 31) | count
    assert lib.dqq_hint_flags(1, 1, 8, B, 0) == 0                                    # nothing known
    assert lib.dqq_hint_flags(1, 1, 8, B, word(100)) == 0                            # a short list: the team kernel
    assert lib.dqq_hint_flags(1, 1, 8, B, word(30000)) == EXPECT_LONG                # a list that fills the chip
    assert lib.dqq_hint_flags(1, 1, 8, B, word(B)) == EXPECT_LONG                    # all of it, once: not yet "expect dense"
    assert lib.dqq_hint_flags(1, 1, 8, B, word(B, streak=1)) == EXPECT_LONG | EXPECT_DENSE
    assert lib.dqq_hint_flags(1, 1, 8, B // 2, word(B, streak=1)) == 0               # a word about another batch size
    assert lib.dqq_hint_flags(1, 0, 8, B, word(B // 2)) == EXPECT_DENSE and lib.dqq_hint_flags(1, 0, 8, B, word(B // 2 - 1)) == 0
    assert lib.dqq_hint_flags(0, 0, 4, B, word(B)) == 0                              # the forward's other layout exists at N = 8 only
    assert lib.dqq_hint_flags(2, 1, 8, B, word(B, streak=3)) == 0 and lib.dqq_hint_flags(0, 1, 16, B, word(B, streak=3)) == 0
    assert lib.dqq_hint_flags(1, 1, 8, 1000, (1 << 62) | (1000 << 32) | 1000) == 0   # batches the lane kernel does not take
    # hint flags are accepted in p_layout (and checked like any other argument); dqq_device_pointer refuses garbage
    assert f(one, one, one, 4, 8, 1e-7, 1e-7, 10, 1, EXPECT_DENSE | EXPECT_LONG, None, None, None, None, 0, None) == -5
    if isinstance(lib, ctypes.CDLL):
        out = ctypes.c_void_p(0)
        assert lib.dqq_device_pointer(None, ctypes.byref(out)) == -1 and lib.dqq_device_pointer(4100, ctypes.byref(out)) == -2
    else:
        assert lib.dqq_device_pointer(None)[0] == -1 and lib.dqq_device_pointer(4100)[0] == -2


def test_python_layer_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from diffqcqp_amd.qcqp import QPFn2
    P = torch.eye(2).unsqueeze(0)
    q = -torch.ones(1, 2, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        QPFn2.apply(P, q, torch.zeros(1, 2, 1), 1e-7, 100)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under diffqcqp_amd/ may reference it."""
    pkg = os.path.join(ROOT, "diffqcqp_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_header_is_plain_c(tmp_path):
    """include/diffqcqp_hip.h is the C ABI: it must compile as C99 (no C++ or torch types) and link against
    the library by name."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "diffqcqp_hip.h"\n'
                   "int main(void) { return dqq_max_n(0, 0) == 64 && dqq_workspace_bytes(0) >= 16 ? 0 : 1; }\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                           "-c", str(src), "-o", str(tmp_path / "use_header.o")])


def test_reference_import_lines_resolve():
    """The reference's own import statements work unchanged against this build: `from diffqcqp import ...`
    (reference qcqp.py:17), `from qcqp import QPFn2, QCQPFn2` (README.md:31), qcqp_no_batch."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    m = importlib.import_module("diffqcqp")
    for name in ("solveQP", "solveBoxQP", "solveQCQP", "solveDerivativesQP", "solveDerivativesBoxQP",
                 "solveDerivativesQCQP", "solveSignedBoxQP"):
        assert callable(getattr(m, name))
    q = importlib.import_module("qcqp")
    assert all(hasattr(q, n) for n in ("QPFn2", "QCQPFn2", "BoxQPFn2", "SignedBoxQPFn2"))
    nb = importlib.import_module("qcqp_no_batch")
    assert hasattr(nb, "QPFn2") and hasattr(nb, "QCQPFn2")


def test_missing_pybind_module_never_rebuilds_the_hip_units():
    """the optional pybind11 module has its own staleness test.  Its absence alone must not recompile the HIP
    translation units (a host without pybind11 would rebuild the library on every build.build() call), a stale module is
    deleted before its rebuild is attempted (a failed rebuild falls back to ctypes, never to a module with another
    argument order), and the outcome is recorded in the stamp file."""
    import time
    from diffqcqp_amd import build
    build.build()
    assert not build.needs_build()
    flags, note = build._read_stamp()
    assert flags == build._flag_stamp() and note is not None and note.startswith("pybind11:")
    if note != "pybind11: ok":
        assert not os.path.exists(build.PYMOD)
        return
    lib_mtime = os.path.getmtime(build.LIB)
    os.remove(build.PYMOD)
    try:
        assert not build.needs_build(), "a missing _dqq.so must not trigger a rebuild of the HIP units"
        assert build._pybind_stale()
        t0 = time.time()
        build.build()
        assert time.time() - t0 < 60 and os.path.getmtime(build.LIB) == lib_mtime     # only the module was rebuilt
        assert os.path.exists(build.PYMOD) and not build._pybind_stale()
        # a recorded failure is not retried until a dependency changes
        os.remove(build.PYMOD)
        build._write_stamp("pybind11: unavailable (test)")
        assert not build.needs_build() and not build._pybind_stale()
    finally:
        build._write_stamp(build._build_pybind())
    assert os.path.exists(build.PYMOD)
