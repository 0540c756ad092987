"""Python binding of libdiffqcqp_hip.so (include/diffqcqp_hip.h): the pybind11 module `_dqq`
(csrc/pybind_module.cpp, built next to the library) when it is there, ctypes otherwise -- the same
symbols, the same argument order, pointers as Python ints either way.  DQQ_BINDING=ctypes|pybind11
forces one of them.

The library is the product: if it is missing or cannot be loaded this module
raises -- there is no CPU or PyTorch fallback behind it.
"""
import ctypes
import os

# torch ships its own libamdhip64.so; it must be the HIP runtime of the process.  Loading our library
# first would pull in /opt/rocm's copy and give the two halves of the process different runtimes
# (launches then fail with hipErrorNoDevice), so torch is imported before the library is opened.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# DQQ_LIB: developer override -- another build of the same C ABI (tools/build_variant.sh), bound with ctypes
LIB_PATH = os.environ.get("DQQ_LIB") or os.path.join(_HERE, "lib", "libdiffqcqp_hip.so")
PYMOD_PATH = os.path.join(_HERE, "lib", "_dqq.so")

P_AUTO, P_DENSE, P_DIAG = 0, 1, 2
F_REFERENCE_ORDER = 0x100   # DQQ_F_REFERENCE_ORDER: ORed into p_layout (16 < N <= 64 on the reference-order kernels)

_ERRORS = {
    -1: "DQQ_E_NULLPTR: a required pointer is NULL",
    -2: "DQQ_E_BAD_SIZE: B < 0, N < 1, or odd N for a QCQP",
    -3: "DQQ_E_UNSUPPORTED_N: the compact diagonal layout (DQQ_P_DIAG) takes N in {2, 4, 8, 16, 32, 64} only",
    -4: "DQQ_E_BAD_LAYOUT: unknown p_layout",
    -5: "DQQ_E_WORKSPACE: workspace missing or too small",
    -6: "DQQ_E_BAD_OPTION: unknown option name",
}

_lib = None
_vp, _i, _d, _i64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_int64, ctypes.c_size_t

# name -> argtypes, in the order of include/diffqcqp_hip.h
SIGNATURES = {
    "dqq_workspace_bytes": ([_i64], _sz),
    "dqq_scratch_bytes": ([_i, _i, _i, _i64, _i], _sz),
    "dqq_max_n": ([_i, _i], _i),
    "dqq_workspace_reset": ([_vp, _sz, _vp], _i),
    "dqq_workspace_status": ([_vp, _sz, _vp, ctypes.POINTER(_i)], _i),
    "dqq_qp_fwd_f64": ([_vp, _vp, _vp, _i64, _i, _d, _d, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp], _i),
    "dqq_qp_bwd_f64": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _i, _vp, _vp, _vp, _vp, _sz, _vp], _i),
    "dqq_qcqp_fwd_f64": ([_vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _d, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp], _i),
    "dqq_qcqp_bwd_f64": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _i, _vp, _vp, _vp,
                          _vp, _sz, _vp], _i),
    "dqq_boxqp_fwd_f64": ([_vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _d, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp], _i),
    "dqq_signedboxqp_fwd_f64": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _d, _i, _i, _i, _vp, _vp, _vp, _vp, _sz,
                                 _vp], _i),
    "dqq_boxqp_bwd_f64": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _i, _vp, _vp, _vp,
                           _vp, _sz, _vp], _i),
    "dqq_set_option": ([ctypes.c_char_p, _i], _i),
    "dqq_get_option": ([ctypes.c_char_p, ctypes.POINTER(_i)], _i),
    "dqq_set_feedback": ([_vp, _sz], _i),
    "dqq_version": ([], ctypes.c_char_p),
}


_binding = None


def ctypes_lib():
    """The library through ctypes (always available when the library is)."""
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "diffqcqp_amd: %s not found. Build it with `python -m diffqcqp_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    handle = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SIGNATURES.items():
        if os.environ.get("DQQ_LIB") and name == "dqq_set_feedback" and not hasattr(handle, name):
            continue   # (developer A/B against a build that predates the entry point: ops.feedback_default then finds no hint)
        fn = getattr(handle, name)  # AttributeError if the library does not export it
        fn.argtypes = argtypes
        fn.restype = restype
    return handle


def pybind_lib():
    """The pybind11 module over the same C ABI, or None when it has not been built."""
    if not (os.path.exists(PYMOD_PATH) and os.path.exists(LIB_PATH)):
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("_dqq", PYMOD_PATH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name in SIGNATURES:
        getattr(mod, name)  # AttributeError if the module does not bind it
    return mod


def lib():
    """Load (once) and return the C-ABI library (pybind11 module or ctypes handle: same call surface); raises if it
    is not there."""
    global _lib, _binding
    if _lib is None:
        want = os.environ.get("DQQ_BINDING", "") or ("ctypes" if os.environ.get("DQQ_LIB") else "")
        mod = None if want == "ctypes" else pybind_lib()
        if mod is None and want == "pybind11":
            raise RuntimeError("diffqcqp_amd: DQQ_BINDING=pybind11 but %s is not built" % PYMOD_PATH)
        _lib, _binding = (mod, "pybind11") if mod is not None else (ctypes_lib(), "ctypes")
    return _lib


def binding():
    lib()
    return _binding


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise ValueError("%s failed: %s" % (what, _ERRORS.get(rc, "error %d" % rc)))
    raise RuntimeError("%s failed: hipError_t %d" % (what, rc))


def set_option(name, value):
    """Shipped library: the three route counters only ("lane_list_drains", "bwd_whole_batches", "fwd_feedback_routes"; 0
    resets).  A developer build (-DDQQ_TUNING, see tuning_build()) also takes the kernel-selection knobs of csrc/tuning.h."""
    check(lib().dqq_set_option(name.encode(), int(value)), "dqq_set_option(%s)" % name)


def tuning_build():
    """Is the loaded library a developer build with run-time tuning knobs (DQQ_EXTRA_FLAGS=-DDQQ_TUNING)?"""
    if binding() == "pybind11":
        return lib().dqq_get_option(b"fwd_lpp")[0] == 0
    v = _i(0)
    return lib().dqq_get_option(b"fwd_lpp", ctypes.byref(v)) == 0


def workspace_reset(ws, stream=0):
    """dqq_workspace_reset: zero-fill the work-list header of `ws` on `stream` (asynchronous)."""
    check(lib().dqq_workspace_reset(ws.data_ptr(), ws.numel() * ws.element_size(), stream or None), "dqq_workspace_reset")


def workspace_status(ws, stream=0):
    """dqq_workspace_status: True if a kernel has found this workspace's work-list header inconsistent since its last
    reset.  Synchronises the stream."""
    if binding() == "pybind11":
        rc, dirty = lib().dqq_workspace_status(ws.data_ptr(), ws.numel() * ws.element_size(), stream or None)
    else:
        v = _i(0)
        rc = lib().dqq_workspace_status(ws.data_ptr(), ws.numel() * ws.element_size(), stream or None, ctypes.byref(v))
        dirty = v.value
    check(rc, "dqq_workspace_status")
    return bool(dirty)


def get_option(name):
    if binding() == "pybind11":
        rc, value = lib().dqq_get_option(name.encode())
        check(rc, "dqq_get_option(%s)" % name)
        return value
    v = _i(0)
    check(lib().dqq_get_option(name.encode(), ctypes.byref(v)), "dqq_get_option(%s)" % name)
    return v.value


FEEDBACK_BYTES = 128   # DQQ_FEEDBACK_BYTES
_feedback = None       # the buffer while it is REGISTERED with the library (None: not registered)
_feedback_buf = None   # the pinned tensor itself: allocated once, never freed (see enable_feedback)


def enable_feedback(on=True):
    """Register (or unregister) the feedback buffer of include/diffqcqp_hip.h dqq_set_feedback: 128 bytes of pinned host
    memory through which the drain launch of the N <= 8 backward tells the next call how many non-diagonal problems it found.
    A timing hint only -- results are the same bits with and without.  Needs a GPU (pinned memory).

    The pinned buffer lives as long as the process: drain kernels still in flight, and launches captured into a
    HIP graph while it was registered, hold its device address and may store into it at any later time -- memory handed back
    to torch's pinned allocator could by then belong to someone else.  Unregistering only makes the library stop passing the
    address to NEW launches and stop reading the words; registering again hands the library the same buffer with its words
    as they are (consistent with the per-workspace record of what was last sent, csrc/launch.h: a zeroed buffer at the same
    address would never be written again for an unchanged count)."""
    global _feedback, _feedback_buf
    if not on:
        check(lib().dqq_set_feedback(None, 0), "dqq_set_feedback(NULL)")
        _feedback = None
        return
    if _feedback is None:
        if _feedback_buf is None:
            _feedback_buf = torch.zeros(FEEDBACK_BYTES // 8, dtype=torch.int64).pin_memory()
        check(lib().dqq_set_feedback(_feedback_buf.data_ptr(), FEEDBACK_BYTES), "dqq_set_feedback")
        _feedback = _feedback_buf


def feedback_words():
    """The registered buffer's words as (B, entries) pairs, index kind * 4 + N / 2 - 1 (a debugging view), or None."""
    if _feedback is None:
        return None
    return [((int(w) >> 32) & 0x3fffffff, int(w) & 0x7fffffff) for w in _feedback.tolist()]   # (bit 31: entries are single problems)


def feedback_streaks():
    """Bits 62..63 of each word: consecutive earlier reports of "three quarters of the batch or more" (saturating at 3)."""
    return None if _feedback is None else [(int(w) >> 62) & 3 for w in _feedback.tolist()]


def version():
    return lib().dqq_version().decode()
