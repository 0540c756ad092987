"""Python binding of libdiffqcqp_hip.so (include/diffqcqp_hip.h): the pybind11 module `_dqq`
(csrc/pybind_module.cpp, built next to the library) when it is there, ctypes otherwise -- the same
symbols, the same argument order, pointers as Python ints either way.  DQQ_BINDING=ctypes|pybind11
forces one of them.

The library is the product: if it is missing or cannot be loaded this module
raises -- there is no CPU or PyTorch fallback behind it.
"""
import ctypes
import os
import threading

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
F_EXPECT_DENSE, F_EXPECT_LONG_LIST = 0x200, 0x400   # hint flags (dqq_hint_flags): routes of identical results

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
    "dqq_qp_bwd_f64": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp], _i),
    "dqq_qcqp_fwd_f64": ([_vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _d, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp], _i),
    "dqq_qcqp_bwd_f64": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _i, _vp, _vp, _vp,
                          _vp, _vp, _sz, _vp], _i),
    "dqq_boxqp_fwd_f64": ([_vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _d, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp], _i),
    "dqq_signedboxqp_fwd_f64": ([_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _d, _i, _i, _i, _vp, _vp, _vp, _vp, _sz,
                                 _vp], _i),
    "dqq_boxqp_bwd_f64": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _d, _i, _vp, _vp, _vp,
                           _vp, _sz, _vp], _i),
    "dqq_set_option": ([ctypes.c_char_p, _i], _i),
    "dqq_get_option": ([ctypes.c_char_p, ctypes.POINTER(_i)], _i),
    "dqq_hint_flags": ([_i, _i, _i, _i64, ctypes.c_ulonglong], _i),
    "dqq_device_pointer": ([_vp, ctypes.POINTER(_vp)], _i),
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


# ---- the route hints (include/diffqcqp_hip.h: "adapting to the data without state in the library") ----------------------
# The C library keeps no state.  This module does what its header describes for a caller: one 8-byte report word of pinned host
# memory per (device, kind, N) -- 16 words per device --, handed to the backward calls as `report`, read back with a plain host
# load before every call and turned into hint flags by the pure function dqq_hint_flags.  A hint selects between kernels of
# identical results; no hints are given while the current stream is being captured.  On by default (the first backward of an
# N <= 8 DQQ_P_AUTO batch creates the device's words; DQQ_FEEDBACK=0 in the environment turns it off).
HINT_WORDS = 16
_hints_on = os.environ.get("DQQ_FEEDBACK", "1") != "0"
_hint_store = {}       # device index -> (pinned int64 tensor, host address, device address); never freed (in-flight launches
                       # and captured graphs hold the device address)
_feedback = None       # the words of cuda:0 (tests poke them), None until created / when off


def enable_feedback(on=True):
    """Turn the hints on or off for this process (the words stay where they are: launches in flight and captured graphs may
    still write to them; switched on again, the words are as they were left)."""
    global _hints_on, _feedback
    _hints_on = bool(on)
    if not on:
        _feedback = None
    elif torch.cuda.is_available():
        _feedback = _words(0)[0]


_hint_lock = threading.Lock()


def _words(dev_index):
    st = _hint_store.get(dev_index)
    if st is not None:
        return st
    with _hint_lock:   # two threads making a device's first backward call: ONE buffer, never a dropped one (its device
        st = _hint_store.get(dev_index)   # address may already be in a launch's arguments)
        if st is not None:
            return st
        buf = torch.zeros(HINT_WORDS, dtype=torch.int64).pin_memory()
        if binding() == "pybind11":
            rc, dev = lib().dqq_device_pointer(buf.data_ptr())
        else:
            out = _vp(0)
            rc = lib().dqq_device_pointer(buf.data_ptr(), ctypes.byref(out))
            dev = out.value or 0
        check(rc, "dqq_device_pointer")
        st = _hint_store[dev_index] = (buf, buf.data_ptr(), dev)
        if dev_index == 0:
            global _feedback
            _feedback = buf
    return st


def hint_index(kind, N):
    return kind * 4 + N // 2 - 1 if (kind in (0, 1) and 2 <= N <= 8 and N % 2 == 0) else -1


def hint(kind, pas, N, B, dev_index, capturing=False):
    """-> (flags to OR into p_layout, device address of the report word or None) for a DQQ_P_AUTO call of (kind, pass, N, B)
    on device dev_index.  (0, None) when hints are off or do not apply; no flags while capturing."""
    i = hint_index(kind, N)
    if not _hints_on or i < 0:
        return 0, None
    _, host, dev = _words(dev_index)
    if capturing:
        return 0, dev + 8 * i
    word = ctypes.c_ulonglong.from_address(host + 8 * i).value
    flags = lib().dqq_hint_flags(kind, pas, N, B, word) if word else 0
    return flags, dev + 8 * i


def feedback_words(dev_index=0):
    """The report words of a device as (B, entries) pairs, index kind * 4 + N / 2 - 1 (a debugging view), or None."""
    if not _hints_on or dev_index not in _hint_store:
        return None
    return [((int(w) >> 32) & 0x3fffffff, int(w) & 0x7fffffff) for w in _hint_store[dev_index][0].tolist()]   # (bit 31: entries are single problems)


def feedback_streaks(dev_index=0):
    """Bits 62..63 of each word: consecutive earlier reports of "three quarters of the batch or more" (saturating at 3)."""
    if not _hints_on or dev_index not in _hint_store:
        return None
    return [(int(w) >> 62) & 3 for w in _hint_store[dev_index][0].tolist()]


def version():
    return lib().dqq_version().decode()
