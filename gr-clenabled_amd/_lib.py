"""Loader for libmi355_clenabled.so (the C ABI in include/mi355_clenabled.h).

There is no fallback of any kind: if the shared library is missing or a call
fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355_clenabled.so")
_lib = None


class Mi355Error(RuntimeError):
    def __init__(self, code, where, detail):
        super().__init__("%s failed: %s (%d) %s" % (where, _strerror(code), code, detail))
        self.code = code


def _strerror(code):
    try:
        return lib().mi355_strerror(code).decode()
    except Exception:  # pragma: no cover
        return "?"


LOG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_char_p)  # mi355_log_fn
_log_keepalive = None


def set_log_callback(fn):
    """fn(level, message) receives every diagnostics / error line of the library (None restores stderr)."""
    global _log_keepalive
    cb = LOG_FN(lambda user, level, msg: fn(level, msg.decode())) if fn is not None else C.cast(None, LOG_FN)
    check(lib().mi355_set_log_callback(cb, None), "mi355_set_log_callback")
    _log_keepalive = cb  # the C side holds the pointer: keep the thunk alive


def lib():
    """Return the ctypes handle, loading the library on first use."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C %s/csrc` (or __graft_entry__.build()); "
            "gr-clenabled_amd has no non-HIP fallback" % (LIB_PATH, _HERE))
    try:
        # torch ships its own libamdhip64 (same SONAME); importing it first makes
        # this library bind to the HIP runtime that owns torch's device memory.
        import torch  # noqa: F401
    except Exception:  # torch is plumbing, not a requirement of the C ABI
        pass
    L = C.CDLL(LIB_PATH)
    vp, i, sz, f = C.c_void_p, C.c_int, C.c_size_t, C.c_float
    pp = C.POINTER(C.c_void_p)
    sigs = {
        "mi355_strerror": (C.c_char_p, [i]),
        "mi355_last_error": (C.c_char_p, []),
        "mi355_version": (C.c_char_p, []),
        "mi355_set_log_callback": (i, [LOG_FN, vp]),
        "mi355_device_count": (i, []),
        "mi355_ctx_create": (i, [i, i, i, i, i, pp]),
        "mi355_ctx_destroy": (i, [vp]),
        "mi355_ctx_device": (i, [vp]),
        "mi355_ctx_stream": (vp, [vp]),
        "mi355_ctx_synchronize": (i, [vp]),
        "mi355_malloc": (i, [vp, sz, pp]),
        "mi355_free": (i, [vp, vp]),
        "mi355_memcpy_h2d": (i, [vp, vp, vp, sz]),
        "mi355_memcpy_d2h": (i, [vp, vp, vp, sz]),
        "mi355_mathop_create": (i, [vp, i, i, sz, pp]),
        "mi355_mathop_destroy": (i, [vp]),
        "mi355_mathop_work": (i, [vp, sz, vp, vp, vp]),
        "mi355_mathop_work_dev": (i, [vp, sz, vp, vp, vp, vp]),
        "mi355_mathconst_create": (i, [vp, i, i, f, sz, pp]),
        "mi355_mathconst_destroy": (i, [vp]),
        "mi355_mathconst_set_k": (i, [vp, f]),
        "mi355_mathconst_get_k": (i, [vp, C.POINTER(f)]),
        "mi355_mathconst_work": (i, [vp, sz, vp, vp]),
        "mi355_mathconst_work_dev": (i, [vp, sz, vp, vp, vp]),
        "mi355_fft_create": (i, [vp, i, i, vp, i, i, i, i, pp]),
        "mi355_fft_destroy": (i, [vp]),
        "mi355_fft_plan_text": (i, [i, C.c_char_p, i]),
        "mi355_fft_work": (i, [vp, i, pp, pp]),
        "mi355_fft_work_dev": (i, [vp, i, vp, vp, vp]),
        "mi355_filter_create": (i, [vp, i, vp, i, i, i, pp]),
        "mi355_filter_destroy": (i, [vp]),
        "mi355_filter_set_taps": (i, [vp, vp, i]),
        "mi355_filter_ntaps": (i, [vp]),
        "mi355_filter_get_taps": (i, [vp, vp, i]),
        "mi355_filter_fftsize": (i, [vp]),
        "mi355_filter_work": (i, [vp, sz, vp, vp]),
        "mi355_filter_work_dev": (i, [vp, sz, vp, vp, vp]),
        "mi355_pfb_create": (i, [vp, vp, i, i, i, i, vp, i, pp]),
        "mi355_pfb_destroy": (i, [vp]),
        "mi355_pfb_noutput": (i, [vp]),
        "mi355_pfb_ninput": (i, [vp]),
        "mi355_pfb_work": (i, [vp, vp, vp]),
        "mi355_pfb_work_dev": (i, [vp, vp, vp, vp]),
        "mi355_pfb_work_dev_n": (i, [vp, i, vp, vp, vp]),
        "mi355_xengine_create": (i, [vp, i, i, i, i, i, pp]),
        "mi355_xengine_destroy": (i, [vp]),
        "mi355_xengine_input_bytes": (sz, [vp]),
        "mi355_xengine_output_items": (sz, [vp]),
        "mi355_xengine_xcorrelate": (i, [vp, vp, vp, i]),
        "mi355_xengine_xcorrelate_dev": (i, [vp, vp, vp, i, vp]),
        "mi355_xengine_xcorrelate_grouped_dev": (i, [vp, vp, vp, i, i, vp]),
        "mi355_xengine_xcorrelate_n_dev": (i, [vp, i, vp, vp, i, i, vp]),
        "mi355_pack3d_dev": (i, [vp, vp, vp, sz, sz, sz, sz, sz, sz, sz, vp]),
        "mi355_xengine_gather": (i, [vp, i, i, pp, vp]),
        "mi355_xengine_selftest_scale": (i, [vp, C.POINTER(C.c_longlong)]),
        "mi355_xengine_last_route": (i, [vp, vp]),
        "mi355_xengine_shard_create": (i, [i, C.POINTER(C.c_int), i, i, i, i, i, pp]),
        "mi355_xengine_shard_destroy": (i, [vp]),
        "mi355_xengine_shard_world": (i, [vp]),
        "mi355_xengine_shard_device": (i, [vp, i]),
        "mi355_xengine_shard_frames_bytes": (sz, [vp]),
        "mi355_xengine_shard_slab_items": (sz, [vp]),
        "mi355_xengine_shard_stream": (vp, [vp, i]),
        "mi355_xengine_shard_submit_dev": (i, [vp, pp, pp, i]),
        "mi355_xengine_shard_wait_stream": (i, [vp, i, vp]),
        "mi355_xengine_shard_synchronize": (i, [vp]),
        "mi355_xengine_shard_xcorrelate": (i, [vp, vp, vp, i]),
        "mi355_xengine_shard_windows": (i, [vp]),
        "mi355_xengine_shard_input_bytes": (sz, [vp]),
        "mi355_xengine_shard_acquire": (i, [vp, pp]),
        "mi355_xengine_shard_submit_acquired": (i, [vp]),
        "mi355_xengine_shard_wait": (i, [vp, vp]),
        "mi355_xengine_shard_pending": (i, [vp]),
        "mi355_xengine_submit": (i, [vp, vp, vp]),
        "mi355_xengine_wait": (i, [vp, vp]),
        "mi355_xengine_pending": (i, [vp]),
        "mi355_xengine_acquire": (i, [vp, pp]),
        "mi355_xengine_submit_acquired": (i, [vp, vp]),
        "mi355_elem_create": (i, [vp, i, f, f, pp]),
        "mi355_elem_destroy": (i, [vp]),
        "mi355_elem_history": (i, [vp]),
        "mi355_elem_work": (i, [vp, sz, vp, vp, vp, vp]),
        "mi355_elem_work_dev": (i, [vp, sz, vp, vp, vp, vp, vp]),
        "mi355_xcorr_fft_create": (i, [vp, i, i, i, pp]),
        "mi355_xcorr_fft_destroy": (i, [vp]),
        "mi355_xcorr_fft_work": (i, [vp, i, vp, vp]),
        "mi355_xcorr_fft_work_dev": (i, [vp, i, vp, vp, vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    L._declared = tuple(sigs)
    _lib = L
    return L


def check(code, where):
    if code != 0:
        raise Mi355Error(code, where, lib().mi355_last_error().decode())
    return code
