"""ctypes binding of liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module (see oracle/oracle.h).  The product package
``gr-clenabled_amd`` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DTYPE_COMPLEX, DTYPE_FLOAT, DTYPE_INT, DTYPE_SHORT, DTYPE_BYTE, DTYPE_PACKEDXY = 1, 2, 3, 4, 5, 6
OP_MULTIPLY, OP_ADD, OP_SUBTRACT, OP_CONJUGATE, OP_MULTIPLY_CONJUGATE = 1, 2, 3, 4, 5
OP_EMPTY_W_COPY, OP_EMPTY = 254, 255
WIN_HAMMING, WIN_HANN, WIN_BLACKMAN, WIN_RECTANGULAR, WIN_KAISER, WIN_BLACKMAN_HARRIS, WIN_BARTLETT, WIN_FLATTOP = range(8)

_NP = {DTYPE_COMPLEX: np.complex64, DTYPE_FLOAT: np.float32, DTYPE_INT: np.int32}


def build():
    """(Re)build liboracle.so with gcc; building the checker is not using it."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        vp, i, sz, f, d = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_double
        L.oracle_mathop.argtypes = [i, i, sz, vp, vp, vp]
        L.oracle_mathconst.argtypes = [i, i, f, sz, vp, vp]
        L.oracle_window.argtypes = [i, i, d, vp]
        L.oracle_window_max_attenuation.argtypes = [i, d]
        L.oracle_window_max_attenuation.restype = d
        L.oracle_firdes_ntaps.argtypes = [d, d, i, d]
        L.oracle_firdes_low_pass.argtypes = [d, d, d, d, i, d, vp, i]
        L.oracle_fft_c2c_f32.argtypes = [i, i, vp, vp]
        L.oracle_fft_c2c_f64.argtypes = [i, i, vp, vp]
        L.oracle_fft_block.argtypes = [i, i, vp, i, i, i, vp, vp, i]
        L.oracle_xcorr_fft.argtypes = [i, i, i, i, vp, vp, i]
        L.oracle_fft_filter_new.argtypes = [i, vp, i]
        L.oracle_fft_filter_new.restype = vp
        L.oracle_fft_filter_free.argtypes = [vp]
        L.oracle_fft_filter_free.restype = None
        L.oracle_fft_filter_set_taps.argtypes = [vp, vp, i]
        L.oracle_fft_filter_fftsize.argtypes = [vp]
        L.oracle_fft_filter_nsamples.argtypes = [vp]
        L.oracle_fft_filter_xformed_taps.argtypes = [vp, vp]
        L.oracle_fft_filter_filter.argtypes = [vp, i, vp, vp]
        L.oracle_fir_ccf_filterN.argtypes = [vp, i, vp, vp, sz, i]
        L.oracle_fir_ccc_filterN.argtypes = [vp, i, vp, vp, sz, i]
        L.oracle_pfb_channelizer.argtypes = [vp, i, i, i, i, vp, i, vp, vp, i]
        L.oracle_xengine_out_len.argtypes = [i, i, i]
        L.oracle_xengine_out_len.restype = sz
        L.oracle_xengine_cf32.argtypes = [i, i, i, i, vp, vp, i]
        L.oracle_xengine_ichar.argtypes = [i, i, i, i, vp, vp, i, i]
        L.oracle_xengine_packed4.argtypes = [i, i, i, vp, vp, i]
        L.oracle_xengine_gather.argtypes = [i, i, i, i, i, i, vp, vp]
        L.oracle_elem.argtypes = [i, f, f, sz, vp, vp, vp, vp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def mathop(dtype, op, a, b):
    a, b = _c(a, _NP[dtype]), _c(b, _NP[dtype])
    out = np.empty_like(a)
    rc = lib().oracle_mathop(dtype, op, a.size, _p(a), _p(b), _p(out))
    if rc:
        raise ValueError("oracle_mathop rc=%d" % rc)
    return out


def mathconst(dtype, op, k, a):
    a = _c(a, _NP[dtype])
    out = np.empty_like(a)
    rc = lib().oracle_mathconst(dtype, op, float(k), a.size, _p(a), _p(out))
    if rc:
        raise ValueError("oracle_mathconst rc=%d" % rc)
    return out


def window(wtype, ntaps, beta=6.76):
    out = np.empty(ntaps, np.float32)
    if lib().oracle_window(wtype, ntaps, float(beta), _p(out)):
        raise ValueError("bad window")
    return out


def firdes_low_pass(gain, fs, cutoff, tw, wtype=WIN_HAMMING, beta=6.76):
    n = lib().oracle_firdes_ntaps(fs, tw, wtype, beta)
    out = np.empty(n, np.float32)
    r = lib().oracle_firdes_low_pass(gain, fs, cutoff, tw, wtype, beta, _p(out), n)
    if r != n:
        raise ValueError("firdes_low_pass failed")
    return out


def fft(x, sign=-1, f64=False):
    x = _c(x, np.complex64)
    out = np.empty_like(x)
    fn = lib().oracle_fft_c2c_f64 if f64 else lib().oracle_fft_c2c_f32
    if fn(x.size, sign, _p(x), _p(out)):
        raise ValueError("fft size must be a power of two")
    return out


def fft_block(n, forward, window, shift, dtype, x, f64=False):
    x = _c(x, _NP[dtype])
    nvec = x.size // n
    out = np.empty(nvec * n, np.complex64)
    w = None if window is None or len(window) == 0 else _c(window, np.float32)
    rc = lib().oracle_fft_block(n, int(forward), _p(w) if w is not None else None, int(shift), dtype,
                                nvec, _p(x), _p(out), int(f64))
    if rc:
        raise ValueError("oracle_fft_block rc=%d" % rc)
    return out


class FFTFilter:
    """Restatement of fft_filter_ccf (stateful overlap-add, lib/fft_filter.cc)."""

    def __init__(self, decimation, taps):
        t = _c(taps, np.float32)
        self._h = lib().oracle_fft_filter_new(decimation, _p(t), t.size)
        self.decimation = decimation
        if not self._h:
            raise ValueError("oracle_fft_filter_new failed")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_fft_filter_free(self._h)
            self._h = None

    @property
    def fftsize(self):
        return lib().oracle_fft_filter_fftsize(self._h)

    @property
    def nsamples(self):
        return lib().oracle_fft_filter_nsamples(self._h)

    def xformed_taps(self):
        out = np.empty(self.fftsize, np.complex64)
        lib().oracle_fft_filter_xformed_taps(self._h, _p(out))
        return out

    def filter(self, nitems, x):
        """nitems outputs; x must hold ceil(nitems*decim/nsamples)*nsamples samples."""
        x = _c(x, np.complex64)
        ns = self.nsamples
        need = -(-nitems * self.decimation // ns) * ns
        if x.size < need:
            x = np.concatenate([x, np.zeros(need - x.size, np.complex64)])
        out = np.empty(need // self.decimation + 2, np.complex64)
        w = lib().oracle_fft_filter_filter(self._h, nitems, _p(x), _p(out))
        return out[:w]


def fir_ccf(taps, x_hist, n, decim=1):
    t, x = _c(taps, np.float32), _c(x_hist, np.complex64)
    assert x.size >= (n - 1) * decim + t.size
    out = np.empty(n, np.complex64)
    lib().oracle_fir_ccf_filterN(_p(t), t.size, _p(x), _p(out), n, decim)
    return out


def fir_ccc(taps, x_hist, n, decim=1):
    t, x = _c(taps, np.complex64), _c(x_hist, np.complex64)
    assert x.size >= (n - 1) * decim + t.size
    out = np.empty(n, np.complex64)
    lib().oracle_fir_ccc_filterN(_p(t), t.size, _p(x), _p(out), n, decim)
    return out


def pfb(taps, buf_items, nch, ninputs_per_iter, ch_map, x_hist, f64=False):
    t, x = _c(taps, np.float32), _c(x_hist, np.complex64)
    m = _c(ch_map, np.int32)
    assert x.size >= buf_items - ninputs_per_iter + t.size
    out = np.empty(m.size * buf_items // ninputs_per_iter, np.complex64)
    rc = lib().oracle_pfb_channelizer(_p(t), t.size, buf_items, nch, ninputs_per_iter, _p(m), m.size,
                                      _p(x), _p(out), int(f64))
    if rc:
        raise ValueError("oracle_pfb_channelizer rc=%d" % rc)
    return out


def xengine_out_len(ninputs, nchan, npol):
    return lib().oracle_xengine_out_len(ninputs, nchan, npol)


def xengine_cf32(ninputs, nchan, npol, ntime, x, acc=None):
    x = _c(x, np.complex64)
    out = np.zeros(xengine_out_len(ninputs, nchan, npol), np.complex64) if acc is None else acc
    rc = lib().oracle_xengine_cf32(ninputs, nchan, npol, ntime, _p(x), _p(out), int(acc is not None))
    if rc:
        raise ValueError("oracle_xengine_cf32 rc=%d" % rc)
    return out


def xengine_ichar(ninputs, nchan, npol, ntime, x, exact=True, acc=None):
    x = _c(x, np.int8)
    out = np.zeros(xengine_out_len(ninputs, nchan, npol), np.complex64) if acc is None else acc
    rc = lib().oracle_xengine_ichar(ninputs, nchan, npol, ntime, _p(x), _p(out), int(acc is not None), int(exact))
    if rc:
        raise ValueError("oracle_xengine_ichar rc=%d" % rc)
    return out


def xengine_packed4(ninputs, nchan, ntime, x, acc=None):
    x = _c(x, np.uint8)
    out = np.zeros(xengine_out_len(ninputs, nchan, 2), np.complex64) if acc is None else acc
    rc = lib().oracle_xengine_packed4(ninputs, nchan, ntime, _p(x), _p(out), int(acc is not None))
    if rc:
        raise ValueError("oracle_xengine_packed4 rc=%d" % rc)
    return out


def xengine_gather(dtype, ninputs, nchan, npol, nframes, frame0, inputs, frame_buffer):
    arr = (C.c_void_p * len(inputs))(*[i.ctypes.data for i in inputs])
    rc = lib().oracle_xengine_gather(dtype, ninputs, nchan, npol, nframes, frame0, arr, _p(frame_buffer))
    if rc:
        raise ValueError("oracle_xengine_gather rc=%d" % rc)
    return frame_buffer


ELEM_LOG10, ELEM_SNR, ELEM_C2MAG, ELEM_C2ARG, ELEM_C2MAGPHASE, ELEM_MAGPHASE2C, ELEM_QUADDEMOD = range(1, 8)
_ELEM_IO = {1: ((np.float32,), (np.float32,)), 2: ((np.float32, np.float32), (np.float32,)), 3: ((np.complex64,), (np.float32,)),
            4: ((np.complex64,), (np.float32,)), 5: ((np.complex64,), (np.float32, np.float32)),
            6: ((np.float32, np.float32), (np.complex64,)), 7: ((np.complex64,), (np.float32,))}


def elem(kind, n, ins, p0=1.0, p1=0.0):
    """Remaining elementwise blocks; returns a tuple of outputs. QUADDEMOD input carries 1 item of history."""
    it, ot = _ELEM_IO[kind]
    ins = [_c(x, t) for x, t in zip(ins, it)]
    outs = [np.empty(n, t) for t in ot]
    rc = lib().oracle_elem(kind, float(p0), float(p1), n, _p(ins[0]), _p(ins[1]) if len(ins) > 1 else None,
                           _p(outs[0]), _p(outs[1]) if len(outs) > 1 else None)
    if rc:
        raise ValueError("oracle_elem rc=%d" % rc)
    return tuple(outs)


def xcorr_fft(n, input_type, inputs, use_f64=False):
    """clxcorrelate_fft_vcf: inputs = list of [nframes*n] complex64 (input 0 = reference); returns num_inputs-1 float32 arrays."""
    ins = [_c(x, np.complex64) for x in inputs]
    nframes = ins[0].size // n
    outs = [np.empty(nframes * n, np.float32) for _ in ins[1:]]
    ip = (C.c_void_p * len(ins))(*[x.ctypes.data for x in ins])
    op = (C.c_void_p * len(outs))(*[x.ctypes.data for x in outs])
    rc = lib().oracle_xcorr_fft(n, len(ins), input_type, nframes, ip, op, 1 if use_f64 else 0)
    if rc:
        raise ValueError("oracle_xcorr_fft rc=%d" % rc)
    return outs
