"""Host-side mirror of the gr::clenabled block API for the hot path.

Constructor arguments are the reference's ``make(...)`` arguments, in the same
positional order GRC passes them (include/clenabled/*.h, grc/*.block.yml), and
``work()/general_work()`` keep the ``noutput_items`` contract.  Everything is a
thin call into the C ABI (include/mi355_clenabled.h); there is no Python or CPU
compute path here.

Two flavours of each work call:
  work(noutput_items, input_items, output_items)         numpy (host) buffers,
      the GNU Radio contract: blocking, H2D / kernel / D2H inside the call
  work_device(noutput_items, input_items, output_items)  torch CUDA tensors,
      enqueue-only on torch's current stream (device-resident chaining / bench)
"""
import ctypes as C

import numpy as np

from ._lib import check, lib

# include/clenabled/GRCLBase.h:57-70, clMathOpTypes.h:11-20
DTYPE_COMPLEX, DTYPE_FLOAT, DTYPE_INT, DTYPE_SHORT, DTYPE_BYTE, DTYPE_PACKEDXY = 1, 2, 3, 4, 5, 6
OCLTYPE_GPU, OCLTYPE_ACCELERATOR, OCLTYPE_CPU, OCLTYPE_ANY = 1, 2, 3, 4
OCLDEVICESELECTOR_FIRST, OCLDEVICESELECTOR_SPECIFIC = 1, 2
MATHOP_MULTIPLY, MATHOP_ADD, MATHOP_SUBTRACT, MATHOP_COMPLEX_CONJUGATE, MATHOP_MULTIPLY_CONJUGATE = 1, 2, 3, 4, 5
MATHOP_EMPTY, MATHOP_EMPTY_W_COPY = 255, 254
CLFFT_FORWARD, CLFFT_BACKWARD = -1, 1
CLXCORR_TRIANGULAR_ORDER, CLXCORR_FULL_MATRIX = 1, 2

_NP_OF = {DTYPE_COMPLEX: np.complex64, DTYPE_FLOAT: np.float32, DTYPE_INT: np.int32}


def _host(a, dtype=None, writable=False):
    if not isinstance(a, np.ndarray) or not a.flags["C_CONTIGUOUS"] or (dtype is not None and a.dtype != dtype):
        if writable:
            raise TypeError("output buffers must be C-contiguous numpy arrays of dtype %s" % dtype)
        a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _hp(a):
    return C.c_void_p(a.ctypes.data)


def _dp(t, nbytes=None, name="device buffer"):
    """Device pointer of a contiguous CUDA tensor; nbytes: what the call reads or writes there (the C ABI takes plain pointers, so a
    tensor that is too short would be a memory fault on the device, not an error)."""
    if not t.is_cuda or not t.is_contiguous():
        raise TypeError("device path needs contiguous CUDA tensors")
    if nbytes is not None and t.numel() * t.element_size() < nbytes:
        raise ValueError("%s holds %d bytes, the call needs %d" % (name, t.numel() * t.element_size(), nbytes))
    return C.c_void_p(t.data_ptr())


def _torch_stream(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _Block:
    """Owns one mi355 context, like every reference block owns one cl::Context
    (lib/GRCLBase.cpp:115-144)."""

    def __init__(self, openCLPlatformType, devSelector, platformId, devId, setDebug):
        self._L = lib()
        self._ctx = C.c_void_p()
        self._h = C.c_void_p()
        check(self._L.mi355_ctx_create(int(openCLPlatformType), int(devSelector), int(platformId), int(devId),
                                       1 if setDebug else 0, C.byref(self._ctx)), "mi355_ctx_create")
        self.device = self._L.mi355_ctx_device(self._ctx)

    _destroy = None

    def stop(self):
        if getattr(self, "_h", None) and self._h.value and self._destroy:
            getattr(self._L, self._destroy)(self._h)
            self._h = C.c_void_p()
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._L.mi355_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()
        return True

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass

    def synchronize(self):
        check(self._L.mi355_ctx_synchronize(self._ctx), "mi355_ctx_synchronize")


def _need(name, arr, items):
    """work() contract: every buffer holds at least noutput_items items (the C ABI copies exactly that many)."""
    if arr.size < items:
        raise ValueError("%s holds %d items, the call needs %d" % (name, arr.size, items))


class clMathOp(_Block):
    """clMathOp::make(idataType, openCLPlatformType, devSelector, platformId, devId,
    operatorType, setDebug=0)  -- include/clenabled/clMathOp.h:42"""
    _destroy = "mi355_mathop_destroy"

    def __init__(self, idataType, openCLPlatformType, devSelector, platformId, devId, operatorType, setDebug=0):
        super().__init__(openCLPlatformType, devSelector, platformId, devId, setDebug)
        self.dtype = idataType
        check(self._L.mi355_mathop_create(self._ctx, int(idataType), int(operatorType), 0, C.byref(self._h)),
              "mi355_mathop_create")

    def work(self, noutput_items, input_items, output_items):
        dt = _NP_OF[self.dtype]
        if len(input_items) < 2 or len(output_items) < 1:
            raise ValueError("clMathOp.work needs two inputs and one output")
        a, b = _host(input_items[0], dt), _host(input_items[1], dt)
        c = _host(output_items[0], dt, writable=True)
        for name, arr in (("input 0", a), ("input 1", b), ("output", c)):
            _need(name, arr, noutput_items)
        check(self._L.mi355_mathop_work(self._h, noutput_items, _hp(a), _hp(b), _hp(c)), "mi355_mathop_work")
        return noutput_items

    testOpenCL = work  # lib/clMathOp_impl.cc:354-359

    def work_device(self, noutput_items, input_items, output_items):
        nb = int(noutput_items) * np.dtype(_NP_OF[self.dtype]).itemsize
        check(self._L.mi355_mathop_work_dev(self._h, noutput_items, _dp(input_items[0], nb, "input 0"), _dp(input_items[1], nb, "input 1"),
                                            _dp(output_items[0], nb, "output"), _torch_stream(self.device)), "mi355_mathop_work_dev")
        return noutput_items


class clMathConst(_Block):
    """clMathConst::make(idataType, openCLPlatformType, devSelector, platformId, devId,
    fValue, operatorType, setDebug=0)  -- include/clenabled/clMathConst.h:51"""
    _destroy = "mi355_mathconst_destroy"

    def __init__(self, idataType, openCLPlatformType, devSelector, platformId, devId, fValue, operatorType, setDebug=0):
        super().__init__(openCLPlatformType, devSelector, platformId, devId, setDebug)
        self.dtype = idataType
        check(self._L.mi355_mathconst_create(self._ctx, int(idataType), int(operatorType), float(fValue), 0,
                                             C.byref(self._h)), "mi355_mathconst_create")

    def k(self):
        v = C.c_float()
        check(self._L.mi355_mathconst_get_k(self._h, C.byref(v)), "mi355_mathconst_get_k")
        return v.value

    def set_k(self, newValue):
        check(self._L.mi355_mathconst_set_k(self._h, float(newValue)), "mi355_mathconst_set_k")

    def work(self, noutput_items, input_items, output_items):
        dt = _NP_OF[self.dtype]
        a = _host(input_items[0], dt)
        c = _host(output_items[0], dt, writable=True)
        _need("input", a, noutput_items)
        _need("output", c, noutput_items)
        check(self._L.mi355_mathconst_work(self._h, noutput_items, _hp(a), _hp(c)), "mi355_mathconst_work")
        return noutput_items

    testOpenCL = work

    def work_device(self, noutput_items, input_items, output_items):
        nb = int(noutput_items) * np.dtype(_NP_OF[self.dtype]).itemsize
        check(self._L.mi355_mathconst_work_dev(self._h, noutput_items, _dp(input_items[0], nb, "input"), _dp(output_items[0], nb, "output"),
                                               _torch_stream(self.device)), "mi355_mathconst_work_dev")
        return noutput_items


class clFFT(_Block):
    """clFFT::make(fftSize, clFFTDir, window, idataType, openCLPlatformType, devSelector,
    platformId, devId, setDebug=0, num_streams=1, shift=False) -- the positional
    order of lib/clFFT_impl.cc:34-36, which is what GRC emits
    (grc/clenabled_clFFT.block.yml:84-89); the header's parameter names differ
    (SURVEY App. B-1).  noutput_items counts VECTORS of fftSize items."""
    _destroy = "mi355_fft_destroy"

    def __init__(self, fftSize, clFFTDir, window, idataType, openCLPlatformType, devSelector, platformId, devId,
                 setDebug=0, num_streams=1, shift=False):
        window = np.ascontiguousarray(window if window is not None else [], dtype=np.float32)
        if not (window.size == 0 or window.size == fftSize):
            # lib/clFFT_impl.cc:74-76
            raise RuntimeError("OpenCL FFT: window not the same length as fft_size")
        super().__init__(openCLPlatformType, devSelector, platformId, devId, setDebug)
        self.fft_size, self.dtype, self.num_streams = int(fftSize), idataType, int(num_streams)
        check(self._L.mi355_fft_create(self._ctx, int(fftSize), int(clFFTDir), _hp(window) if window.size else None,
                                       int(window.size), int(idataType), int(num_streams), 1 if shift else 0,
                                       C.byref(self._h)), "mi355_fft_create")

    def work(self, noutput_items, input_items, output_items):
        dt = _NP_OF[self.dtype]
        if len(input_items) < self.num_streams or len(output_items) < self.num_streams:
            raise ValueError("clFFT.work needs %d input and output streams" % self.num_streams)
        ins = [_host(x, dt) for x in input_items[:self.num_streams]]
        outs = [_host(x, np.complex64, writable=True) for x in output_items[:self.num_streams]]
        for i, (x, y) in enumerate(zip(ins, outs)):
            _need("input %d" % i, x, noutput_items * self.fft_size)
            _need("output %d" % i, y, noutput_items * self.fft_size)
        pi = (C.c_void_p * len(ins))(*[x.ctypes.data for x in ins])
        po = (C.c_void_p * len(outs))(*[x.ctypes.data for x in outs])
        check(self._L.mi355_fft_work(self._h, noutput_items, pi, po), "mi355_fft_work")
        return noutput_items

    def testOpenCL(self, noutput_items, input_items, output_items):
        # the reference's test hook counts SAMPLES (lib/clFFT_impl.cc:520-524)
        return self.work(noutput_items // self.fft_size, input_items, output_items) * self.fft_size

    def work_device(self, noutput_items, input_items, output_items):
        st = _torch_stream(self.device)
        n = int(noutput_items) * self.fft_size
        for x, y in zip(input_items[:self.num_streams], output_items[:self.num_streams]):
            check(self._L.mi355_fft_work_dev(self._h, noutput_items, _dp(x, n * np.dtype(_NP_OF[self.dtype]).itemsize, "input"),
                                             _dp(y, n * 8, "output"), st), "mi355_fft_work_dev")
        return noutput_items


class _FilterBase(_Block):
    _destroy = "mi355_filter_destroy"

    def _create(self, decimation, taps, complex_taps, use_time):
        self._complex = complex_taps
        self.decimation = int(decimation)
        t = np.ascontiguousarray(taps, dtype=np.complex64 if complex_taps else np.float32)
        check(self._L.mi355_filter_create(self._ctx, int(decimation), _hp(t), int(t.size), 1 if complex_taps else 0,
                                          1 if use_time else 0, C.byref(self._h)), "mi355_filter_create")

    def taps(self):
        n = self._L.mi355_filter_ntaps(self._h)
        out = np.empty(n, np.complex64 if self._complex else np.float32)
        check(min(self._L.mi355_filter_get_taps(self._h, _hp(out), n), 0), "mi355_filter_get_taps")
        return out

    def ntaps(self):
        return self._L.mi355_filter_ntaps(self._h)

    def set_taps2(self, taps):
        t = np.ascontiguousarray(taps, dtype=np.complex64 if self._complex else np.float32)
        check(self._L.mi355_filter_set_taps(self._h, _hp(t), int(t.size)), "mi355_filter_set_taps")

    set_taps = set_taps2

    def history(self):
        return self.ntaps()

    def fftsize(self):
        return self._L.mi355_filter_fftsize(self._h)

    def set_nthreads(self, n):  # lib/clFilter_impl.cc:413-415: only meaningful for the CPU FFTW plan
        pass

    def work(self, noutput_items, input_items, output_items):
        """input_items[0] is the history-prefixed buffer: noutput*decim + ntaps-1 items."""
        x = _host(input_items[0], np.complex64)
        need = noutput_items * self.decimation + self.ntaps() - 1
        if x.size < need:
            raise ValueError("filter work(): need %d input items (history included), got %d" % (need, x.size))
        y = _host(output_items[0], np.complex64, writable=True)
        check(self._L.mi355_filter_work(self._h, noutput_items, _hp(x), _hp(y)), "mi355_filter_work")
        return noutput_items

    def testOpenCL(self, noutput_items, input_items, output_items):
        return self.work(noutput_items, input_items, output_items)

    def work_device(self, noutput_items, input_items, output_items):
        need = int(noutput_items) * self.decimation + self.ntaps() - 1  # history-prefixed input, like work()
        check(self._L.mi355_filter_work_dev(self._h, noutput_items, _dp(input_items[0], need * 8, "input"),
                                            _dp(output_items[0], int(noutput_items) * 8, "output"),
                                            _torch_stream(self.device)), "mi355_filter_work_dev")
        return noutput_items


class clFilter(_FilterBase):
    """clFilter::make(openclPlatform, devSelector, platformId, devId, decimation, taps,
    nthreads=1, setDebug=0, use_time=False)  -- include/clenabled/clFilter.h:52-53"""

    def __init__(self, openclPlatform, devSelector, platformId, devId, decimation, taps, nthreads=1, setDebug=0,
                 use_time=False):
        super().__init__(openclPlatform, devSelector, platformId, devId, setDebug)
        self._create(decimation, taps, False, use_time)


class clComplexFilter(_FilterBase):
    """clComplexFilter::make(openclPlatform, devSelector, platformId, devId, decimation,
    taps, nthreads=1, setDebug=0)  -- include/clenabled/clComplexFilter.h:706
    The reference only has the time-domain kernel for complex taps; ``use_time``
    is an additive keyword selecting the fast-convolution kernel instead."""

    def __init__(self, openclPlatform, devSelector, platformId, devId, decimation, taps, nthreads=1, setDebug=0,
                 use_time=True):
        super().__init__(openclPlatform, devSelector, platformId, devId, setDebug)
        self._create(decimation, taps, True, use_time)


class clPolyphaseChannelizer(_Block):
    """clPolyphaseChannelizer::make(openCLPlatformType, devSelector, platformId, devId, taps,
    buf_items, num_channels, ninputs_per_iter, ch_map, setDebug=0)
    -- include/clenabled/clPolyphaseChannelizer.h:48-49"""
    _destroy = "mi355_pfb_destroy"

    def __init__(self, openCLPlatformType, devSelector, platformId, devId, taps, buf_items, num_channels,
                 ninputs_per_iter, ch_map, setDebug=0):
        if buf_items % num_channels != 0:
            # lib/clPolyphaseChannelizer_impl.cc:59-62
            raise ValueError("buf_items must be a multiple of num_channels")
        super().__init__(openCLPlatformType, devSelector, platformId, devId, setDebug)
        t = np.ascontiguousarray(taps, dtype=np.float32)
        m = np.ascontiguousarray(ch_map, dtype=np.int32)
        self._ntaps = int(t.size)
        self.buf_items = int(buf_items)
        check(self._L.mi355_pfb_create(self._ctx, _hp(t), int(t.size), int(buf_items), int(num_channels),
                                       int(ninputs_per_iter), _hp(m), int(m.size), C.byref(self._h)), "mi355_pfb_create")

    def history(self):
        return self._ntaps

    def noutput(self):
        return self._L.mi355_pfb_noutput(self._h)

    def ninput(self):
        return self._L.mi355_pfb_ninput(self._h)

    def general_work(self, noutput_items, ninput_items, input_items, output_items):
        x = _host(input_items[0], np.complex64)
        if x.size < self.ninput():
            raise ValueError("pfb general_work(): need %d input items (history included)" % self.ninput())
        y = _host(output_items[0], np.complex64, writable=True)
        check(self._L.mi355_pfb_work(self._h, _hp(x), _hp(y)), "mi355_pfb_work")
        return self.noutput()

    def work_device(self, input_items, output_items, nbuf=1):
        """Device-resident call; nbuf > 1: that many consecutive buffers of the stream in one launch (general_work() with
        noutput_items = nbuf * noutput()); input nbuf * buf_items - ninputs_per_iter + ntaps items, output nbuf * noutput()."""
        nbuf = int(nbuf)
        nin = (self.ninput() + (nbuf - 1) * self.buf_items) * 8  # k buffers of the stream share the history in front
        nout = nbuf * self.noutput() * 8
        if nbuf == 1:
            check(self._L.mi355_pfb_work_dev(self._h, _dp(input_items[0], nin, "input"), _dp(output_items[0], nout, "output"),
                                             _torch_stream(self.device)), "mi355_pfb_work_dev")
            return self.noutput()
        x, y = input_items[0], output_items[0]
        check(self._L.mi355_pfb_work_dev_n(self._h, nbuf, _dp(x, nin, "input"), _dp(y, nout, "output"), _torch_stream(self.device)),
              "mi355_pfb_work_dev_n")
        return nbuf * self.noutput()


class clXEngine(_Block):
    """clXEngine::make(openCLPlatformType, devSelector, platformId, devId, setDebug, data_type,
    polarization, num_inputs, output_format, first_channel, num_channels, integration,
    antenna_list, ...)  -- include/clenabled/clXEngine.h:48-52.  Only the
    correlation path (xcorrelate / frame gather) is implemented; file/PDU output
    arguments are accepted and ignored."""
    _destroy = "mi355_xengine_destroy"

    def __init__(self, openCLPlatformType, devSelector, platformId, devId, setDebug, data_type, polarization, num_inputs,
                 output_format, first_channel, num_channels, integration, antenna_list=(), output_file=False,
                 file_base="", rollover_size_mb=0, internal_synchronizer=False, sync_timestamp=0, object_name="",
                 starting_chan_center_freq=0.0, channel_width=0.0, disable_output=False, pipeline_integration=0):
        if num_inputs < 2:
            # lib/clXEngine_impl.cc:106-109
            raise IndexError("Please specify at least 2 inputs to correlate.")
        super().__init__(openCLPlatformType, devSelector, platformId, devId, setDebug)
        self.data_type = data_type
        self.npol = 2 if data_type == DTYPE_PACKEDXY else int(polarization)
        self.num_inputs, self.num_channels, self.integration = int(num_inputs), int(num_channels), int(integration)
        self.pipeline_integration = int(pipeline_integration)
        check(self._L.mi355_xengine_create(self._ctx, int(data_type), self.npol, int(num_inputs), int(num_channels),
                                           int(integration), C.byref(self._h)), "mi355_xengine_create")

    def get_input_buffer_size(self):  # lib/clXEngine_impl.h:176 (items, not bytes)
        return self.num_inputs * self.num_channels * self.npol * self.integration

    def input_bytes(self):
        return self._L.mi355_xengine_input_bytes(self._h)

    def get_output_buffer_size(self):
        return self._L.mi355_xengine_output_items(self._h)

    def xcorrelate(self, input_matrix, cross_correlation, accumulate=False):
        """xcorrelate(char*/XComplex* input_matrix, XComplex* cross_correlation)
        -- lib/clXEngine_impl.h:179-201, followed by the blocking read-back of
        runThread (lib/clXEngine_impl.cc:1257)."""
        x = np.ascontiguousarray(input_matrix)
        if x.nbytes < self.input_bytes():
            raise ValueError("xcorrelate: input needs %d bytes" % self.input_bytes())
        y = _host(cross_correlation, np.complex64, writable=True)
        check(self._L.mi355_xengine_xcorrelate(self._h, _hp(x), _hp(y), 1 if accumulate else 0), "mi355_xengine_xcorrelate")
        return self.get_output_buffer_size()

    def submit(self, input_matrix, accumulator=None):
        """Asynchronous xcorrelate: enqueue one integration (at most two in flight); the pinned
        double buffers + worker thread of start()/runThread() (lib/clXEngine_impl.cc:304-382,1234-1299)."""
        x = np.ascontiguousarray(input_matrix)
        if x.nbytes < self.input_bytes():
            raise ValueError("submit: input needs %d bytes" % self.input_bytes())
        acc = None if accumulator is None else _hp(_host(accumulator, np.complex64))
        check(self._L.mi355_xengine_submit(self._h, _hp(x), acc), "mi355_xengine_submit")

    def wait(self, cross_correlation):
        """Block for the oldest submitted integration and return its matrix."""
        y = _host(cross_correlation, np.complex64, writable=True)
        check(self._L.mi355_xengine_wait(self._h, _hp(y)), "mi355_xengine_wait")
        return self.get_output_buffer_size()

    def pending(self):
        return self._L.mi355_xengine_pending(self._h)

    def acquire(self):
        """Zero-copy submit: the pinned frame buffer of the next free slot as a writable int8 numpy view (the
        reference's pinned char_input / complex_input, lib/clXEngine_impl.cc:325-362); fill it, then submit_acquired()."""
        p = C.c_void_p()
        check(self._L.mi355_xengine_acquire(self._h, C.byref(p)), "mi355_xengine_acquire")
        buf = (C.c_int8 * self.input_bytes()).from_address(p.value)
        return np.frombuffer(buf, dtype=np.int8)

    def submit_acquired(self, accumulator=None):
        acc = None if accumulator is None else _hp(_host(accumulator, np.complex64))
        check(self._L.mi355_xengine_submit_acquired(self._h, acc), "mi355_xengine_submit_acquired")

    def xcorrelate_device(self, input_matrix, cross_correlation, accumulate=False, stations_per_group=None):
        """Device-resident xcorrelate.  stations_per_group: the input is the receive buffer of the multi-GPU corner turn,
        [group][t][station in group][chan][pol] (gr-clenabled_amd/shard.py), read in place."""
        nin, nout = self.input_bytes(), self.get_output_buffer_size() * 8
        if stations_per_group:
            check(self._L.mi355_xengine_xcorrelate_grouped_dev(self._h, _dp(input_matrix, nin, "input"), _dp(cross_correlation, nout, "output"), 1 if accumulate else 0,
                                                               int(stations_per_group), _torch_stream(self.device)),
                  "mi355_xengine_xcorrelate_grouped_dev")
            return self.get_output_buffer_size()
        check(self._L.mi355_xengine_xcorrelate_dev(self._h, _dp(input_matrix, nin, "input"), _dp(cross_correlation, nout, "output"),
                                                   1 if accumulate else 0, _torch_stream(self.device)),
              "mi355_xengine_xcorrelate_dev")
        return self.get_output_buffer_size()

    def xcorrelate_n_device(self, nint, input_matrices, cross_correlations, accumulate=False, stations_per_group=None):
        """nint integration windows in one launch (the per-integration loop of lib/clXEngine_impl.cc:1234-1299, batched).  input_matrices:
        nint windows back to back, or with stations_per_group the receive buffer of ONE all-to-all over nint windows,
        [group][window][t][station in group][chan][pol]; cross_correlations: nint matrices back to back."""
        check(self._L.mi355_xengine_xcorrelate_n_dev(self._h, int(nint), _dp(input_matrices, int(nint) * self.input_bytes(), "input"),
                                                     _dp(cross_correlations, int(nint) * self.get_output_buffer_size() * 8, "output"), 1 if accumulate else 0,
                                                     int(stations_per_group or 0), _torch_stream(self.device)),
              "mi355_xengine_xcorrelate_n_dev")
        return nint * self.get_output_buffer_size()

    def pack3d_device(self, dst, src, width_bytes, rows, nblocks, src_pitch, src_block_stride, dst_pitch, dst_block_stride):
        """Strided device copy on this block's context (the send-side packing of the corner turn)."""
        check(self._L.mi355_pack3d_dev(self._ctx, _dp(dst), _dp(src), int(width_bytes), int(rows), int(nblocks), int(src_pitch),
                                       int(src_block_stride), int(dst_pitch), int(dst_block_stride), _torch_stream(self.device)),
              "mi355_pack3d_dev")

    def last_route(self):
        """Which kernels the last device-side call ran (mi355_xengine_last_route): a dict of mi355_xe_route's fields."""
        class Route(C.Structure):
            _fields_ = [("kernel", C.c_char * 64)] + [(k, C.c_int) for k in ("launches", "windows", "workgroups", "units_per_workgroup", "tsplit",
                                                                           "in_launch_reduce", "touches", "pace")]
        r = Route()
        check(self._L.mi355_xengine_last_route(self._h, C.byref(r)), "mi355_xengine_last_route")
        d = {k: int(getattr(r, k)) for k, _ in Route._fields_[1:]}
        d["kernel"] = r.kernel.decode()
        return d

    def selftest_scale(self):
        """Sums S in [-2^24, 2^24] whose single-precision IChar scale differs from (float)((double)S / 127 / 127): must be 0."""
        n = C.c_longlong(-1)
        check(self._L.mi355_xengine_selftest_scale(self._ctx, C.byref(n)), "mi355_xengine_selftest_scale")
        return int(n.value)

    def gather(self, nframes, frame0, input_items, frame_buffer):
        """Host frame gather of work_processor (lib/clXEngine_impl.cc:987-1061)."""
        ins = [np.ascontiguousarray(x) for x in input_items]
        p = (C.c_void_p * len(ins))(*[x.ctypes.data for x in ins])
        check(self._L.mi355_xengine_gather(self._h, int(nframes), int(frame0), p, _hp(frame_buffer)), "mi355_xengine_gather")
        return nframes


class clXEngineSharded:
    """clXEngine over several devices of one process: `world` ranks on devices `device_ids` (mi355_xengine_shard_*, the single-process
    counterpart of shard.py).  IChar only.  The reference has no such class -- it picks ONE device per block (devId,
    lib/GRCLBase.cpp:115-134); this is the form a flowgraph (one process) can use.  xcorrelate(): `windows` integration windows in the
    reference's frame layout in, `windows` triangular-order matrices out (lib/clXEngine_impl.h:179-201 over the devices)."""

    def __init__(self, device_ids, polarization, num_inputs, num_channels, integration, windows=1):
        self._L = lib()
        self._h = C.c_void_p()
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        self.world, self.npol, self.windows = len(device_ids), int(polarization), int(windows)
        self.num_inputs, self.num_channels, self.integration = int(num_inputs), int(num_channels), int(integration)
        check(self._L.mi355_xengine_shard_create(self.world, ids, self.npol, self.num_inputs, self.num_channels, self.integration, self.windows,
                                                 C.byref(self._h)), "mi355_xengine_shard_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.mi355_xengine_shard_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def frames_bytes(self):
        return self._L.mi355_xengine_shard_frames_bytes(self._h)

    def slab_items(self):
        return self._L.mi355_xengine_shard_slab_items(self._h)

    def get_output_buffer_size(self):  # items of ONE window's full matrix
        return self.slab_items() * self.world

    def device(self, rank):
        return self._L.mi355_xengine_shard_device(self._h, int(rank))

    def stream(self, rank):
        return self._L.mi355_xengine_shard_stream(self._h, int(rank))

    def xcorrelate(self, input_matrix, cross_correlation, accumulate=False):
        x = np.ascontiguousarray(input_matrix)
        need = self.windows * self.integration * self.num_inputs * self.num_channels * self.npol * 2
        if x.nbytes < need:
            raise ValueError("xcorrelate: input needs %d bytes" % need)
        y = _host(cross_correlation, np.complex64, writable=True)
        if y.size < self.windows * self.get_output_buffer_size():
            raise ValueError("xcorrelate: output needs %d items" % (self.windows * self.get_output_buffer_size()))
        check(self._L.mi355_xengine_shard_xcorrelate(self._h, _hp(x), _hp(y), 1 if accumulate else 0), "mi355_xengine_shard_xcorrelate")
        return self.windows * self.get_output_buffer_size()

    def input_bytes(self):
        return self._L.mi355_xengine_shard_input_bytes(self._h)

    def acquire(self):
        """Streaming host path: the pinned frame buffer of the next free slot (`windows` integration windows in the reference's frame layout,
        lib/clXEngine_impl.cc:325-362) as a writable int8 numpy view; fill it, then submit_acquired()."""
        p = C.c_void_p()
        check(self._L.mi355_xengine_shard_acquire(self._h, C.byref(p)), "mi355_xengine_shard_acquire")
        return np.frombuffer((C.c_int8 * self.input_bytes()).from_address(p.value), dtype=np.int8)

    def submit_acquired(self):
        """Per rank, on its own stream: upload of its antenna group out of the pinned buffer, exchange, correlation, download -- enqueue only."""
        check(self._L.mi355_xengine_shard_submit_acquired(self._h), "mi355_xengine_shard_submit_acquired")

    def wait(self, cross_correlation):
        """Block for the oldest submitted exchange; `windows` matrices."""
        y = _host(cross_correlation, np.complex64, writable=True)
        if y.size < self.windows * self.get_output_buffer_size():
            raise ValueError("wait: output needs %d items" % (self.windows * self.get_output_buffer_size()))
        check(self._L.mi355_xengine_shard_wait(self._h, _hp(y)), "mi355_xengine_shard_wait")
        return self.windows * self.get_output_buffer_size()

    def pending(self):
        return self._L.mi355_xengine_shard_pending(self._h)

    def submit_device(self, frames, outs, accumulate=False):
        """frames[r] / outs[r]: CUDA tensors on the rank's device (antenna-group frames / windows x slab matrices); enqueue only."""
        fp = (C.c_void_p * self.world)(*[_dp(t, self.frames_bytes(), "frames").value for t in frames])
        op = (C.c_void_p * self.world)(*[_dp(t, self.windows * self.slab_items() * 8, "output").value for t in outs])
        check(self._L.mi355_xengine_shard_submit_dev(self._h, fp, op, 1 if accumulate else 0), "mi355_xengine_shard_submit_dev")

    def wait_current_stream(self, rank):
        """The rank's compute stream waits for everything enqueued so far on torch's current stream of the rank's device."""
        check(self._L.mi355_xengine_shard_wait_stream(self._h, int(rank), _torch_stream(self.device(rank))), "mi355_xengine_shard_wait_stream")

    def synchronize(self):
        check(self._L.mi355_xengine_shard_synchronize(self._h), "mi355_xengine_shard_synchronize")


class _Elem(_Block):
    """Remaining elementwise family over mi355_elem_*."""
    _destroy = "mi355_elem_destroy"
    _kind = 0
    _in = ()
    _out = ()

    def _create(self, p0=0.0, p1=0.0):
        check(self._L.mi355_elem_create(self._ctx, self._kind, float(p0), float(p1), C.byref(self._h)), "mi355_elem_create")

    def history(self):
        return self._L.mi355_elem_history(self._h)

    def work(self, noutput_items, input_items, output_items):
        ins = [_host(x, t) for x, t in zip(input_items, self._in)]
        need = noutput_items + self.history() - 1
        if any(x.size < need for x in ins):
            raise ValueError("work(): need %d input items (history included)" % need)
        outs = [_host(y, t, writable=True) for y, t in zip(output_items, self._out)]
        check(self._L.mi355_elem_work(self._h, noutput_items, _hp(ins[0]), _hp(ins[1]) if len(ins) > 1 else None,
                                      _hp(outs[0]), _hp(outs[1]) if len(outs) > 1 else None), "mi355_elem_work")
        return noutput_items

    testOpenCL = work

    def work_device(self, noutput_items, input_items, output_items):
        i, o = input_items, output_items
        nin = [(int(noutput_items) + self.history() - 1) * np.dtype(t).itemsize for t in self._in]
        nout = [int(noutput_items) * np.dtype(t).itemsize for t in self._out]
        check(self._L.mi355_elem_work_dev(self._h, noutput_items, _dp(i[0], nin[0], "input 0"), _dp(i[1], nin[1], "input 1") if len(self._in) > 1 else None,
                                          _dp(o[0], nout[0], "output 0"), _dp(o[1], nout[1], "output 1") if len(self._out) > 1 else None,
                                          _torch_stream(self.device)), "mi355_elem_work_dev")
        return noutput_items


class clLog(_Elem):
    """clLog::make(openCLPlatformType, devSelector, platformId, devId, nValue, kValue, setDebug=0) -- clLog.h:49"""
    _kind, _in, _out = 1, (np.float32,), (np.float32,)

    def __init__(self, openCLPlatformType, devSelector, platformId, devId, nValue, kValue, setDebug=0):
        super().__init__(openCLPlatformType, devSelector, platformId, devId, setDebug)
        self._create(nValue, kValue)


class clSNR(_Elem):
    """clSNR::make(openCLPlatformType, devSelector, platformId, devId, nValue, kValue, setDebug=0) -- clSNR.h:49"""
    _kind, _in, _out = 2, (np.float32, np.float32), (np.float32,)

    def __init__(self, openCLPlatformType, devSelector, platformId, devId, nValue, kValue, setDebug=0):
        super().__init__(openCLPlatformType, devSelector, platformId, devId, setDebug)
        self._create(nValue, kValue)


class clComplexToMag(_Elem):
    """clComplexToMag::make(openCLPlatformType, devSelector, platformId, devId, setDebug=0) -- clComplexToMag.h:49"""
    _kind, _in, _out = 3, (np.complex64,), (np.float32,)

    def __init__(self, openCLPlatformType, devSelector, platformId, devId, setDebug=0):
        super().__init__(openCLPlatformType, devSelector, platformId, devId, setDebug)
        self._create()


class clComplexToArg(clComplexToMag):
    """clComplexToArg::make(...) -- clComplexToArg.h:49"""
    _kind = 4


class clComplexToMagPhase(clComplexToMag):
    """clComplexToMagPhase::make(...) -- clComplexToMagPhase.h:49; outputs (mag, phase)"""
    _kind, _out = 5, (np.float32, np.float32)


class clMagPhaseToComplex(clComplexToMag):
    """clMagPhaseToComplex::make(...) -- clMagPhaseToComplex.h:49; inputs (mag, phase)"""
    _kind, _in, _out = 6, (np.float32, np.float32), (np.complex64,)


class clQuadratureDemod(_Elem):
    """clQuadratureDemod::make(gain, openCLPlatformType, devSelector, platformId, devId, setDebug=0)
    -- clQuadratureDemod.h:49; history 2 (lib/clQuadratureDemod_impl.cc:81)"""
    _kind, _in, _out = 7, (np.complex64,), (np.float32,)

    def __init__(self, gain, openCLPlatformType, devSelector, platformId, devId, setDebug=0):
        super().__init__(openCLPlatformType, devSelector, platformId, devId, setDebug)
        self._create(gain, 0.0)


class clxcorrelate_fft_vcf(_Block):
    """clxcorrelate_fft_vcf::make(fftSize, num_inputs, openCLPlatformType, devSelector, platformId, devId, input_type=1)
    -- include/clenabled/clxcorrelate_fft_vcf.h:50.  Input 0 is the reference; output s-1 is the half-swapped magnitude of
    the unscaled inverse FFT of X0 * conj(Xs) (lib/clxcorrelate_fft_vcf_impl.cc:1058-1143).  input_type 1 = the inputs
    are spectra, 2 = time series (forward FFT first)."""
    _destroy = "mi355_xcorr_fft_destroy"

    def __init__(self, fftSize, num_inputs, openCLPlatformType, devSelector, platformId, devId, input_type=1):
        super().__init__(openCLPlatformType, devSelector, platformId, devId, 0)
        self.fft_size, self.num_inputs, self.input_type = int(fftSize), int(num_inputs), int(input_type)
        check(self._L.mi355_xcorr_fft_create(self._ctx, self.fft_size, self.num_inputs, self.input_type, C.byref(self._h)),
              "mi355_xcorr_fft_create")

    def work(self, noutput_items, input_items, output_items):
        if len(input_items) != self.num_inputs or len(output_items) != self.num_inputs - 1:
            raise ValueError("work(): %d inputs and %d outputs expected" % (self.num_inputs, self.num_inputs - 1))
        ins = [_host(x, np.complex64) for x in input_items]
        outs = [_host(y, np.float32, writable=True) for y in output_items]
        need = noutput_items * self.fft_size
        if any(x.size < need for x in ins) or any(y.size < need for y in outs):
            raise ValueError("work(): every buffer must hold noutput_items vectors of fft_size items")
        ip = (C.c_void_p * len(ins))(*[_hp(x) for x in ins])
        op = (C.c_void_p * len(outs))(*[_hp(y) for y in outs])
        check(self._L.mi355_xcorr_fft_work(self._h, noutput_items, ip, op), "mi355_xcorr_fft_work")
        return noutput_items

    work_test = work

    def work_device(self, noutput_items, input_items, output_items):
        ip = (C.c_void_p * len(input_items))(*[_dp(x) for x in input_items])
        op = (C.c_void_p * len(output_items))(*[_dp(y) for y in output_items])
        check(self._L.mi355_xcorr_fft_work_dev(self._h, noutput_items, ip, op, _torch_stream(self.device)), "mi355_xcorr_fft_work_dev")
        return noutput_items
