"""CPU: the C-ABI library loads and exports every symbol include/mi355_clenabled.h
declares; argument validation that needs no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "mi355_clenabled.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(pkg):
    L = pkg.lib()
    names = _declared()
    assert len(names) >= 45
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, "declared in the header but not exported: %s" % missing
    # and the Python binding table covers the header exactly
    assert sorted(L._declared) == names


def test_version_and_strerror(pkg):
    L = pkg.lib()
    assert b"gfx950" in L.mi355_version()
    assert L.mi355_strerror(0) == b"ok"
    assert L.mi355_strerror(-3) == b"unsupported configuration"


def test_fft_plan_text_covers_every_length_class(pkg):
    """mi355_fft_plan_text: the path a clFFT length takes (planning only -- runs without a device).  Every 2-3-5-7 length up to 15360 (14336 with a factor 7)
    that is not a power of two gets a mixed-radix plan whose radices multiply to the length and fit 1024 threads of <= 16 values;
    lengths with a larger prime factor, or longer ones, are chirp-z (the reference's clFFT plans: lib/clFFT_impl.cc:91-128)."""
    import ctypes as C
    L = pkg._lib.lib()
    buf = C.create_string_buffer(256)

    def plan(n):
        rc = L.mi355_fft_plan_text(n, buf, 256)
        return rc, buf.value.decode()

    assert plan(4096) == (0, "one pass")
    assert plan(65536) == (0, "two tile passes 256 x 256")
    assert plan(131072) == (0, "two tile passes 256 x 512")
    assert plan(1 << 22) == (0, "four passes")
    assert plan(1000) == (0, "mixed radix 10 x 10 x 10")
    assert plan(4099) == (0, "chirp-z, m = 16384 (fused)")  # a prime
    assert plan(20000) == (0, "mixed radix, two passes 125 x 160")  # longer than a workgroup holds: two passes
    assert plan(921600) == (0, "mixed radix, two passes 960 x 960") and plan(1000000)[1].startswith("chirp-z")  # beyond 960 x 960
    assert plan(11 * 1024) == (0, "mixed radix 11 x 16 x 8 x 8") and plan(13 * 1024)[1].startswith("mixed radix 13")
    assert plan(17 * 1024)[1].startswith("chirp-z")          # a prime factor above 13 (clFFT itself refuses those)
    assert plan(11 * 1024 + 11)[1].startswith("chirp-z")     # 11 values per thread with a radix 11: 1024 threads hold 11264
    assert plan(1)[0] != 0 and plan((1 << 24) + 2)[0] != 0 and plan((1 << 23) + 1)[0] != 0
    allowed = {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16}
    count = 0
    for n in range(6, 15361):
        m = n
        for f in (2, 3, 5, 7):
            while m % f == 0:
                m //= f
        if m != 1 or n & (n - 1) == 0 or n == 7:  # (7 alone has no two-pass factorisation: chirp-z)
            continue
        rc, text = plan(n)
        if n % 7 == 0 and n > 14 * 1024:  # a radix-7 (or 14) pass leaves a thread 14 values: 1024 threads hold 14336
            assert rc == 0 and text.startswith("chirp-z"), (n, text)
            continue
        assert rc == 0 and text.startswith("mixed radix "), (n, text)
        radices = [int(v) for v in text[len("mixed radix "):].split(" x ")]
        prod = 1
        for r in radices:
            prod *= r
        assert prod == n and set(radices) <= allowed and len(radices) >= 2, (n, radices)
        assert min((16 // r) * r for r in radices) * 1024 >= n, (n, radices)
        count += 1
    assert count > 350


def test_cpu_device_type_is_refused_not_emulated(pkg):
    """OCLTYPE_CPU (3) must NOT fall back to a host implementation."""
    L = pkg.lib()
    ctx = C.c_void_p()
    assert L.mi355_ctx_create(3, 1, 0, 0, 0, C.byref(ctx)) == -3
    assert not ctx.value
    assert L.mi355_ctx_create(9, 1, 0, 0, 0, C.byref(ctx)) == -1
    assert L.mi355_ctx_create(1, 7, 0, 0, 0, C.byref(ctx)) == -1


def test_log_callback_receives_errors_and_can_be_removed(pkg):
    """mi355_set_log_callback: the sink the block layer hands GNU Radio's logger to (GR_LOG_ERROR of lib/clXEngine_impl.cc:107)."""
    L = pkg.lib()
    got = []
    pkg.set_log_callback(lambda level, msg: got.append((level, msg)))
    try:
        ctx = C.c_void_p()
        assert L.mi355_ctx_create(3, 1, 0, 0, 0, C.byref(ctx)) == -3
        assert got == [(3, "OCLTYPE_CPU requested: this library has no CPU path")]  # MI355_LOG_ERROR, same text as last_error
        assert L.mi355_last_error().decode() == got[0][1]
    finally:
        pkg.set_log_callback(None)
    assert L.mi355_ctx_create(3, 1, 0, 0, 0, C.byref(ctx)) == -3
    assert len(got) == 1


@pytest.mark.gpu
def test_debug_contexts_log_through_the_callback(gpu):
    """setDebug: what the reference prints to std::cout (lib/GRCLBase.cpp:96-120) arrives as INFO lines; silent without debug."""
    import numpy as np
    got = []
    gpu.set_log_callback(lambda level, msg: got.append((level, msg)))
    try:
        gpu.clFFT(4096, gpu.CLFFT_FORWARD, list(np.ones(4096, np.float32)), gpu.DTYPE_COMPLEX, 1, 2, 0, 0, 0, 1, True)
        assert got == []
        gpu.clFFT(4096, gpu.CLFFT_FORWARD, list(np.ones(4096, np.float32)), gpu.DTYPE_COMPLEX, 1, 2, 0, 0, 1, 1, True)
        gpu.clFilter(1, 2, 0, 0, 1, [0.1] * 65, 1, 1, False)
        gpu.clXEngine(1, 2, 0, 0, True, gpu.DTYPE_BYTE, 1, 8, 1, 0, 16, 32, [])
    finally:
        gpu.set_log_callback(None)
    text = [m for lvl, m in got if lvl == 1]
    assert any(m.startswith("context on device 0 (gfx950") for m in text)
    assert any(m.startswith("clFFT: 4096 points, forward") for m in text)
    assert any(m.startswith("clFilter: 65 real taps") and "transform size 256" in m for m in text)
    assert any(m.startswith("clXEngine: 8 inputs x 1 pol, 16 channels, 32 frames") for m in text)


def test_block_constructor_errors_mirror_reference(pkg):
    # lib/clFFT_impl.cc:74-76 -> runtime_error before any device work
    with pytest.raises(RuntimeError):
        pkg.clFFT(64, pkg.CLFFT_FORWARD, [1.0] * 63, pkg.DTYPE_COMPLEX, 1, 1, 0, 0)
    # lib/clPolyphaseChannelizer_impl.cc:59-62 -> invalid_argument
    with pytest.raises(ValueError):
        pkg.clPolyphaseChannelizer(1, 1, 0, 0, [1.0] * 8, 10, 4, 4, [0])
    # lib/clXEngine_impl.cc:106-109 -> out_of_range
    with pytest.raises(IndexError):
        pkg.clXEngine(1, 1, 0, 0, False, pkg.DTYPE_BYTE, 1, 1, 1, 0, 16, 16)


def test_no_gpu_means_loud_failure(pkg):
    """Without a device every block constructor raises; nothing silently computes on the CPU."""
    if pkg.lib().mi355_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.Mi355Error):
        pkg.clMathOp(pkg.DTYPE_COMPLEX, 1, 1, 0, 0, pkg.MATHOP_MULTIPLY)


def test_dropin_python_module_name():
    """`import clenabled` (the reference's module name) resolves to the MI355X block classes."""
    import subprocess
    import sys
    code = ("import clenabled as c; names = ['clMathOp','clMathConst','clFFT','clFilter','clComplexFilter','clPolyphaseChannelizer',"
            "'clXEngine','clLog','clSNR','clComplexToMag','clComplexToArg','clComplexToMagPhase','clMagPhaseToComplex',"
            "'clQuadratureDemod','clxcorrelate_fft_vcf','CLFFT_FORWARD','DTYPE_COMPLEX','MATHOP_MULTIPLY'];"
            "missing = [n for n in names if not hasattr(c, n)]; assert not missing, missing; print('ok')")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "gr-clenabled_amd", "python"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_device_paths_refuse_short_tensors(gpu):
    """The C ABI takes plain device pointers; the Python mirror knows the tensors and refuses the ones that are too short for the
    call (a short buffer is otherwise a memory fault on the device), for every block of the hot path."""
    import torch
    from conftest import GPU_ARGS
    z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device="cuda")
    fft = gpu.clFFT(1024, gpu.CLFFT_FORWARD, [], gpu.DTYPE_COMPLEX, *GPU_ARGS)
    with pytest.raises(ValueError):
        fft.work_device(4, [z(4 * 1024 * 2)], [z(4 * 1024 * 2 - 2)])
    mul = gpu.clMathOp(gpu.DTYPE_COMPLEX, *GPU_ARGS, gpu.MATHOP_MULTIPLY)
    with pytest.raises(ValueError):
        mul.work_device(100, [z(200), z(198)], [z(200)])
    fil = gpu.clFilter(*GPU_ARGS, 4, np.ones(9, np.float32), 1, 0, True)
    with pytest.raises(ValueError):
        fil.work_device(100, [z(2 * (400 + 8) - 2)], [z(200)])  # the history-prefixed input is one item short
    with pytest.raises(ValueError):
        fil.work_device(100, [z(2 * (400 + 8))], [z(198)])
    assert fil.work_device(100, [z(2 * (400 + 8))], [z(200)]) == 100
    pfb = gpu.clPolyphaseChannelizer(*GPU_ARGS, np.ones(64, np.float32), 256, 8, 8, list(range(8)))
    with pytest.raises(ValueError):
        pfb.work_device([z(2 * (3 * 256 + 56))], [z(2 * 3 * 256 - 2)], nbuf=3)
    with pytest.raises(ValueError):
        pfb.work_device([z(2 * (3 * 256 + 56) - 2)], [z(2 * 3 * 256)], nbuf=3)
    assert pfb.work_device([z(2 * (3 * 256 + 56))], [z(2 * 3 * 256)], nbuf=3) == 3 * 256
    xe = gpu.clXEngine(*GPU_ARGS, False, gpu.DTYPE_BYTE, 1, 4, 1, 0, 8, 32, [])
    vis = z(2 * xe.get_output_buffer_size())
    with pytest.raises(ValueError):
        xe.xcorrelate_device(z(xe.input_bytes() - 1, torch.int8), vis)
    with pytest.raises(ValueError):
        xe.xcorrelate_n_device(2, z(2 * xe.input_bytes(), torch.int8), vis)
    arg = gpu.clQuadratureDemod(1.0, *GPU_ARGS)
    with pytest.raises(ValueError):
        arg.work_device(10, [z(2 * 10)], [z(10)])  # history 2: eleven input items
    torch.cuda.synchronize()
