"""The C++ host layer (gr::clenabled block classes over the C ABI) and its timing CLI, the
counterpart of the reference's test-clenabled tools."""
import os
import subprocess

import pytest

from conftest import ROOT

CLI = os.path.join(ROOT, "gr-clenabled_amd", "test-clenabled-mi355")
HOSTLIB = os.path.join(ROOT, "gr-clenabled_amd", "libgnuradio-clenabled-mi355.so")


def test_host_library_exports_the_block_factories():
    assert os.path.exists(HOSTLIB), "run __graft_entry__.build()"
    syms = subprocess.run(["nm", "-DC", HOSTLIB], capture_output=True, text=True, check=True).stdout
    for cls in ("clMathOp", "clMathConst", "clFFT", "clFilter", "clComplexFilter", "clPolyphaseChannelizer", "clXEngine", "clLog",
                "clSNR", "clComplexToMag", "clComplexToArg", "clComplexToMagPhase", "clMagPhaseToComplex", "clQuadratureDemod",
                "clxcorrelate_fft_vcf"):
        assert "gr::clenabled::%s::make(" % cls in syms, cls


def test_cli_fails_loudly_without_a_gpu(pkg):
    if pkg.lib().mi355_device_count() > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([CLI, "--iterations=1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "mi355_ctx_create" in r.stderr


@pytest.mark.gpu
def test_cli_runs_every_block_and_checks_known_answers(gpu):
    r = subprocess.run([CLI, "--iterations=20"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if "MSPS" in l]
    assert len(lines) == 15 and all(l.rstrip().endswith("ok") for l in lines), r.stdout
    r = subprocess.run([CLI, "--iterations=5", "--fft-only", "--fft-size=2048", "2048"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "clFFT forward N=2048" in r.stdout  # the reference's FFTValidationTest size


@pytest.mark.gpu
def test_blocks_against_a_scheduler(gpu):
    """The caller plays GNU Radio's scheduler: io signatures, history / output multiple, consume counts of general_work() with several
    output multiples, the X-engine's "xcorr" / "sync" message ports and its stream-tag synchroniser (lib/clXEngine_impl.cc:1152-1232)."""
    r = subprocess.run([CLI, "--scheduler-contract"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "MISMATCH" not in r.stdout and r.stdout.count(" ok") >= 11, r.stdout


@pytest.mark.gpu
def test_xengine_streaming_file_sink_and_json(gpu, tmp_path):
    """work_test() streaming with ragged calls: result-handler delivery, file sink with 1 MB rollover and
    JSON sidecars (format of lib/clXEngine_impl.cc:438-465), pipeline integration."""
    import json
    import numpy as np
    r = subprocess.run([CLI, "--xengine-stream=%s" % tmp_path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok") == 3, r.stdout
    N, F, T, nint = 8, 64, 16, 60
    block = F * (N * (N + 1) // 2)
    files = sorted(p for p in os.listdir(tmp_path) if not p.endswith(".json"))
    assert files[0] == "xcorr_001" and len(files) >= 2  # 60 x 18432 B > 1 MB -> rolled over
    total = 0
    for i, name in enumerate(files):
        assert name == "xcorr_%03d" % (i + 1)
        data = np.fromfile(os.path.join(tmp_path, name), dtype=np.complex64)
        assert data.size % block == 0 and np.allclose(data, T)
        total += data.size // block
        meta = json.load(open(os.path.join(tmp_path, name + ".json")))
        assert meta["num_baselines"] == 36 and meta["channels"] == F and meta["antennas"] == N and meta["polarizations"] == 1
        assert meta["ntime"] == T and meta["samples_per_block"] == block and meta["bytes_per_block"] == block * 8
        assert meta["data_type"] == "cf32_le" and meta["data_format"] == "triangular order"
        assert meta["sync_timestamp"] == 1234567 and meta["object_name"] == "3C286" and meta["first_channel"] == 100
        assert meta["antenna_names"] == ["a%d" % k for k in range(8)]
        assert meta["first_seq_num"] == (0 if i == 0 else meta["first_seq_num"]) and meta["first_seq_num"] % T == 0
    assert total == nint
    first = np.fromfile(os.path.join(tmp_path, files[0]), dtype=np.complex64).size // block
    assert first * block * 8 >= 1000000 > (first - 1) * block * 8  # rolled exactly when >= 1 MB had been written
