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
    for cls in ("clMathOp", "clMathConst", "clFFT", "clFilter", "clComplexFilter", "clPolyphaseChannelizer", "clXEngine"):
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
    assert len(lines) == 7 and all(l.rstrip().endswith("ok") for l in lines), r.stdout
    r = subprocess.run([CLI, "--iterations=5", "--fft-only", "--fft-size=2048", "2048"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "clFFT forward N=2048" in r.stdout  # the reference's FFTValidationTest size
