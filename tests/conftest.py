import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference (test infrastructure)."""
    o = entry.load_oracle()
    o.lib()
    return o


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes over libmi355_clenabled.so)."""
    return entry.load_package()


@pytest.fixture(scope="session")
def gpu(pkg):
    """Skip-free guard: a gpu-marked test must really run on the HIP path."""
    n = pkg.lib().mi355_device_count()
    assert n > 0, "gpu test running without a visible HIP device: %s" % pkg.lib().mi355_last_error().decode()
    return pkg


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def crandn(rng, n):
    return (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)


def relerr(got, ref):
    """max |got-ref| / max |ref|: the tolerance form used throughout ('Tolerances')."""
    ref = np.asarray(ref)
    scale = float(np.abs(ref).max())
    return float(np.abs(np.asarray(got) - ref).max()) / (scale if scale > 0 else 1.0)


GPU_ARGS = (1, 2, 0, 0)  # openCLPlatformType=GPU, devSelector=SPECIFIC, platformId=0, devId=0
