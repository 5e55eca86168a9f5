import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "radiocapture-rf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _have_gpu():
    try:
        from rcf import native
        return native.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    """-m gpu tests must FAIL (not skip) when the HIP library or the device is missing."""
    from rcf import native
    native.lib()            # raises loudly if librcf.so is missing
    assert native.device_count() > 0, "no HIP device visible: -m gpu tests need an MI355X"
    return native
