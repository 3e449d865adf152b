import os
import sys

import pytest

# torch bundles its own HIP runtime; whichever runtime a process loads first serves both torch and librustray_hip.so.
# Loaded second (after the library has initialised /opt/rocm's runtime), torch finds "No HIP GPUs": import it first,
# as bench.py does, so that tests that hand torch tensors to the C ABI work whatever subset of the suite is run.
try:
    import torch  # noqa: F401
except ImportError:
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def hip():
    """The product library.  GPU tests fail loudly (no fallback) when it is missing or sees no device."""
    from rustray_amd import capi
    capi.lib()
    if capi.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box")
    return capi
