import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Parity tests compare RGBA8 bit for bit with the oracle: new gyms default to the exact pixel arithmetic in this test
# session (MV_PIXELS_EXACT); tests/test_fast_pixels_gpu.py switches its gyms to the default product mode explicitly.
os.environ.setdefault("MV_PIXEL_MODE", "exact")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def hip():
    """The product library.  On a GPU box a missing/unloadable library is a FAILURE, never a skip."""
    import megaverse_amd.extension as ext
    ext.load_library()
    if not _has_gpu():
        pytest.skip("no HIP device in this container")
    return ext
