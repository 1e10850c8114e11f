import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def emu_lib():
    """The HIP sources compiled for the host against the fiber emulator (tests/hipemu)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import build_emu
    from tidy3d_amd.lib import load_library
    return load_library(build_emu.build())


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; GPU tests fail loudly if it is missing (no fallback)."""
    from tidy3d_amd.lib import load_library
    return load_library()
