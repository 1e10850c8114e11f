import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (-m "not gpu") is ~11 min of independent oracle / emulator runs when taken one by one:
    spread it over 4 (6 on eight cores and more) workers when pytest-xdist is there and the caller did not choose.  GPU tests stay serial."""
    try:
        import xdist  # noqa: F401
    except ImportError:
        return None
    opt = config.option
    if getattr(opt, "markexpr", "") == "not gpu" and not getattr(opt, "numprocesses", None) \
            and not os.environ.get("PYTEST_XDIST_WORKER") and (os.cpu_count() or 1) >= 4:
        n = 6 if (os.cpu_count() or 1) >= 8 else 4      # (the multi-rank gloo tests start 2 - 4 processes of their own)
        opt.numprocesses = n
        opt.dist = "load"
        opt.tx = ["popen"] * n             # what xdist's own (earlier) hook derives from -n N
    return None


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
