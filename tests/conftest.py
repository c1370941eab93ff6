import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes plumbing over the C-ABI library), built on demand."""
    import __graft_entry__ as ge
    p = ge.load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build_library()
    if os.environ.get("LDTEST_PGEN_PORTABLE"):   # (tests/test_pairphase.py::test_portable_bit_deposit_path re-runs the reader tests this way)
        assert p.lib().ldp_pgen_debug_force_portable(1) == 0
    return p


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    # -m gpu tests must fail loudly, never silently skip to a fallback, when the HIP path is unusable
    if pkg.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible")
    return pkg
