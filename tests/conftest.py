import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand from oracle/ssg_oracle.c."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def native():
    """ctypes binding of libshipsim.so; builds it when missing (hipcc cross-compiles without a GPU)."""
    from ship_sim_gym_amd import _native as N
    if not os.path.exists(N.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "ship_sim_gym_amd", "csrc"), "-s", "-j8"])
    N.lib()
    return N
