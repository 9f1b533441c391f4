import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure); builds oracle/libstfem_oracle.so on demand."""
    from oracle import oracle
    oracle.build()
    # the GPU box shows 256 logical CPUs but grants a share of 16: libgomp's default (one thread per
    # visible CPU, spinning) makes every small parallel loop of the oracle take seconds there
    oracle.lib().stfo_set_threads(min(8, len(os.sched_getaffinity(0))))
    return oracle


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
