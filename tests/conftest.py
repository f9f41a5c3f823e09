import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# The GPU suite exercises the WIDEST window tables (GLV width 16: 242 GB with the commitment table) unless a test sets its own
# budget: the library's default since round 5 is a stated 108 GB (tests/test_gpu_tables.py covers that default on its own).
os.environ.setdefault("ETH_KZG_AMD_TABLE_GB", "max")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    o = Oracle(use_precomp=True, threads=min(8, os.cpu_count() or 1))
    yield o
    o.close()


@pytest.fixture(scope="session")
def oracle_noprecomp():
    from oracle_lib import Oracle
    o = Oracle(use_precomp=False, threads=min(8, os.cpu_count() or 1))
    yield o
    o.close()
