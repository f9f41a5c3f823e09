import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


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
