import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# tests/ loads libc_eth_kzg_hooks.so: the product library's objects + the stage-level test hooks (csrc/Makefile).  The PRODUCT
# library libc_eth_kzg.so has no test code and is what smoke(), bench.py, the C consumers (tests/c/abi_runner.c, tools/first_result)
# and the Node binding load; tests/test_abi_exports.py checks that it exports none of the hooks.
_HOOKS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust-eth-kzg_amd", "libc_eth_kzg_hooks.so")
if os.path.exists(_HOOKS):
    os.environ.setdefault("ETH_KZG_AMD_LIB", _HOOKS)

# The GPU suite exercises the WIDEST window tables (GLV width 16: 242 GB with the commitment table) unless a test sets its own
# budget: the library's default since round 5 is a stated 108 GB (tests/test_gpu_tables.py covers that default on its own).
os.environ.setdefault("ETH_KZG_AMD_TABLE_GB", "max")


# the configuration a drop-in user gets -- ETH_KZG_AMD_TABLE_GB unset: 108 GB, nine windows of mixed 15 / 14-bit
# widths, the GLV commitment table -- is tested as hard as the widest one: the module-scoped `ctx` fixtures of test_gpu_parity.py and
# test_gpu_fullsize.py are parametrised over TABLE_BUDGETS, so every test that takes `ctx` runs on both (ids "tables-default" /
# "tables-max").  The budget is an environment setting for the fixture's LIFETIME: contexts a test creates next to `ctx` follow it
# (they share ctx's tables through the registry instead of building another 200 GB).  242 + 106 GB do not fit one GPU: pytest runs all
# tests of one parameter before it tears the fixture down and sets up the next.
TABLE_BUDGETS = ["default", "max"]


def table_budget_ctx(budget, make):
    """Generator for a module-scoped fixture: the context make() returns, created and used under the table budget `budget`."""
    saved = os.environ.get("ETH_KZG_AMD_TABLE_GB")
    if budget == "default":
        os.environ.pop("ETH_KZG_AMD_TABLE_GB", None)
    else:
        os.environ["ETH_KZG_AMD_TABLE_GB"] = budget
    c = None
    try:
        c = make()
        want = {"default": (15, 9), "max": (16, 8)}.get(budget)
        if want:
            assert (c.window_bits(), c.window_count()) == want, f"tables-{budget}: FK20 table of nominal width {c.window_bits()}"
        yield c
    finally:
        if c is not None:
            c.close()
        if saved is None:
            os.environ.pop("ETH_KZG_AMD_TABLE_GB", None)
        else:
            os.environ["ETH_KZG_AMD_TABLE_GB"] = saved


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
