import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def bh():
    """The product package with the HIP library initialised on cuda:0 (GPU tests only)."""
    import benlsip_jl_amd as pkg
    pkg.init(0)
    rc = pkg._lib.lib().bh_selftest()
    assert rc == 0, pkg._lib.lib().bh_last_error_detail()
    return pkg


@pytest.fixture(scope="session")
def ref():
    import benlsip_ref
    return benlsip_ref
