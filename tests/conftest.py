import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_library():
    """the HIP library is git-ignored: cross-compile it (hipcc, no GPU needed) when a checkout has none"""
    lib = os.path.join(ROOT, "kyber-rs_amd", "libkyber_ed25519_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build_hip()
    return lib


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def engine():
    import kyber_rs_amd
    eng = kyber_rs_amd.Engine(0)
    yield eng
