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


_SESSION_ENGINE = []


def _option_keys(crosscheck=False):
    """the options of the product library (c_abi.inc outside `#ifdef KYB_CROSSCHECK`), or all of the cross-check build's"""
    import re
    src = open(os.path.join(ROOT, "kyber-rs_amd", "csrc", "c_abi.inc")).read()
    if not crosscheck:
        src = re.sub(r"#ifdef KYB_CROSSCHECK.*?#endif", "", src, flags=re.S)
    keys = set(re.findall(r'strcmp\(key, "([a-z_0-9.]+)"\)', src))
    return sorted(keys - {"diag.dev_kib", "diag.host_kib"})          # (read-only counters of the context's memory, not options)


@pytest.fixture(scope="session")
def engine():
    import kyber_rs_amd
    eng = kyber_rs_amd.Engine(0)
    _SESSION_ENGINE.append(eng)
    yield eng


_SESSION_XENGINE = []


@pytest.fixture(scope="session")
def xengine():
    """the CROSS-CHECK build (csrc/Makefile CROSSCHECK=1: the product's sources plus the alternative kernels and the options that select them),
    loaded beside the product library: tests that compare kernel variants with each other and with the oracle run here"""
    import kyber_rs_amd
    eng = kyber_rs_amd.Engine(0, crosscheck=True)
    _SESSION_XENGINE.append(eng)
    yield eng


@pytest.fixture(autouse=True)
def _cross_check_engine_options_are_left_as_found():
    eng = _SESSION_XENGINE[0] if _SESSION_XENGINE else None
    before = {k: eng.get_option(k) for k in _option_keys(True)} if eng else None
    yield
    eng = _SESSION_XENGINE[0] if _SESSION_XENGINE else None
    if eng is None or before is None:
        return
    after = {k: eng.get_option(k) for k in _option_keys(True)}
    changed = {k: (before[k], after[k]) for k in before if before[k] != after[k]}
    for k, (was, _now) in changed.items():
        eng.set_option(k, was)
    assert not changed, f"cross-check engine options left changed by this test (restored now): {changed}"


@pytest.fixture(autouse=True)
def _engine_options_are_left_as_found():
    """the session's engine is shared: a test that changes a kernel-selection option and does not put it back silently changes which
    kernels every later test runs (one did: the small-batch kernels stayed switched off for the rest of the session)"""
    eng = _SESSION_ENGINE[0] if _SESSION_ENGINE else None
    before = {k: eng.get_option(k) for k in _option_keys()} if eng else None
    yield
    eng = _SESSION_ENGINE[0] if _SESSION_ENGINE else None
    if eng is None:
        return
    after = {k: eng.get_option(k) for k in _option_keys()}
    if before is None:
        return
    changed = {k: (before[k], after[k]) for k in before if before[k] != after[k]}
    for k, (was, _now) in changed.items():
        eng.set_option(k, was)
    assert not changed, f"engine options left changed by this test (restored now): {changed}"
