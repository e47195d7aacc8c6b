"""Engine lifecycle in a fresh process: init -> work -> shutdown -> init again -> work; a second init on another
device index is refused; calls after shutdown report KYB_E_NOT_INIT."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import kyber_rs_amd, oracle_lib, synth
orc = oracle_lib.Oracle()
s = synth.scalars(5000, 77)
want = orc.mul_base_batch(s, nthreads=8)
lib = kyber_rs_amd.load_library()
assert lib.kyb_abi_version() == kyber_rs_amd.ABI_VERSION
for cycle in range(3):
    eng = kyber_rs_amd.Engine(0)
    assert np.array_equal(eng.mul_base(s), want)
    big = np.tile(s, (40, 1))                       # 200,000 items: pipelined host path, bounce buffers, copy threads
    assert np.array_equal(eng.mul_base(big)[-5000:], want)
    assert lib.kyb_init(0) == 0                     # idempotent
    assert lib.kyb_init(1) == -2                    # one process per GPU
    lib.kyb_shutdown()
    out = np.zeros((1, 32), dtype=np.uint8)
    assert lib.kyb_mul_base_batch(s.ctypes.data_as(ctypes.c_void_p), 1, out.ctypes.data_as(ctypes.c_void_p), None) == -1
lib.kyb_shutdown()                                  # twice is harmless
print("LIFECYCLE OK")
"""


def test_init_shutdown_cycles():
    r = subprocess.run([sys.executable, "-c", SCRIPT % (ROOT, os.path.join(ROOT, "tests"))], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "LIFECYCLE OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
