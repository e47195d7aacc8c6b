"""A ten-second run of tools/fuzz_small_batches.py (random operations, sizes around every routing boundary, random kernel-selection
options, hostile inputs) — the four-minute run is in profiles/r02/fuzz_small_batches.log."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("build", ["product", "crosscheck"])
def test_small_batch_paths_differential_soak(build):
    """the library that ships with its own options, and the cross-check build with the kernel-variant selectors in the mix"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_small_batches.py"), "8", "7", build], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all equal to the oracle" in r.stdout
