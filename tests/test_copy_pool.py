"""The threaded memcpy behind the host-pointer API for pageable buffers (kyber-rs_amd/csrc/host_copy_pool.h),
compiled on its own under ThreadSanitizer and hammered with random job shapes and thread counts."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_copy_pool_under_tsan(tmp_path):
    src = os.path.join(ROOT, "tests", "hostcheck", "copy_pool_test.cpp")
    exe = str(tmp_path / "copy_pool_test")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-o", exe, src])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-4000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
