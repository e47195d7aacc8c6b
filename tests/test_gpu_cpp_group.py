"""Builds and runs tests/cpp/test_group.cpp on the GPU: the reference's generic group test
(util/test/group_test.rs:210-555) through the C++ mirror of the trait surface, one batch-of-1 call per
trait method — then re-checks the points it logged against the oracle (compare_groups,
group_test.rs:567-585)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_group_test_through_cpp_mirror(oracle):
    src = os.path.join(ROOT, "tests", "cpp", "test_group.cpp")
    out = os.path.join(ROOT, "tests", "cpp", "_build", "test_group")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    libdir = os.path.join(ROOT, "kyber-rs_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wno-unused-function", "-o", out, src,
                           "-L", libdir, "-lkyber_ed25519_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([out], capture_output=True, text=True, timeout=300)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and r.stdout.strip().endswith("OK")
    pts = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("POINT ")]
    s1 = bytes.fromhex([ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("S1 ")][0])
    s2 = bytes.fromhex([ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("S2 ")][0])
    base = oracle.base()
    assert pts[0] == oracle.encode(base).hex()                                   # gen
    assert pts[1] == oracle.mul_base((4).to_bytes(32, "little")).hex()           # 4B
    p1 = oracle.mul_ext(s1, base)
    assert pts[2] == oracle.encode(p1).hex()                                     # s1 * B
    assert pts[3] == oracle.mul(s2, p1).hex()                                    # DH shared secret
    assert len(pts) == 4 + 5 + 2 + 5
    for h in pts[4:]:                                                            # picked / embedded points decode
        assert oracle.decode(bytes.fromhex(h))[1] == 1
