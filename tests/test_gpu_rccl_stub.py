"""The library's own RCCL call sequence (csrc/engine_group.inc: dlopen of librccl, ncclCommInitAll, one ncclBroadcast per rank inside
ncclGroupStart / ncclGroupEnd, ncclCommDestroy) needs two distinct GPUs and has run nowhere yet.  Here it runs on the ONE GPU of the test
box against a stand-in librccl.so.1 (tests/rccl_stub/rccl_stub.cpp, compiled against RCCL's real header, found through LD_LIBRARY_PATH in
front of the real library): the stand-in logs every call with its arguments and performs the broadcast as device-to-device copies, so the
test checks the argument values that travel through the library's hand-written function-pointer types — count = 335,232, dtype = ncclUint8,
root = 0, in place, each rank's call on its own device between one GroupStart / GroupEnd pair — and that the image arrives (checksum
validation inside kyb_group_create_ex, then products on the receiving rank against the oracle).  And the failure side: with
KYB_GROUP_REQUIRE_RCCL a transport that cannot be used is an ERROR naming the step; without the flag the fallback is reported, not silent."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np
import kyber_rs_amd, synth
flags = int(sys.argv[1])
out = {{}}
try:
    grp = kyber_rs_amd.Group([0, 0], flags=flags)
except kyber_rs_amd.KyberHipError as e:
    print(json.dumps({{"error": str(e)}})); sys.exit(0)
out["transport"], out["note"] = grp.transport, grp.transport_note
out["last_error"] = kyber_rs_amd.load_library().kyb_last_error().decode()
s = synth.raw256(64, 77)
out["mul_base"] = grp.mul_base(s).tobytes().hex()                       # 32 items on each rank
out["tables_equal"] = grp.engine(0).base_table().tobytes() == grp.engine(1).base_table().tobytes()
grp.close()
print(json.dumps(out))
'''


def _build_stub(tmp_path):
    so = tmp_path / "librccl.so.1"
    subprocess.check_call(["g++", "-O1", "-fPIC", "-shared", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-o", str(so),
                           os.path.join(ROOT, "tests", "rccl_stub", "rccl_stub.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
    return so


def _child(flags, env_extra):
    env = dict(os.environ, KYB_NO_TORCH="1", **env_extra)      # torch would bring its own librccl (same SONAME) into the process first
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT), str(flags)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_rccl_call_sequence_against_a_stand_in_library(tmp_path, oracle):
    _build_stub(tmp_path)
    log = tmp_path / "calls.log"
    env = {"LD_LIBRARY_PATH": f"{tmp_path}:" + os.environ.get("LD_LIBRARY_PATH", ""), "KYB_RCCL_STUB_LOG": str(log)}
    REQUIRE, EVEN = 1, 2
    out = _child(REQUIRE | EVEN, env)
    assert "error" not in out, out
    assert out["transport"] == "rccl" and out["note"] == "" and out["tables_equal"]
    want = oracle.mul_base_batch(synth.raw256(64, 77), nthreads=8).tobytes().hex()
    assert out["mul_base"] == want                              # rank 1 multiplies with the image it RECEIVED
    calls = [ln.split() for ln in log.read_text().splitlines()]
    names = [c[0] for c in calls]
    assert names == ["ncclCommInitAll", "ncclGroupStart", "ncclBroadcast", "ncclBroadcast", "ncclGroupEnd", "ncclCommDestroy", "ncclCommDestroy"], names
    assert calls[0][1:] == ["ndev=2", "devlist=0,0"]
    for rank, c in enumerate(calls[2:4]):
        kv = dict(x.split("=") for x in c[1:])
        assert kv == {"rank": str(rank), "count": "335232", "dtype": "1", "root": "0", "in_place": "1", "in_group": "1", "current_device": "0", "comm_device": "0",
                      "stream_null": "0"}, kv
    assert calls[4][1] == "pending=2"
    assert sorted(c[1] for c in calls[5:]) == ["rank=0", "rank=1"]


def test_a_broken_rccl_transport_is_an_error_when_required_and_a_reported_fallback_otherwise(tmp_path):
    _build_stub(tmp_path)
    REQUIRE, EVEN = 1, 2
    env = {"LD_LIBRARY_PATH": f"{tmp_path}:" + os.environ.get("LD_LIBRARY_PATH", ""), "KYB_RCCL_STUB_LOG": str(tmp_path / "calls.log"), "KYB_RCCL_STUB_FAIL_INIT": "5"}
    out = _child(REQUIRE | EVEN, env)
    assert "KYB_E_TRANSPORT" in out.get("error", "") and "ncclCommInitAll returned 5" in out["error"], out
    out = _child(EVEN, env)                                    # the same failure without REQUIRE: host copy, and it says why
    assert out["transport"] == "host-copy" and "ncclCommInitAll returned 5" in out["note"] and out["tables_equal"]
    assert out["last_error"].startswith("warning: table image moved by host copy") and "ncclCommInitAll returned 5" in out["last_error"]
    out = _child(REQUIRE, {})                                  # a repeated device is refused before RCCL is touched: required -> error
    assert "KYB_E_TRANSPORT" in out.get("error", "") and "repeats a device" in out["error"], out
    out = _child(0, {})
    assert out["transport"] == "host-copy" and "repeats a device" in out["note"]


def test_a_missing_rccl_library_is_a_reported_fallback_or_an_error_never_a_crash(tmp_path):
    """the message was built with two dlerror() calls, the second answers NULL -> std::string(nullptr)."""
    REQUIRE, EVEN = 1, 2
    env = {"KYB_RCCL_LIBRARY": str(tmp_path / "no_such_librccl.so.1")}
    out = _child(EVEN, env)
    assert out["transport"] == "host-copy" and "librccl could not be loaded" in out["note"] and "no_such_librccl" in out["note"] and out["tables_equal"], out
    out = _child(REQUIRE | EVEN, env)
    assert "KYB_E_TRANSPORT" in out.get("error", "") and "librccl could not be loaded" in out["error"], out
