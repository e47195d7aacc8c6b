"""The C-ABI library loads and exports every symbol include/kyber_ed25519.h declares.  CPU only:
no compute call is made (kyb_init must FAIL here — there is no GPU and no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "kyber_ed25519.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kyb_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_python_binding_agree():
    import kyber_rs_amd
    assert header_symbols() == sorted(kyber_rs_amd.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    import kyber_rs_amd
    lib = kyber_rs_amd.load_library()
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_no_cpu_fallback_without_a_gpu():
    import torch
    import kyber_rs_amd
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present; the failure path is for GPU-less hosts")
    lib = kyber_rs_amd.load_library()
    assert lib.kyb_init(0) == -3          # KYB_E_NO_DEVICE
    assert b"no CPU path" in lib.kyb_last_error()
    out = ctypes.create_string_buffer(32)
    assert lib.kyb_mul_base_batch(bytes(32), 1, out, None) == -1    # KYB_E_NOT_INIT
    with pytest.raises(kyber_rs_amd.KyberHipError):
        kyber_rs_amd.Engine(0)


def test_product_tree_does_not_reference_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    bad = []
    for base in ("kyber-rs_amd", "kyber_rs_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".cpp", ".inc", ".rs")):
                    t = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"oracle/|liboracle|oracle_lib|bigint_model", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_header_is_plain_c(tmp_path):
    """the boundary is a C ABI: the header must compile as C99 (no C++-isms, no torch / HIP types)"""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "kyber_ed25519.h"\nint main(void) { return (int)(KYB_BASE_TABLE_BYTES == 0); }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
    txt = open(os.path.join(ROOT, "include", "kyber_ed25519.h")).read()
    code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    assert not re.search(r"hipStream_t|hipError_t|torch|at::|std::|#include\s*<hip", code)


def test_every_engine_option_is_documented_in_the_header():
    """kyb_set_option keys (c_abi.inc) and the option list of include/kyber_ed25519.h name the same options (the round-1 review
    found the two drifting apart)"""
    import re
    hdr = open(os.path.join(ROOT, "include", "kyber_ed25519.h")).read()
    src = open(os.path.join(ROOT, "kyber-rs_amd", "csrc", "c_abi.inc")).read()
    keys = set(re.findall(r'strcmp\(key, "([a-z_0-9.]+)"\)', src))
    assert len(keys) > 20
    listed = set(re.findall(r"^ \*\s+([a-z_]+\.[a-z_0-9]+)\b", hdr, re.M))
    mentioned = set(re.findall(r"\b([a-z_]+\.[a-z_0-9]+)\b", hdr))
    assert not (keys - mentioned), f"options without a word in the header: {sorted(keys - mentioned)}"
    assert not {k for k in listed if k not in keys and not k.endswith(".h") and not k.endswith(".rs")}, "the header lists an option the engine does not know"
