"""The Rust module (kyber-rs_amd/rust/edwards25519_hip/, source only: no toolchain in this image) is checked as text:
* its extern "C" block declares only functions the header declares, with the same number of parameters;
* every `impl ... for Point / Curve / SuiteEd25519` block of the reference exists for the HIP types with the same method
  names (tools/check_rust_shim.py) — only where /root/reference is present (the build container)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "kyber-rs_amd", "rust", "edwards25519_hip")


def test_ffi_block_matches_the_header():
    hdr = open(os.path.join(ROOT, "include", "kyber_ed25519.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decl = {m.group(1): len([a for a in m.group(2).split(",") if a.strip() and a.strip() != "void"])
            for m in re.finditer(r"\b(kyb_\w+)\s*\(([^)]*)\)\s*;", hdr)}
    ffi = open(os.path.join(SHIM, "ffi.rs")).read()
    ffi = re.sub(r"//.*", "", ffi)
    block = ffi[ffi.index('extern "C" {'):]
    block = block[:block.index("\n}")]
    fns = {m.group(1): len([a for a in m.group(2).split(",") if a.strip()]) for m in re.finditer(r"pub fn (kyb_\w+)\s*\(([^)]*)\)", block, flags=re.S)}
    assert len(fns) >= 30
    for name, nargs in fns.items():
        assert name in decl, f"{name} is not declared in include/kyber_ed25519.h"
        assert decl[name] == nargs, f"{name}: {nargs} parameters in ffi.rs, {decl[name]} in the header"
    version = int(re.search(r"#define KYB_ABI_VERSION (\d+)", hdr).group(1))
    assert f"KYB_ABI_VERSION: c_int = {version}" in ffi


def test_every_reference_impl_block_has_its_counterpart():
    if not os.path.isdir("/root/reference/src/group/edwards25519"):
        import pytest
        pytest.skip("reference sources not present on this machine")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rust_shim.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout
