"""The C-ABI library loads and exports every symbol include/kyber_ed25519.h declares.  CPU only:
no compute call is made (kyb_init must FAIL here — there is no GPU and no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(crosscheck_only=False):
    """what the header declares outside (default) / inside its `#ifdef KYB_CROSSCHECK` block (the test hooks of the cross-check build)"""
    txt = open(os.path.join(ROOT, "include", "kyber_ed25519.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    guarded = "".join(re.findall(r"#ifdef KYB_CROSSCHECK(.*?)#endif", txt, flags=re.S))
    if not crosscheck_only:
        txt = re.sub(r"#ifdef KYB_CROSSCHECK.*?#endif", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kyb_[a-z0-9_]+)\s*\(", guarded if crosscheck_only else txt)))


def test_header_and_python_binding_agree():
    import kyber_rs_amd
    assert header_symbols() == sorted(kyber_rs_amd.ABI_SYMBOLS)
    assert header_symbols(crosscheck_only=True) == sorted(kyber_rs_amd.CROSSCHECK_ONLY_SYMBOLS)


def test_test_hooks_are_not_in_the_product_library():
    """round-5 review item 5: fault injection and the scratch reader shipped in the product.  Now the product exports exactly the two
    benchmark diagnostics (the chip's multiply-add rate and the in-kernel clock stamps, which bench.py's roofline needs); the fault
    injection counters, kyb_diag_scratch_read and kyb_diag_coop exist in the cross-check build only."""
    import subprocess
    libdir = os.path.join(ROOT, "kyber-rs_amd")

    def exported(lib):
        out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(libdir, lib)], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if " T " in ln and ln.split()[-1].startswith("kyb_")}
    prod = exported("libkyber_ed25519_hip.so")
    assert sorted(n for n in prod if n.startswith("kyb_diag")) == ["kyb_diag_mad_peak", "kyb_diag_wave_stamps"]
    assert prod == set(header_symbols()), sorted(prod ^ set(header_symbols()))
    strings = subprocess.run(["strings", "-n", "8", os.path.join(libdir, "libkyber_ed25519_hip.so")], capture_output=True, text=True).stdout
    assert "diag.fail_alloc_after" not in strings and "diag.fail_launch_after" not in strings and "injected launch failure" not in strings
    assert "k_coop_selftest" not in strings
    if os.path.exists(os.path.join(libdir, "libkyber_ed25519_hip_crosscheck.so")):
        cross = exported("libkyber_ed25519_hip_crosscheck.so")
        assert cross == prod | set(header_symbols(crosscheck_only=True)), sorted(cross ^ (prod | set(header_symbols(crosscheck_only=True))))


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


VARIANT_OPTIONS = ("mul.select", "mul_base.select", "mul.algo", "mul.ladder_waves", "mul_base.radix", "mul_base.block", "mul_base.block64", "mul_base.small_chunks",
                   "finish.batched", "finish.min_items", "encode.batched", "finish.four", "mul_base.quarters", "ladder.y_only", "verify.by_encoding", "verify.overlap", "mul.grid_per_cu",
                   "poly.segments", "poly.batch_segments")


def test_the_product_library_has_one_kernel_per_regime_and_no_variant_selectors():
    """round-4 review item 7: ~35 routing / variant knobs and the A/B leftover kernels shipped in the product ABI.  Now: the selectors of kernel
    and algorithm variants are parsed inside `#ifdef KYB_CROSSCHECK` only; the product library's code objects do not contain the alternative
    kernels (windowed variable base, radix-16 / -32 fixed base, fused signing, 2- / 4-wave ladder budgets, 512- / 768-thread fixed base), the
    cross-check library — same ABI, test infrastructure — does."""
    import re
    import subprocess
    src = open(os.path.join(ROOT, "kyber-rs_amd", "csrc", "c_abi.inc")).read()
    product = re.sub(r"#ifdef KYB_CROSSCHECK.*?#endif", "", src, flags=re.S)
    for key in VARIANT_OPTIONS:
        assert f'"{key}"' in src and f'"{key}"' not in product, key
    kept = set(re.findall(r'strcmp\(key, "([a-z_0-9.]+)"\)', product))
    assert kept == {"device.cus", "coop.max_items", "coop.base_max_items", "coop.decode_max_items", "coop.verify_max_items", "coop.ladder_max_items",
                    "coop.ladder_enc_max_items", "coop.share_by_load", "ladder.pair_max_items", "ladder.quad_max_items", "ladder.skip_canonical", "mul.short_scalars", "ext.projective",
                    "host.in_place", "host.zero_copy_kib", "host.pipe_chunks", "host.copy_threads", "defer.fuse", "defer.max_nodes", "defer.keep_mib",
                    "diag.dev_kib", "diag.host_kib"}, sorted(kept)
    libdir = os.path.join(ROOT, "kyber-rs_amd")
    names = {}
    for lib in ("libkyber_ed25519_hip.so", "libkyber_ed25519_hip_crosscheck.so"):
        path = os.path.join(libdir, lib)
        if not os.path.exists(path):
            import pytest
            pytest.skip(f"{lib} not built")
        out = subprocess.run(["strings", "-n", "8", path], capture_output=True, text=True).stdout
        names[lib] = set(re.findall(r"_Z\d+k_\w+", out))
    prod, cross = names["libkyber_ed25519_hip.so"], names["libkyber_ed25519_hip_crosscheck.so"]
    only_cross = {n for n in cross - prod}
    assert prod <= cross and only_cross, "the cross-check build is the product plus the variants"
    joined = " ".join(sorted(only_cross))
    for marker in ("k_mul_base32", "_Z10k_mul_baseI", "_Z6k_signI", "12k_mul_ladderILi2E", "12k_mul_ladderILi4E", "k_mul_base64ILb1ELi512E", "k_mul_base64ILb1ELi768E"):
        assert marker in joined, (marker, joined[:600])
    assert not any(("k_mul_base32" in n or "_Z10k_mul_baseI" in n or "_Z6k_signI" in n or "k_mul_ladderILi2E" in n or "ELi768E" in n) for n in prod)
    assert any("5k_mulI" in n or "_Z5k_mul" in n for n in only_cross), "the windowed variable-base kernel"
