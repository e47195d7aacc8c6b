"""The Rust module (kyber-rs_amd/rust/edwards25519_hip/, source only: no toolchain in this image) is checked as text:
* its extern "C" block declares only functions the header declares, with the same parameter and return TYPES;
* every `impl ... for Point` block of the reference exists for the module's Point with the same method names, and NO file of the module is a
  copy of a reference file (line overlap / difflib ratio below 0.30, no function body shared with its namesake) — tools/check_rust_shim.py,
  only where /root/reference is present (the build container);
* the module is FFI forwarding + delegation: it does not define a curve or a suite of its own (round 2's cloned curve.rs / suite.rs stay gone)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "kyber-rs_amd", "rust", "edwards25519_hip")


C_TO_RUST = {"uint8_t": "u8", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "size_t": "size_t", "int": "c_int", "char": "c_char",
             "void": "c_void", "double": "f64", "float": "f32", "kyb_ctx": "kyb_ctx", "kyb_group": "kyb_group"}


def _c_type_to_rust(t: str) -> str:
    """`const uint8_t*` -> `*const u8`, `kyb_ctx**` -> `*mut *mut kyb_ctx`, `size_t` -> `size_t` (pointer constness and width included)"""
    t = t.strip()
    stars = t.count("*")
    base = t.replace("*", " ").split()
    const = "const" in base
    names = [w for w in base if w not in ("const", "struct", "unsigned")]
    if not names and "unsigned" in base:
        names = ["unsigned"]
    assert len(names) == 1, t
    rust = "c_uint" if names[0] == "unsigned" or (names[0] == "int" and "unsigned" in base) else C_TO_RUST[names[0]]
    for level in range(stars):
        rust = ("*const " if const and level == 0 else "*mut ") + rust      # `const T*`: the pointee of the innermost pointer is const
    return rust


def _c_param_type(p: str) -> str:
    """the type of one C parameter declaration (`const uint8_t* scalars`, `void* stream`, `int`): drop a trailing identifier"""
    p = re.sub(r"/\*.*?\*/", "", p).strip()
    m = re.match(r"^(.*?[\s\*])(\w+)$", p)
    if m and m.group(2) not in C_TO_RUST and m.group(2) not in ("const", "unsigned"):
        return m.group(1)
    return p


def test_ffi_block_matches_the_header():
    """every function of the extern "C" block is declared in the header with the same parameter TYPES in the same order (pointer constness,
    integer width and signedness, usize against u32 — an arity check alone lets such a slip in the never-compiled Rust through) and the same
    return type"""
    hdr = open(os.path.join(ROOT, "include", "kyber_ed25519.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decl = {}
    for m in re.finditer(r"^\s*([\w\s\*]+?)\b(kyb_\w+)\s*\(([^)]*)\)\s*;", hdr, flags=re.M):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3)
        plist = [a for a in params.split(",") if a.strip() and a.strip() != "void"]
        decl[name] = (None if ret == "void" else _c_type_to_rust(ret), [_c_type_to_rust(_c_param_type(a)) for a in plist])
    ffi = open(os.path.join(SHIM, "ffi.rs")).read()
    ffi = re.sub(r"//.*", "", ffi)
    block = ffi[ffi.index('extern "C" {'):]
    block = block[:block.index("\n}")]
    fns = {}
    for m in re.finditer(r"pub fn (kyb_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        params = [re.sub(r"\s+", " ", a.split(":", 1)[1]).strip() for a in m.group(2).split(",") if a.strip()]
        fns[m.group(1)] = (None if m.group(3) is None else re.sub(r"\s+", " ", m.group(3)).strip(), params)
    assert len(fns) >= 30
    for name, (ret, params) in fns.items():
        assert name in decl, f"{name} is not declared in include/kyber_ed25519.h"
        want_ret, want = decl[name]
        assert len(want) == len(params), f"{name}: {len(params)} parameters in ffi.rs, {len(want)} in the header"
        assert params == want, f"{name}: ffi.rs has {params}, the header says {want}"
        assert ret == want_ret, f"{name}: ffi.rs returns {ret}, the header {want_ret}"
    version = int(re.search(r"#define KYB_ABI_VERSION (\d+)", hdr).group(1))
    assert f"KYB_ABI_VERSION: c_int = {version}" in ffi


def test_the_type_comparison_catches_a_slip():
    assert _c_type_to_rust("const uint8_t*") == "*const u8" and _c_type_to_rust("uint8_t*") == "*mut u8"
    assert _c_type_to_rust("kyb_ctx**") == "*mut *mut kyb_ctx" and _c_type_to_rust("const kyb_group*") == "*const kyb_group"
    assert _c_type_to_rust(_c_param_type("uint32_t index")) == "u32" and _c_type_to_rust(_c_param_type("size_t n")) == "size_t"
    assert _c_type_to_rust(_c_param_type("void* stream")) == "*mut c_void" and _c_type_to_rust(_c_param_type("const int* devices")) == "*const c_int"


def test_every_reference_impl_block_has_its_counterpart():
    if not os.path.isdir("/root/reference/src/group/edwards25519"):
        import pytest
        pytest.skip("reference sources not present on this machine")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rust_shim.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout


def test_the_module_defines_a_point_and_nothing_else():
    files = sorted(os.listdir(SHIM))
    assert files == ["ffi.rs", "mod.rs", "point.rs"], files
    src = "".join(open(os.path.join(SHIM, f)).read() for f in files)
    code = re.sub(r"//.*", "", src)
    assert not re.search(r"\b(struct|enum)\s+(Curve|Suite)\w*", code) and "impl Group for" not in code
    patch = open(os.path.join(os.path.dirname(SHIM), "kyber-rs.hip-feature.patch")).read()
    assert '+pub use super::edwards25519_hip::Point;' in patch and '+#[cfg(feature = "hip")]' in patch and "--- a/src/group.rs" in patch
    # ONE code path (round-4 review): nothing names the reference's CPU point or its formulas; pick is embed without data; the rejection loop
    # decodes and multiplies on the engine; a single add is an engine call, not the reference's ge.rs; has_small_order asks the engine
    point = open(os.path.join(SHIM, "point.rs")).read()
    code = re.sub(r"//.*", "", point)
    for word in ("CpuPoint", "GroupElement", "pair_on_cpu", "hip-single-add", "WEAK_KEYS", "fn to_cpu", "fn from_cpu"):
        assert word not in code, word
    for needle in ("self.embed(None, rand)", "ffi::kyb_decode_batch", "ffi::kyb_add_batch", "ffi::kyb_point_checks_batch", "COFACTOR_LE", "ORDER_LE", "pub fn materialize"):
        assert needle in code, needle
    # pick / embed keep the receiver's var_time flag (point.rs:90-92, 106-167 take `self`)
    assert point.count("..self }") >= 6


def test_the_symbol_comparison_with_the_cpp_mirror_catches_a_second_code_path():
    """tools/check_rust_shim.py part 3 on doctored copies: a method that loses its engine call, a method that gains one, a CPU delegation"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_rust_shim as chk
    point = open(os.path.join(SHIM, "point.rs")).read()
    bad = []
    rows = chk.one_code_path(bad)
    assert not bad and len(rows) == 16 and {r[0] for r in rows} >= {"null", "base", "pick", "set", "embed", "data", "add", "sub", "neg", "mul", "marshal_binary", "unmarshal_binary", "eq"}
    assert dict((r[0], r[1]) for r in rows)["embed"] == "`kyb_decode_batch`, `kyb_mul_batch`"
    for old, new, what in (("ffi::kyb_defer_neg(a.handle(), &mut h)", "0", "`neg`"),                                   # an operation that no longer reaches the engine
                           ("ffi::kyb_add_batch(a.as_ptr()", "ffi::kyb_sum_batch(a.as_ptr()", "`add`"),                 # ... or reaches another entry point
                           ("self.embed(None, rand)", "Self::from_limbs(CpuPoint::default().pick(rand).limbs())", "CpuPoint")):
        assert old in point
        bad = []
        chk.one_code_path(bad, rust_text=point.replace(old, new))
        assert bad and any(what in b for b in bad), (what, bad)


def test_a_writer_of_one_value_field_writes_all_three():
    """round-5 advice: `Point::set` copied `ge` and `pend` and kept the RECEIVER's `enc` — after `p.unmarshal_binary(a); p.set(&q)` the point
    held q's limbs with a's bytes, and marshal_binary / eq / has_small_order answered for a.  tools/check_rust_shim.py part 5 on the module as it
    is (clean) and on that very slip (caught); the C++ mirror's `set` is `*this = p`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_rust_shim as chk
    bad = []
    assert chk.value_fields_move_together(bad) >= 10 and not bad, bad
    text = open(os.path.join(SHIM, "point.rs")).read()
    start = text.index("fn set(&mut self, p: &Self) -> Self {")
    end = text.index("}\n", text.index("*self\n", start)) + 2
    doctored = text[:start] + "fn set(&mut self, p: &Self) -> Self {\n        self.ge = p.ge;\n        self.pend = p.pend;\n        *self\n    }\n" + text[end:]
    bad = []
    chk.value_fields_move_together(bad, doctored)
    assert len(bad) == 1 and "fn set" in bad[0] and "enc" in bad[0], bad
    cpp = open(os.path.join(ROOT, "kyber-rs_amd", "host", "edwards25519.hpp")).read()
    assert "Point set(const Point& p) { *this = p; return *this; }" in cpp


def test_the_call_site_check_catches_a_wrong_arity_and_a_missing_declaration(tmp_path, monkeypatch):
    """tools/check_rust_shim.py part 4 on a doctored copy of the module: one argument dropped from a call, one call of an undeclared entry point"""
    import shutil
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_rust_shim as chk
    bad = []
    calls, declared = chk.ffi_call_sites(bad)
    assert not bad and calls >= 25 and declared >= 50
    copy = tmp_path / "edwards25519_hip"
    shutil.copytree(SHIM, copy)
    point = (copy / "point.rs").read_text()
    assert "ffi::kyb_defer_get(self.pend, out.as_mut_ptr() as *mut i32, std::ptr::null_mut())" in point
    point = point.replace("ffi::kyb_defer_get(self.pend, out.as_mut_ptr() as *mut i32, std::ptr::null_mut())", "ffi::kyb_defer_get(self.pend, out.as_mut_ptr() as *mut i32)")
    point = point.replace("ffi::kyb_defer_mark()", "ffi::kyb_defer_marker()")
    (copy / "point.rs").write_text(point)
    monkeypatch.setattr(chk, "SHIM", str(copy))
    bad = []
    chk.ffi_call_sites(bad)
    assert any("kyb_defer_get is called with 2 arguments" in b for b in bad) and any("kyb_defer_marker" in b for b in bad), bad


def test_the_byte_constants_of_the_module_are_the_curve_s():
    """point.rs spells a few 32-byte constants out (it cannot include csrc/consts.inc): the group order, the cofactor, the neutral element's
    encoding and the y of the two classes of order-8 points that has_small_order() compares a kept encoding with — against Python integers and the
    words tools/gen_constants.py derived from the curve definition"""
    import struct
    point = open(os.path.join(SHIM, "point.rs")).read()
    def const(name):
        m = re.search(r"const %s: \[u8; 32\] = \[([^\]]*)\];" % name, point)
        assert m, name
        return bytes(int(x, 0) for x in m.group(1).split(","))
    consts = open(os.path.join(ROOT, "kyber-rs_amd", "csrc", "consts.inc")).read()
    def words(name):
        m = re.search(r"#define %s\s+\{([^}]*)\}" % name, consts)
        return struct.pack("<8I", *[int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")])
    P, L = 2**255 - 19, 2**252 + 27742317777372353535851937790883648493
    assert int.from_bytes(const("ORDER_LE"), "little") == L and int.from_bytes(const("COFACTOR_LE"), "little") == 8
    assert const("NEUTRAL_ENC") == (1).to_bytes(32, "little")
    assert const("ORDER8_Y0_LE") == words("KYB_W_ORDER8_Y0") and const("ORDER8_Y1_LE") == words("KYB_W_ORDER8_Y1")
    # and from the curve itself: a point of order 8 has y^2 = +-sqrt(-1)-ish roots of 4 P = (+-1, 0) ... checked the direct way: x^2 = (y^2 - 1) / (d y^2 + 1),
    # doubling (x, y) three times gives the neutral element, twice does not
    d = (-121665 * pow(121666, P - 2, P)) % P
    def dbl(pt):
        x, y = pt
        den = d * x * x * y * y % P
        return (2 * x * y * pow(1 + den, P - 2, P)) % P, ((y * y + x * x) * pow(1 - den, P - 2, P)) % P
    for name in ("ORDER8_Y0_LE", "ORDER8_Y1_LE"):
        y = int.from_bytes(const(name), "little")
        x2 = (y * y - 1) * pow(d * y * y + 1, P - 2, P) % P
        x = pow(x2, (P + 3) // 8, P)
        if x * x % P != x2:
            x = x * pow(2, (P - 1) // 4, P) % P
        assert x * x % P == x2, name
        p2 = dbl((x, y)); p4 = dbl(p2); p8 = dbl(p4)
        assert p8 == (0, 1) and p4 != (0, 1), name
