"""The reference-side half of the boundary, checked mechanically (round-4 review: the shipped "patch" named files the reference does not have).

kyber-rs_amd/rust/kyber-rs.hip-feature.patch must be a real unified diff against the reference tree: it is dry-run-applied (`patch -p1 --dry-run`
and `git apply --check`) to a scratch copy of /root/reference, applied, the module directory is dropped in where INTEGRATION.md §3 says, and — no
Rust toolchain in this image — a walk over the `mod` / `pub` / `pub use` lines of the PATCHED tree checks that every path the module `use`s is
reachable from where the module sits at the visibility the tree gives it, with `cfg(feature = "hip")` taken as set; that the name
`group::edwards25519::Point` then resolves to the module's type; that the reference's CPU point is compiled out (it would not type-check against
`constants::NULL_POINT` otherwise); and that the module's type has every method and supertrait `trait Point` demands.  INTEGRATION.md §3 shows the
patch byte for byte.  Build container only (skips where the reference is absent, e.g. on the GPU box)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
RUST = os.path.join(ROOT, "kyber-rs_amd", "rust")
PATCH = os.path.join(RUST, "kyber-rs.hip-feature.patch")
MODULE_AT = ("group", "edwards25519_hip")          # crate::group::edwards25519_hip

needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "group", "edwards25519")), reason="reference sources not present on this machine")


# ---- a very small model of rustc's module tree: enough for `mod`, `pub mod`, `pub use`, item definitions and cfg(feature = "hip") ----------
def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return "\n".join(ln.split("//")[0] for ln in text.split("\n"))


def _cfg_on(attr):
    """cfg attributes this check understands, with feature "hip" SET and `test` unset; None = not a cfg attribute"""
    m = re.fullmatch(r"#\[cfg\((.*)\)\]", attr.strip())
    if not m:
        return None
    e = m.group(1).replace(" ", "")
    return {'feature="hip"': True, 'not(feature="hip")': False, "test": False, "not(test)": True}.get(e, True)


class Tree:
    def __init__(self, src):
        self.src = src

    def file_of(self, mod):
        """module path (tuple of names below `crate`) -> its source file"""
        if not mod:
            return os.path.join(self.src, "lib.rs")
        base = os.path.join(self.src, *mod)
        for cand in (base + ".rs", os.path.join(base, "mod.rs")):
            if os.path.isfile(cand):
                return cand
        return None

    def statements(self, mod):
        """top-level statements of a module file with the cfg verdict of the attributes in front of them: [(enabled, text)]"""
        text = _strip_comments(open(self.file_of(mod)).read())
        out, depth, cur, attrs = [], 0, "", []
        for ln in text.split("\n"):
            s = ln.strip()
            if depth == 0 and s.startswith("#[") and not cur:
                attrs.append(s)
                continue
            if depth == 0 and not s:
                continue
            cur += " " + s
            depth += ln.count("{") - ln.count("}")
            if depth == 0 and (s.endswith(";") or s.endswith("}")):
                verdicts = [v for v in (_cfg_on(a) for a in attrs) if v is not None]
                out.append((all(verdicts), re.sub(r"\s+", " ", cur).strip()))
                cur, attrs = "", []
        return out

    @staticmethod
    def _use_leaves(body, prefix=()):
        """`a::{b, c::{d, self}, e as f, g::*}` -> [(('a','b'),'b'), (('a','c','d'),'d'), (('a','c'),'c'), (('a','e'),'f'), (('a','g','*'),'*')]"""
        body = body.strip()
        m = re.match(r"^([\w:]*?)(?:::)?\{(.*)\}$", body, flags=re.S)
        if m:
            head = tuple(x for x in m.group(1).split("::") if x)
            parts, depth, cur = [], 0, ""
            for ch in m.group(2):
                if ch == "," and depth == 0:
                    parts.append(cur); cur = ""
                else:
                    depth += ch == "{"; depth -= ch == "}"; cur += ch
            parts.append(cur)
            out = []
            for part in parts:
                if part.strip():
                    out += Tree._use_leaves(part, prefix + head)
            return out
        m = re.match(r"^([\w:\*]+?)(?:\s+as\s+(\w+))?$", body)
        assert m, body
        path = prefix + tuple(x for x in m.group(1).split("::") if x)
        if path[-1] == "self":
            path = path[:-1]
        return [(path, m.group(2) or path[-1])]

    def absolute(self, mod, path):
        """a `use` path as written inside module `mod` -> ('crate', ...) absolute, or None for an external crate"""
        if path[0] == "crate":
            return tuple(path[1:])
        if path[0] == "self":
            return tuple(mod) + tuple(path[1:])
        if path[0] == "super":
            up = list(mod)
            rest = list(path)
            while rest and rest[0] == "super":
                up.pop(); rest.pop(0)
            return tuple(up) + tuple(rest)
        if self.file_of(tuple(mod) + (path[0],)) is not None or any(re.match(r"(pub(\([\w:]+\))? )?mod %s\b" % path[0], t) for _, t in self.statements(mod)):
            return tuple(mod) + tuple(path)           # a child module (Rust 2018: relative paths start at the current module)
        return None

    def lookup(self, mod, name, user, seen=()):
        """how module `mod` offers `name` to code living in module `user`: 'mod' | 'item' | ('use', absolute path) | None; AssertionError if private"""
        key = (mod, name)
        if key in seen:
            return None
        for on, t in self.statements(mod):
            if not on:
                continue
            m = re.match(r"^(pub(?:\(([\w:]+)\))? )?mod (\w+)\s*[;{]", t)
            if m and m.group(3) == name:
                vis_ok = m.group(1) is not None or tuple(user[:len(mod)]) == tuple(mod)
                assert vis_ok, f"crate::{'::'.join(mod + (name,))} is a private module, not visible from crate::{'::'.join(user)}"
                return "mod"
            m = re.match(r"^(pub(?:\(([\w:]+)\))? )?(?:unsafe )?(?:struct|enum|trait|fn|const|static|type|union) (\w+)", t)
            if m and m.group(3) == name:
                assert m.group(1) is not None or tuple(user[:len(mod)]) == tuple(mod), f"crate::{'::'.join(mod + (name,))} is private"
                return "item"
            m = re.match(r"^(pub(?:\(([\w:]+)\))? )?use (.*);$", t)
            if m:
                public = m.group(1) is not None or tuple(user[:len(mod)]) == tuple(mod)
                for path, alias in self._use_leaves(m.group(3)):
                    if alias == name and public:
                        a = self.absolute(mod, path)
                        return ("use", a) if a is not None else "external"
                    if alias == "*" and public:
                        a = self.absolute(mod, path[:-1])
                        if a is not None and self.file_of(a) is not None and self.lookup(a, name, user, seen + (key,)) is not None:
                            return ("use", a + (name,))
            if re.match(r"^(pub )?(static|lazy_static)", t) or "lazy_static!" in t:
                if re.search(r"\bstatic ref %s\b" % name, t):
                    return "item"
        return None

    def resolve(self, path, user, trusted=0):
        """walks an absolute path from the crate root for code in module `user`; returns the chain of steps; raises AssertionError when it breaks.
        The first `trusted` segments came out of a `pub use` written inside the tree: their privacy was the re-exporting module's business."""
        mod, steps = (), []
        for i, name in enumerate(path):
            how = self.lookup(mod, name, user if i >= trusted else mod)
            assert how is not None, f"crate::{'::'.join(path)}: `{name}` is not declared (or is cfg'd out) in {os.path.relpath(self.file_of(mod), self.src)}"
            steps.append((mod, name, how))
            if how == "mod":
                mod = mod + (name,)
                assert self.file_of(mod) is not None, f"module crate::{'::'.join(mod)} has no file"
            elif isinstance(how, tuple):
                return steps + self.resolve(how[1] + tuple(path[i + 1:]), user, trusted=len(how[1]))
            else:
                assert i == len(path) - 1 or how == "external" or how == "item", path
                break
        return steps


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    """a scratch copy of the reference with the patch dry-run-checked, applied, and the module dropped in"""
    dst = str(tmp_path_factory.mktemp("kyber-rs") / "tree")
    shutil.copytree(REF, dst)
    for cmd in (["patch", "-p1", "--dry-run", "-i", PATCH], ["git", "apply", "--check", PATCH]):
        r = subprocess.run(cmd, cwd=dst, capture_output=True, text=True)
        assert r.returncode == 0, (cmd, r.stdout, r.stderr)
    r = subprocess.run(["patch", "-p1", "-i", PATCH], cwd=dst, capture_output=True, text=True)
    assert r.returncode == 0 and "garbage" not in r.stdout + r.stderr, r.stdout + r.stderr
    shutil.copytree(os.path.join(RUST, "edwards25519_hip"), os.path.join(dst, "src", *MODULE_AT))
    return dst


@needs_reference
def test_the_patch_is_what_the_generator_writes_and_applies_cleanly(patched):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_reference_patch.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for rel in ("build.rs", "src/group.rs", "src/group/edwards25519_hip/mod.rs", "src/group/edwards25519_hip/point.rs", "src/group/edwards25519_hip/ffi.rs"):
        assert os.path.isfile(os.path.join(patched, rel)), rel
    assert not os.path.exists(os.path.join(patched, "src", "group", "mod.rs"))          # the reference's group module is src/group.rs
    cargo = open(os.path.join(patched, "Cargo.toml")).read()
    feats = cargo[cargo.index("[features]"):].split("\n[")[0]
    assert re.search(r"^hip = \[\]$", feats, flags=re.M) and "hip-single-add" not in cargo
    # without the feature nothing is compiled differently: every changed or added line of the Rust sources sits behind a cfg on the feature
    import difflib
    for rel in ("src/group.rs", "src/group/edwards25519/mod.rs"):
        old, new = open(os.path.join(REF, rel)).read().split("\n"), open(os.path.join(patched, rel)).read().split("\n")
        ops = [d for d in difflib.ndiff(old, new) if d[:2] in ("+ ", "- ")]
        assert ops and all(d.startswith("+ ") for d in ops), ops                   # nothing removed, nothing rewritten
        added = [i for i, ln in enumerate(new) if ln not in old]
        for i in added:
            is_cfg = re.fullmatch(r'#\[cfg\((not\()?feature = "hip"\)?\)\]', new[i].strip()) is not None
            assert is_cfg or re.fullmatch(r'#\[cfg\(feature = "hip"\)\]', new[i - 1].strip()), (rel, new[i])      # a new statement exists only under the feature
        for i, ln in enumerate(new):                                               # an old statement is switched OFF only under the feature
            if ln.strip() == '#[cfg(not(feature = "hip"))]':
                assert new[i + 1] in old, new[i + 1]
    assert "CARGO_FEATURE_HIP" in open(os.path.join(patched, "build.rs")).read()


@needs_reference
def test_every_path_the_module_uses_is_reachable_in_the_patched_tree(patched):
    t = Tree(os.path.join(patched, "src"))
    checked = []
    for fname, sub in (("mod.rs", ()), ("point.rs", ("point",)), ("ffi.rs", ("ffi",))):
        user = MODULE_AT + sub
        text = _strip_comments(open(os.path.join(patched, "src", *MODULE_AT, fname)).read())
        uses = re.findall(r"^\s*(?:pub )?use\s+(.*?);", text, flags=re.S | re.M)
        assert uses, fname
        for u in uses:
            for path, alias in Tree._use_leaves(re.sub(r"\s+", " ", u)):
                if path[0] in ("core", "std", "serde"):
                    continue
                a = t.absolute(user, path)
                assert a is not None, (fname, path)
                steps = t.resolve(a, user)
                checked.append("::".join(("crate",) + a))
                assert steps
        # functions reached through a module alias: marshalling::point_marshal_to(..)
        for fn in set(re.findall(r"\bmarshalling::(\w+)\(", text)):
            t.resolve(("group", "internal", "marshalling", fn), user)
            checked.append("crate::group::internal::marshalling::" + fn)
    for want in ("crate::cipher::Stream", "crate::encoding::MarshallingError", "crate::group::internal::marshalling", "crate::group::edwards25519::Scalar",
                 "crate::group::PointError", "crate::group::PointCanCheckCanonicalAndSmallOrder", "crate::group::internal::marshalling::point_unmarshal_from_random",
                 "crate::group::edwards25519_hip::ffi"):
        assert want in checked, (want, checked)
    # the model notices a private module used from outside its parent, and a path that is cfg'd out under the feature
    with pytest.raises(AssertionError, match="private module"):
        t.resolve(("group", "internal", "marshalling"), ("share", "poly"))
    with pytest.raises(AssertionError, match="cfg'd out"):
        t.resolve(("group", "edwards25519", "point"), MODULE_AT + ("point",))
    assert "serde" in open(os.path.join(patched, "Cargo.toml")).read()
    scalar = open(os.path.join(patched, "src", "group", "edwards25519", "scalar.rs")).read()
    assert re.search(r"pub struct Scalar \{\s*pub v: \[u8; 32\],", scalar)              # the module reads `s.v`


@needs_reference
def test_under_the_feature_the_name_point_is_the_modules_type_and_the_cpu_point_is_compiled_out(patched):
    t = Tree(os.path.join(patched, "src"))
    steps = t.resolve(("group", "edwards25519", "Point"), ("group", "edwards25519", "curve"))
    assert (("group", "edwards25519_hip"), "Point", ("use", ("group", "edwards25519_hip", "point", "Point"))) in steps, steps
    assert steps[-1] == (("group", "edwards25519_hip", "point"), "Point", "item")
    # ... without the feature it is the reference's: the two statements are each other's complement
    mod_rs = open(os.path.join(patched, "src", "group", "edwards25519", "mod.rs")).read()
    assert '#[cfg(not(feature = "hip"))]\nmod point;' in mod_rs and '#[cfg(not(feature = "hip"))]\npub use point::Point;' in mod_rs
    assert '#[cfg(feature = "hip")]\npub use super::edwards25519_hip::Point;' in mod_rs
    # nothing that is still compiled names the CPU point's module (it compares `self` with constants::NULL_POINT, which has the re-exported type)
    ed = os.path.join(patched, "src", "group", "edwards25519")
    for f in sorted(os.listdir(ed)):
        if f in ("point.rs", "mod.rs") or f.endswith("_test.rs"):
            continue
        code = _strip_comments(open(os.path.join(ed, f)).read())
        assert not re.search(r"\b(super|edwards25519)::point\b", code), f
    consts = _strip_comments(open(os.path.join(ed, "constants.rs")).read())
    assert "EdPoint::default().null()" in consts                                  # what NULL_POINT needs of the type: Default and Point::null
    # the type offers what `trait Point` (group.rs) demands: every method without a default body, every supertrait
    group_rs = _strip_comments(open(os.path.join(patched, "src", "group.rs")).read())
    trait = group_rs[group_rs.index("pub trait Point:"):]
    head, body = trait[:trait.index("{")], trait[trait.index("{"):]
    depth, end = 0, 0
    for i, ch in enumerate(body):
        depth += ch == "{"; depth -= ch == "}"
        if depth == 0:
            end = i; break
    methods = set(re.findall(r"\bfn (\w+)", body[:end]))
    point_rs = _strip_comments(open(os.path.join(patched, "src", *MODULE_AT, "point.rs")).read())
    impl = point_rs[point_rs.index("impl group::Point for Point"):]
    have = set(re.findall(r"\bfn (\w+)", impl[:impl.index("\nimpl ")]))
    assert methods and methods <= have, methods - have
    derives = set(x.strip() for x in re.search(r"#\[derive\(([^)]*)\)\]\s*#\[serde[^\]]*\]\s*pub struct Point", point_rs).group(1).split(","))
    for sup in re.findall(r"\b([A-Z]\w+)\b", head.split(":", 1)[1]):
        if sup == "DeserializeOwned":
            sup = "Deserialize"
        assert sup in derives or re.search(r"impl (?:[\w:]+::)?%s for Point\b" % sup, point_rs), f"supertrait {sup} is neither derived nor implemented"


def test_integration_md_shows_the_patch_byte_for_byte():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```diff\n(.*?)```", doc, flags=re.S)
    assert open(PATCH).read() in blocks, "INTEGRATION.md §3 must show kyber-rs_amd/rust/kyber-rs.hip-feature.patch as it is"
    assert "src/group/mod.rs" not in doc and "hip-single-add" not in doc and "CpuPoint" not in doc
