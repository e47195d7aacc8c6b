#!/usr/bin/env python3
"""Regenerates kyber-rs_amd/rust/kyber-rs.hip-feature.patch: the change a kyber-rs maintainer applies to the reference tree (besides copying
kyber-rs_amd/rust/edwards25519_hip/ to src/group/edwards25519_hip/) to put the Ed25519 group on the engine under the cargo feature `hip`.

The edits are made on a scratch copy of the four files concerned and the patch is `diff -u` of the copy against the reference, so it is a real
unified diff with line numbers (`git apply --check` / `patch -p1 --dry-run` succeed: tests/test_reference_patch.py).  Build container only —
the reference is read here, nothing of it is stored: the patch carries the few context lines a diff needs.

  python tools/make_reference_patch.py [--reference /root/reference] [--check]      (--check: the committed patch is what would be generated)
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "kyber-rs_amd", "rust", "kyber-rs.hip-feature.patch")

BUILD_RS = '''\
// Links libkyber_ed25519_hip.so (the MI355X engine behind src/group/edwards25519_hip) when the crate is built with `--features hip`.
// KYBER_ED25519_HIP_LIB_DIR names the directory that holds the library (`python __graft_entry__.py build` of the engine's repository
// leaves it in kyber-rs_amd/).  Without the feature this script does nothing.
fn main() {
    println!("cargo:rerun-if-env-changed=KYBER_ED25519_HIP_LIB_DIR");
    if std::env::var_os("CARGO_FEATURE_HIP").is_some() {
        let dir = std::env::var("KYBER_ED25519_HIP_LIB_DIR")
            .expect("set KYBER_ED25519_HIP_LIB_DIR to the directory that holds libkyber_ed25519_hip.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=kyber_ed25519_hip");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
}
'''


def replace_once(text, old, new, where):
    assert text.count(old) == 1, f"{where}: expected exactly one occurrence of {old!r}"
    return text.replace(old, new)


def edited_tree(ref, dst):
    """the files the patch touches, edited, under dst (same relative paths)"""
    def put(rel, text):
        path = os.path.join(dst, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        open(path, "w").write(text)

    cargo = open(os.path.join(ref, "Cargo.toml")).read()
    put("Cargo.toml", replace_once(cargo, "[dependencies]\n",
                                   "[features]\n# Ed25519 curve arithmetic on an AMD MI355X through libkyber_ed25519_hip.so (src/group/edwards25519_hip)\nhip = []\n\n[dependencies]\n",
                                   "Cargo.toml"))
    assert not os.path.exists(os.path.join(ref, "build.rs")), "the reference has a build.rs now: merge by hand"
    put("build.rs", BUILD_RS)
    group = open(os.path.join(ref, "src", "group.rs")).read()
    put("src/group.rs", replace_once(group, "pub mod edwards25519;\n", "pub mod edwards25519;\n#[cfg(feature = \"hip\")]\npub mod edwards25519_hip;\n", "src/group.rs"))
    mod = open(os.path.join(ref, "src", "group", "edwards25519", "mod.rs")).read()
    # the CPU point is compiled OUT under the feature: it compares itself with constants::NULL_POINT, whose type is the re-exported name
    mod = replace_once(mod, "mod point;\n", "#[cfg(not(feature = \"hip\"))]\nmod point;\n", "edwards25519/mod.rs")
    mod = replace_once(mod, "pub use point::Point;\n",
                       "#[cfg(feature = \"hip\")]\npub use super::edwards25519_hip::Point;\n#[cfg(not(feature = \"hip\"))]\npub use point::Point;\n", "edwards25519/mod.rs")
    put("src/group/edwards25519/mod.rs", mod)
    return ["Cargo.toml", "build.rs", "src/group.rs", "src/group/edwards25519/mod.rs"]


def generate(ref):
    with tempfile.TemporaryDirectory() as tmp:
        files = edited_tree(ref, os.path.join(tmp, "b"))
        out = []
        for rel in files:
            old = os.path.join(ref, rel)
            r = subprocess.run(["diff", "-u", "--label", ("a/" + rel) if os.path.exists(old) else "/dev/null", "--label", "b/" + rel,
                                old if os.path.exists(old) else "/dev/null", os.path.join(tmp, "b", rel)], capture_output=True, text=True)
            assert r.returncode == 1, (rel, r.stderr)
            out.append(r.stdout)
        return "".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    text = generate(a.reference)
    if a.check:
        same = os.path.exists(PATCH) and open(PATCH).read() == text
        print("patch is up to date" if same else "patch differs from what tools/make_reference_patch.py generates")
        return 0 if same else 1
    open(PATCH, "w").write(text)
    print(f"wrote {PATCH}: {text.count(chr(10))} lines")
    return 0


if __name__ == "__main__":
    sys.exit(main())
