#!/usr/bin/env python3
"""Method-for-method check of the Rust module kyber-rs_amd/rust/edwards25519_hip/ against the reference's own
src/group/edwards25519/{point,curve,suite}.rs: every `impl <Trait> for <Type>` block of the reference
must exist in the module for the corresponding type, with the same set of `fn` names.  Reads the reference as text
(build container only; the GPU box has no /root/reference: the test that calls this skips there).

  python tools/check_rust_shim.py [--reference /root/reference] [--markdown]
"""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "kyber-rs_amd", "rust", "edwards25519_hip")
PAIRS = [("point.rs", "point.rs", {"Point": "Point"}), ("curve.rs", "curve.rs", {"Curve": "CurveHip"}), ("suite.rs", "suite.rs", {"SuiteEd25519": "SuiteEd25519Hip"})]


def impl_blocks(text):
    """{(trait or '', type): (first line number, set of fn names)} for every impl block of a file"""
    out = {}
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"\s*impl(?:<[^>]*>)?\s+(?:([\w:]+(?:<[^>]*>)?)\s+for\s+)?(\w+)\s*\{?", lines[i])
        if m and not lines[i].lstrip().startswith("//"):
            trait, typ = (m.group(1) or ""), m.group(2)
            depth, fns, start = 0, set(), i + 1
            j = i
            seen_open = False
            while j < len(lines):
                code = lines[j].split("//")[0]
                if depth == 1 or (depth == 0 and not seen_open):
                    f = re.match(r"\s*(?:pub\s+)?(?:const\s+)?fn\s+(\w+)", code)
                    if f and seen_open:
                        fns.add(f.group(1))
                depth += code.count("{") - code.count("}")
                seen_open = seen_open or "{" in code
                if seen_open and depth == 0:
                    break
                j += 1
            key = (trait.split("::")[-1].split("<")[0], typ)
            prev = out.get(key, (start, set()))
            out[key] = (prev[0], prev[1] | fns)
            i = j
        i += 1
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--markdown", action="store_true")
    a = ap.parse_args()
    ref_dir = os.path.join(a.reference, "src", "group", "edwards25519")
    if not os.path.isdir(ref_dir):
        print("reference not present: nothing to compare")
        return 0
    bad, rows = [], []
    for ref_file, shim_file, names in PAIRS:
        ref = impl_blocks(open(os.path.join(ref_dir, ref_file)).read())
        shim = impl_blocks(open(os.path.join(SHIM, shim_file)).read())
        for (trait, typ), (line, fns) in sorted(ref.items(), key=lambda kv: kv[1][0]):
            if typ not in names:
                continue
            key = (trait, names[typ])
            if key not in shim:
                bad.append(f"{shim_file}: missing `impl {trait or '(inherent)'} for {names[typ]}` ({ref_file}:{line})")
                continue
            sline, sfns = shim[key]
            missing = fns - sfns
            if missing:
                bad.append(f"{shim_file}: `impl {trait or '(inherent)'} for {names[typ]}` lacks {sorted(missing)} ({ref_file}:{line})")
            rows.append((f"`impl {trait} for {typ}`" if trait else f"`impl {typ}`", f"{ref_file}:{line}", ", ".join(f"`{f}`" for f in sorted(fns)) or "—",
                         f"rust/edwards25519_hip/{shim_file}:{sline}", "all present" if not missing else "MISSING " + ", ".join(sorted(missing))))
    if a.markdown:
        print("| reference impl block | at | methods | module | status |\n|---|---|---|---|---|")
        for r in rows:
            print("| " + " | ".join(r) + " |")
    for b in bad:
        print("MISMATCH:", b)
    print(f"{len(rows)} impl blocks compared, {len(bad)} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
