#!/usr/bin/env python3
"""Checks of the Rust module kyber-rs_amd/rust/edwards25519_hip/ against the reference's src/group/edwards25519/ (read as text; build
container only — the GPU box has no /root/reference and the test that calls this skips there):

1. every `impl <Trait> for Point` block of the reference's point.rs exists in the module's point.rs with the same set of `fn` names
   (the module must be a complete stand-in for the type);
2. the module holds FFI forwarding and delegation, NOT a second copy of the reference (round-2 review: curve.rs / suite.rs were clones,
   point.rs carried transliterated host logic).  For every .rs file of the module against every .rs file of the reference directory:
     * line overlap  = module code lines that also occur in the reference file / module code lines                             < 0.30
     * difflib ratio of the two sequences of code lines                                                                        < 0.30
       (code line: comments stripped, whitespace normalised, at least one letter or digit — a lone `}` or `);` is not evidence of
       anything; what remains in common for point.rs are the trait signatures an `impl group::Point for Point` must spell out)
     * no function of the module has a body (> 2 code lines) equal to the body of a same-named function of the reference.

  python tools/check_rust_shim.py [--reference /root/reference] [--markdown]
"""
import argparse
import difflib
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "kyber-rs_amd", "rust", "edwards25519_hip")
MAX_SIMILARITY = 0.30


def strip(text):
    """code lines: comments and blank lines removed, inner whitespace collapsed"""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for ln in text.split("\n"):
        ln = re.sub(r"//.*", "", ln)
        ln = re.sub(r"\s+", " ", ln).strip()
        if ln and re.search(r"[A-Za-z0-9]", ln):
            out.append(ln)
    return out


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


def fn_bodies(text):
    """{fn name: [normalised body lines]} (the last definition of a name wins; nested braces followed)"""
    code = "\n".join(strip(text))
    out = {}
    for m in re.finditer(r"\bfn\s+(\w+)", code):
        k = code.find("{", m.end())
        semi = code.find(";", m.end())
        if k < 0 or (0 <= semi < k):
            continue                      # a declaration without a body
        depth, j = 0, k
        while j < len(code):
            depth += code[j] == "{"
            depth -= code[j] == "}"
            if depth == 0:
                break
            j += 1
        out[m.group(1)] = [ln.strip() for ln in code[k + 1:j].split("\n") if ln.strip()]
    return out


def similarity(shim_text, ref_text):
    a, b = strip(shim_text), strip(ref_text)
    if len(a) < 8:                        # a three-line mod.rs shares `mod point;` with anybody's
        return 0.0, 0.0
    ref_set = set(b)
    overlap = sum(1 for ln in a if ln in ref_set) / len(a)
    return overlap, difflib.SequenceMatcher(None, a, b, autojunk=False).ratio()


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
    # 1. the stand-in is complete
    ref = impl_blocks(open(os.path.join(ref_dir, "point.rs")).read())
    shim = impl_blocks(open(os.path.join(SHIM, "point.rs")).read())
    for (trait, typ), (line, fns) in sorted(ref.items(), key=lambda kv: kv[1][0]):
        if typ != "Point":
            continue
        key = (trait, "Point")
        if key not in shim:
            bad.append(f"point.rs: missing `impl {trait or '(inherent)'} for Point` (point.rs:{line})")
            continue
        sline, sfns = shim[key]
        missing = fns - sfns
        if missing:
            bad.append(f"point.rs: `impl {trait or '(inherent)'} for Point` lacks {sorted(missing)} (point.rs:{line})")
        rows.append((f"`impl {trait} for Point`" if trait else "`impl Point`", f"point.rs:{line}", ", ".join(f"`{f}`" for f in sorted(fns)) or "—",
                     f"rust/edwards25519_hip/point.rs:{sline}", "all present" if not missing else "MISSING " + ", ".join(sorted(missing))))
    # 2. ... and is not a copy
    ref_files = sorted(f for f in os.listdir(ref_dir) if f.endswith(".rs") and f != "constants.rs")      # (constants.rs is a 3,700-line literal table)
    sims = []
    for sf in sorted(f for f in os.listdir(SHIM) if f.endswith(".rs")):
        st = open(os.path.join(SHIM, sf)).read()
        sb = fn_bodies(st)
        for rf in ref_files:
            rt = open(os.path.join(ref_dir, rf)).read()
            ov, ratio = similarity(st, rt)
            sims.append((sf, rf, ov, ratio))
            if ov >= MAX_SIMILARITY or ratio >= MAX_SIMILARITY:
                bad.append(f"{sf} is too close to the reference's {rf}: line overlap {ov:.2f}, difflib ratio {ratio:.2f} (limit {MAX_SIMILARITY})")
            for name, body in fn_bodies(rt).items():
                if name in sb and len(body) > 2 and sb[name] == body:
                    bad.append(f"{sf}: fn {name} has the body of {rf}'s fn {name} ({len(body)} lines): forward or delegate instead")
    if a.markdown:
        print("| reference impl block | at | methods | module | status |\n|---|---|---|---|---|")
        for r in rows:
            print("| " + " | ".join(r) + " |")
    worst = sorted(sims, key=lambda t: -max(t[2], t[3]))[:4]
    for sf, rf, ov, ratio in worst:
        print(f"similarity {sf} vs {rf}: line overlap {ov:.2f}, difflib ratio {ratio:.2f}")
    for b in bad:
        print("MISMATCH:", b)
    print(f"{len(rows)} impl blocks compared, {len(sims)} file pairs measured, {len(bad)} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
