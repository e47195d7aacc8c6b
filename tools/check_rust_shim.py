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

3. ONE code path, the same on both sides of the language border (round-4 review): for every method of the trait surface the set of `kyb_*`
   entry points the Rust method can reach — through the helper functions of its file — equals the set the C++ mirror's method of that
   name reaches (host/edwards25519.hpp, which the GPU tests drive), and nothing in the module names the reference's CPU point or its
   `ge.rs` formulas.  (This part needs no reference and also runs where /root/reference is absent.)

4. Call sites: every `ffi::kyb_*(..)` call passes as many arguments as ffi.rs declares (whose types tests/test_rust_shim.py compares with the
   header), pointer parameters get pointer-shaped arguments, and the brackets of every file balance — the little a text check can do for
   source no compiler has seen.

5. A Point's value is `ge`, `pend` and `enc` together: every writer of one writes all three (value_fields_move_together).

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


def _body_at(code, k):
    """the text between the brace at code[k] and its partner"""
    depth, j = 0, k
    while j < len(code):
        depth += code[j] == "{"
        depth -= code[j] == "}"
        if depth == 0:
            break
        j += 1
    return code[k + 1:j], j


def rust_fn_texts(text):
    """{fn name: body text} of every fn with a body in a Rust file (bodies of one name in several impl blocks — three `fmt` — are joined)"""
    code = re.sub(r"//.*", "", text)
    out = {}
    for m in re.finditer(r"\bfn\s+(\w+)", code):
        k, depth = m.end(), 0                       # the first `{` or `;` outside (), [] — `-> [u8; 32] {` has a body
        while k < len(code) and not (depth == 0 and code[k] in "{;"):
            depth += code[k] in "(["
            depth -= code[k] in ")]"
            k += 1
        if k >= len(code) or code[k] == ";":
            continue
        body, _ = _body_at(code, k)
        out[m.group(1)] = out.get(m.group(1), "") + "\n" + body
    return out


def cpp_method_texts(text, cls="Point"):
    """{method name: body text} of the methods defined inside `class <cls> { ... };` (member functions at class depth, `operator==` included)"""
    code = re.sub(r"//.*", "", text)
    m = re.search(r"\bclass\s+%s\s*\{" % cls, code)
    body, _ = _body_at(code, m.end() - 1)
    out, depth, i = {}, 0, 0
    while i < len(body):
        c = body[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
        elif depth == 0:
            f = re.match(r"(operator==|\w+)\s*\(", body[i:])
            if f and (i == 0 or not (body[i - 1].isalnum() or body[i - 1] == "_")):
                # the parameter list, then `const`, then either a body or (a declaration, an initialiser) something else
                j, d = i + f.end() - 1, 0
                while j < len(body):
                    d += body[j] == "("
                    d -= body[j] == ")"
                    if d == 0:
                        break
                    j += 1
                rest = re.match(r"\s*(?:const\s*)?\{", body[j + 1:])
                if rest:
                    k = j + 1 + rest.end() - 1
                    mb, end = _body_at(body, k)
                    out[f.group(1)] = out.get(f.group(1), "") + "\n" + mb
                    i = end + 1
                    continue
        i += 1
    return out


def reachable_symbols(fns, name, skip=()):
    """kyb_* names in the body of `name` and in the bodies of the file's own functions it calls, transitively"""
    seen, todo, syms = set(), [name], set()
    while todo:
        f = todo.pop()
        if f in seen or f not in fns:
            continue
        seen.add(f)
        syms |= set(re.findall(r"\bkyb_\w+", fns[f]))
        # a call of one of the file's own functions: bare, `self.f(`, `Self::f(`, `this->f(` — or `operand.f(` / `operand->f(` when the name
        # cannot be a method of another type (`l.data()` of a std::array, `cell.set(..)`, `bytes.hash(..)` are not ours)
        for pre, callee in re.findall(r"((?:\bself\s*\.|\bSelf\s*::|\bthis\s*->|\.|->)?)\s*\b(\w+)\s*(?:::<[^>]*>)?\(", fns[f]):
            if callee not in fns or callee in skip:
                continue
            if pre in (".", "->") and callee in AMBIGUOUS:
                continue
            todo.append(callee)
    return syms


AMBIGUOUS = {"data", "set", "get", "hash", "cmp", "eq", "fmt", "new", "from", "size", "fill", "add", "sub", "neg", "mul", "null", "base"}
# trait surface: Rust method -> the mirror's method.  (`must` / `ensure_init` / `engine_must` are error and context plumbing, not operations.)
SURFACE = [("null", "null"), ("base", "base"), ("pick", "pick"), ("set", "set"), ("embed_len", "embed_len"), ("embed", "embed"), ("data", "data"),
           ("add", "add"), ("sub", "sub"), ("neg", "neg"), ("mul", "mul"), ("marshal_binary", "marshal_binary"), ("unmarshal_binary", "unmarshal_binary"),
           ("eq", "operator=="), ("has_small_order", "has_small_order"), ("is_canonical", "is_canonical")]


def one_code_path(bad, markdown=False, rust_text=None, cpp_text=None):
    rust_text = rust_text if rust_text is not None else open(os.path.join(SHIM, "point.rs")).read()
    rust = rust_fn_texts(rust_text)
    cpp = cpp_method_texts(cpp_text if cpp_text is not None else open(os.path.join(ROOT, "kyber-rs_amd", "host", "edwards25519.hpp")).read())
    rows = []
    for rname, cname in SURFACE:
        if rname not in rust:
            bad.append(f"one code path: the Rust module has no fn {rname}")
            continue
        if cname not in cpp:
            bad.append(f"one code path: the C++ mirror has no method {cname}")
            continue
        rs, cs = reachable_symbols(rust, rname), reachable_symbols(cpp, cname)
        if rs != cs:
            bad.append(f"one code path: `{rname}` reaches {sorted(rs)} in Rust but {sorted(cs)} in the C++ mirror (`{cname}`)")
        rows.append((rname, ", ".join(f"`{x}`" for x in sorted(rs)) or "— (host only)", "same" if rs == cs else "DIFFERENT"))
    code = re.sub(r"//.*", "", rust_text + "".join(open(os.path.join(SHIM, f)).read() for f in sorted(os.listdir(SHIM)) if f.endswith(".rs") and f != "point.rs"))
    for word in ("CpuPoint", "edwards25519::point", "edwards25519::ge", "GroupElement", "hip-single-add", "pair_on_cpu"):
        if word in code:
            bad.append(f"one code path: the module names `{word}` — the reference's CPU point / formulas are a second path")
    if markdown:
        print("\n| trait method | engine entry points reached (Rust module = C++ mirror) | |\n|---|---|---|")
        for r in rows:
            print("| `" + r[0] + "` | " + r[1] + " | " + r[2] + " |")
    return rows


def _split_top(argtext):
    """top-level comma-separated pieces of an argument list (nested (), [], {}, <> of turbofish kept together)"""
    parts, depth, cur = [], 0, ""
    for ch in argtext:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        parts.append(cur)
    return [x.strip() for x in parts]


def ffi_call_sites(bad):
    """4. the module has never met a compiler: every `ffi::kyb_*(..)` call in point.rs / mod.rs must at least pass as many arguments as ffi.rs
    declares for that function, every function called must be declared, pointer arguments must be pointer-shaped (`.as_ptr()`, `.as_mut_ptr()`,
    `ptr::null*()`, `&mut x`, a `*const`/`*mut` cast) and braces / parentheses of every file must balance."""
    ffi = re.sub(r"//.*", "", open(os.path.join(SHIM, "ffi.rs")).read())
    block = ffi[ffi.index('extern "C" {'):]
    block = block[:block.index("\n}")]
    decl = {}
    for m in re.finditer(r"pub fn (kyb_\w+)\s*\(([^)]*)\)", block, flags=re.S):
        decl[m.group(1)] = [a.split(":", 1)[1].strip() for a in _split_top(m.group(2)) if ":" in a]
    calls = 0
    for fname in sorted(f for f in os.listdir(SHIM) if f.endswith(".rs")):
        text = re.sub(r"//.*", "", open(os.path.join(SHIM, fname)).read())
        for opener, closer in ("()", "[]", "{}"):
            if text.count(opener) != text.count(closer):
                bad.append(f"{fname}: {text.count(opener)} `{opener}` against {text.count(closer)} `{closer}`")
        if fname == "ffi.rs":
            continue
        for m in re.finditer(r"\bffi::(kyb_\w+)\s*\(", text):
            name, k, depth = m.group(1), m.end(), 1
            while k < len(text) and depth:
                depth += text[k] == "("
                depth -= text[k] == ")"
                k += 1
            args = _split_top(text[m.end():k - 1])
            calls += 1
            if name not in decl:
                bad.append(f"{fname}: calls ffi::{name}, which ffi.rs does not declare")
                continue
            if len(args) != len(decl[name]):
                bad.append(f"{fname}: ffi::{name} is called with {len(args)} arguments, ffi.rs declares {len(decl[name])}")
                continue
            for a, ty in zip(args, decl[name]):
                ptr_arg = bool(re.search(r"as_ptr\(\)|as_mut_ptr\(\)|null\(\)|null_mut\(\)|^&mut |as \*(const|mut) |ext_mut\(\)", a))
                if ty.startswith("*") != ptr_arg and not (ty.startswith("*") and re.fullmatch(r"\w+", a)):
                    bad.append(f"{fname}: ffi::{name}: argument `{a[:40]}` against parameter type `{ty}`")
    return calls, len(decl)


def value_fields_move_together(bad, rust_text=None):
    """5. A Point's VALUE is three fields — the limbs `ge`, the handle `pend`, the bytes it was unmarshalled from `enc` — and a writer of one
    that forgets another leaves a point whose marshal_binary / eq / has_small_order answer for some OTHER point (round-5 advice: `set` copied
    `ge` and `pend` and kept the receiver's `enc`).  Every fn of point.rs that assigns `self.ge`, `self.pend` or `self.enc` field by field must
    assign all three; whole-value writes (`*self = ...`, a `Point { .. }` literal) are fine by construction, and a struct literal that names
    one of the three must name all three or spread (`..`) from a point it means to copy."""
    text = rust_text if rust_text is not None else open(os.path.join(SHIM, "point.rs")).read()
    checked = 0
    for name, body in rust_fn_texts(text).items():
        wrote = {f for f in ("ge", "pend", "enc") if re.search(r"\bself\.%s\s*=[^=]" % f, body)}
        if wrote:
            checked += 1
            if wrote != {"ge", "pend", "enc"} and name not in ("materialize", "ext_mut"):       # (materialize keeps the bytes on purpose: same point, now with limbs; ext_mut hands `ge` to the engine call that writes it)
                bad.append(f"point.rs: fn {name} assigns self.{{{', '.join(sorted(wrote))}}} but not self.{{{', '.join(sorted({'ge', 'pend', 'enc'} - wrote))}}}")
        for lit in re.finditer(r"\bPoint\s*\{([^{}]*)\}", body):
            fields = {f for f in ("ge", "pend", "enc") if re.search(r"(?:^|[,{\s])%s\s*(?::|,|$)" % f, lit.group(1).strip())}
            spread = re.search(r"\.\.\s*\*?\w", lit.group(1))
            if fields and fields != {"ge", "pend", "enc"} and not (spread and not re.search(r"\.\.\s*\*?self\b", lit.group(1))):
                # spreading from `self` keeps the RECEIVER's other fields: then all three must be named
                bad.append(f"point.rs: fn {name} builds Point {{ {', '.join(sorted(fields))}, ..self }} without {sorted({'ge', 'pend', 'enc'} - fields)}")
            checked += 1
    return checked


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
    bad, rows = [], []
    surface = one_code_path(bad, a.markdown)
    print(f"one code path: {len(surface)} trait methods compared with the C++ mirror, symbol set by symbol set")
    calls, declared = ffi_call_sites(bad)
    print(f"ffi call sites: {calls} calls of {declared} declared entry points checked for arity and pointer shape")
    print(f"value fields: {value_fields_move_together(bad)} writers of a Point's ge / pend / enc checked for writing all three")
    if not os.path.isdir(ref_dir):
        for b in bad:
            print("MISMATCH:", b)
        print(f"reference not present: impl blocks and similarity not compared; {len(bad)} mismatches")
        return 1 if bad else 0
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
