#!/usr/bin/env python3
"""Instruction histogram and cycle estimate of a kernel's hottest loop (the ladder step by default).

  python tools/isa_loop_stats.py [--unit kernels_ladder] [--kernel k_mul_ladderILi3] [--flags "-DX"]

Compiles one translation unit to gfx950 assembly (hipcc -S), takes the named kernel, finds the basic block with the
most v_mad_u64_u32 (the 256-step loop body) and prints its instruction mix weighted with the per-instruction issue
cycles measured by tools/microbench/valu_rates.hip (profiles/r02/valu_rates_selfconsistent_mi355x.jsonl)."""
import argparse, collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CYC = {"v_mad_u64_u32": 4.55, "v_mad_i64_i32": 4.5, "v_mul_lo_u32": 4.4, "v_lshlrev_b32": 4.43, "v_lshrrev_b64": 4.21, "v_ashrrev_i64": 4.21, "v_alignbit_b32": 4.22,
       "v_cndmask_b32": 4.25, "v_mad_u32_u24": 4.19, "v_mul_u32_u24": 4.4, "v_add3_u32": 4.25, "v_lshl_add_u32": 4.22, "v_and_or_b32": 4.23, "v_lshl_add_u64": 4.3,
       "v_mov_b64": 4.2, "v_bfe_u32": 4.49, "v_lshl_or_b32": 4.25, "v_add_lshl_u32": 4.25, "s_nop": 2.0}
DEFAULT = 2.45

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unit", default="kernels_ladder")
    ap.add_argument("--kernel", default="k_mul_ladderILi3")
    ap.add_argument("--flags", default="")
    ap.add_argument("--keep", default="/tmp/isa_loop.s")
    a = ap.parse_args()
    src = os.path.join(ROOT, "kyber-rs_amd", "csrc", a.unit + ".hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", a.keep, src] + a.flags.split()
    subprocess.check_call(cmd)
    txt = open(a.keep).read()
    m = re.search(r"^(_Z\w*%s\w*):.*?\n(.*?)\n\s*\.amdhsa_kernel \1" % re.escape(a.kernel), txt, re.S | re.M)
    if not m:
        sys.exit("kernel not found")
    body = m.group(2)
    meta = re.search(r"\.amdhsa_next_free_vgpr (\d+)", txt[m.end():m.end() + 4000])
    scratch = re.search(r"; ScratchSize: (\d+)", txt[m.start():m.end() + 6000])
    # loops = regions from a label to a LATER branch back to it; among those with the most multiply-adds per pass take the smallest
    # (the innermost loop around the arithmetic: a loop body may span several basic blocks, e.g. a uniform branch inside it)
    lines = body.split("\n")
    label_at = {m_.group(1): i for i, ln in enumerate(lines) for m_ in [re.match(r"\s*(\.LBB\d+_\d+):", ln)] if m_}
    regions = []
    for i, ln in enumerate(lines):
        m_ = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)\b", ln)
        if m_ and m_.group(1) in label_at and label_at[m_.group(1)] <= i:
            regions.append("\n".join(lines[label_at[m_.group(1)]:i + 1]))
    regions = regions or [body]
    most = max(r.count("v_mad_u64_u32") + r.count("v_mad_i64_i32") for r in regions)
    inner = [r for r in regions if r.count("v_mad_u64_u32") + r.count("v_mad_i64_i32") >= 0.9 * most]
    # an outer loop around the arithmetic loop carries the same multiply-adds: the inner one is the shortest of them
    best = min(inner, key=len)
    hist = collections.Counter()
    for ln in best.split("\n"):
        ln = ln.strip()
        if not ln or ln.startswith((";", ".", "//")) or ln.endswith(":"):
            continue
        op = ln.split()[0]
        op = re.sub(r"_e32$|_e64$|_dpp$|_sdwa$", "", op)
        hist[op] += 1
    tot = 0.0
    for op, c in hist.most_common():
        cyc = CYC.get(op, DEFAULT if op.startswith("v_") else 1.0)
        tot += c * cyc
        print(f"{c:6d}  {op:22s} x {cyc:4.2f} = {c * cyc:8.1f}")
    n = sum(hist.values())
    mads = hist["v_mad_u64_u32"] + hist["v_mad_i64_i32"]
    print(f"total {n} instructions, {mads} mads, est. {tot:.0f} issue cycles per iteration; mads = {mads * 4.55 / tot:.3f} of them"
          f"; vgpr {meta.group(1) if meta else '?'} scratch {scratch.group(1) if scratch else '?'}")

if __name__ == "__main__":
    main()
