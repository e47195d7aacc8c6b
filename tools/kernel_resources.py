#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS of every kernel in the library, from the compiler's own metadata.

  python tools/kernel_resources.py [--flags "-DX"] [--json out.json]

Compiles each kernel translation unit to gfx950 assembly (hipcc -S --cuda-device-only; no GPU needed) and reads the
`.amdhsa_*` directives and the `; ScratchSize` / `; Occupancy` comments of every kernel.  Used to keep "0 bytes of scratch" a
checked property (tests/test_build_resources.py: default-path kernels 0 scratch and within their VGPR caps, spilling variants listed) and to compare builds.
"""
import argparse, concurrent.futures, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kyber-rs_amd", "csrc")
UNITS = ["kernels_base", "kernels_base_alt", "kernels_ladder", "kernels_window", "kernels_verify", "kernels_misc", "kernels_msm", "kernels_coop"]      # cross-check build (with -DKYB_CROSSCHECK)
PRODUCT_UNITS = ["kernels_base", "kernels_ladder", "kernels_verify", "kernels_misc", "kernels_msm", "kernels_coop"]                                # csrc/Makefile without CROSSCHECK=1


def unit_asm(unit, flags):
    with tempfile.NamedTemporaryFile(suffix=".s", delete=False) as f:
        out = f.name
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
           os.path.join(CSRC, unit + ".hip")] + flags
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{unit}: {r.stderr[-2000:]}")
    txt = open(out).read()
    os.unlink(out)
    return txt


def parse(txt, unit):
    rows = []
    for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)\s*$(.*?)^\s*\.end_amdhsa_kernel", txt, re.M | re.S):
        name, body = m.group(1), m.group(2)
        def d(key, default=0):
            x = re.search(r"\.amdhsa_" + key + r" (\d+)", body)
            return int(x.group(1)) if x else default
        # the comment block that precedes the descriptor carries scratch and occupancy
        head = txt[max(0, m.start() - 3000):m.start()]
        def c(key):
            x = re.findall(r"; " + key + r": (\d+)", head)
            return int(x[-1]) if x else None
        rows.append({"unit": unit, "kernel": name, "vgpr": d("next_free_vgpr"), "agpr_offset": d("accum_offset"),
                     "sgpr": d("next_free_sgpr"), "lds": d("group_segment_fixed_size"),
                     "scratch": d("private_segment_fixed_size"), "occupancy": c("Occupancy")})
    return rows


def collect(flags=(), units=None):
    flags = list(flags)
    units = list(units if units is not None else UNITS)
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        texts = list(ex.map(lambda u: unit_asm(u, flags), units))
    rows = []
    for u, t in zip(units, texts):
        rows += parse(t, u)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", default="")
    ap.add_argument("--json", default=None)
    ap.add_argument("--product", action="store_true", help="the product library's units and flags (default: the cross-check superset needs --flags -DKYB_CROSSCHECK)")
    a = ap.parse_args()
    rows = collect(a.flags.split(), PRODUCT_UNITS if a.product else None)
    print(f"{'kernel':<64} {'vgpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>8} {'occ':>4}")
    for r in rows:
        print(f"{r['kernel'][:64]:<64} {r['vgpr']:>5} {r['sgpr']:>5} {r['lds']:>7} {r['scratch']:>8} {str(r['occupancy']):>4}")
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
