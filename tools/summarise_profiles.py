#!/usr/bin/env python3
"""Summarise gpurun_out/prof/* (tools/collect_profiles.sh) into profiles/<round>/: the rocprofv3
kernel_stats.csv of each workload verbatim, and one JSON per workload with the PMC means per dispatch
of the dominant kernel plus derived VALUBusy / bytes (FETCH_SIZE doubled: gfx950 correction,
MI355X_MICROARCH.md §HBM)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kyber_rs_amd  # noqa: E402

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/prof"
dst = os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
DOM = {"mul": "k_mul_ladder<", "mul_enc": "k_mul_ladder<", "mul_base": "k_mul_base", "sign": "k_mul_base", "verify": "k_mul_ladder<"}
ITEMS = {"mul": 1 << 20, "mul_enc": 1 << 20, "mul_base": 1 << 20, "sign": 1 << 18, "verify": 1 << 20}
for w in ("mul", "mul_enc", "mul_base", "sign", "verify"):
    ks = sorted(glob.glob(f"{src}/{w}_trace/*/*kernel_stats.csv"), key=os.path.getmtime)
    if ks:
        shutil.copyfile(ks[-1], os.path.join(dst, f"{w}_kernel_stats.csv"))      # newest run only
    out = {}
    names = set()
    for p in ("sq", "fetch", "write"):
        for f in sorted(glob.glob(f"{src}/{w}_pmc_{p}/*/*counter_collection.csv"), key=os.path.getmtime)[-1:]:
            agg = collections.defaultdict(list)
            meta = {}
            for r in csv.DictReader(open(f)):
                if DOM[w] in r["Kernel_Name"]:
                    names.add(r["Kernel_Name"].split("(")[0])
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size") if k in r}
            for k, v in agg.items():
                out[k] = {"dispatches": len(v), "mean_per_dispatch": sum(v) / len(v)}
            if meta:
                out["_dispatch"] = meta
    if not out:
        continue
    g = lambda k: out[k]["mean_per_dispatch"]
    d = {"kernel": "/".join(sorted(names)) or DOM[w].rstrip("<"), "kernel_sources_id": kyber_rs_amd.kernel_sources_id(), "note": "separate --pmc passes (SQ / FETCH_SIZE+GRBM / WRITE_SIZE+TCC), MI355X, means over the dispatches of the dominant kernel"}
    if "SQ_ACTIVE_INST_VALU" in out and "GRBM_GUI_ACTIVE" in out:
        d["VALUBusy_pct"] = 100 * g("SQ_ACTIVE_INST_VALU") * 4 / 1024 / (g("GRBM_GUI_ACTIVE") / 8)
    if "FETCH_SIZE" in out:
        d["fetch_bytes_per_dispatch_corrected_x2"] = g("FETCH_SIZE") * 1024 * 2
    if "WRITE_SIZE" in out:
        d["write_bytes_per_dispatch"] = g("WRITE_SIZE") * 1024
    if "TCC_HIT_sum" in out:
        d["L2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    out["_derived"] = d
    json.dump(out, open(os.path.join(dst, f"{w}_pmc_summary.json"), "w"), indent=1)
    print(w, json.dumps(d))
