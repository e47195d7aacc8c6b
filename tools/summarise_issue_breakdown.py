#!/usr/bin/env python3
"""gpurun_out/prof_issue/* (tools/profile_issue_breakdown.sh) -> profiles/<round>/issue_breakdown.json and .md: per workload, the dominant kernel's
counters (mean per dispatch) and what they say per SIMD-cycle: share of cycles the VALU is busy, share in which a wave had an LDS / scalar / memory
instruction in flight, wait cycles per wave, LDS bank conflicts."""
import collections, csv, glob, json, os, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
tag = sys.argv[2] if len(sys.argv) > 2 else ""
src = "gpurun_out/prof_issue"
dst = os.path.join("profiles", rnd)
os.makedirs(dst, exist_ok=True)
DOM = {"mul": "k_mul_ladder<", "mul_base": "k_mul_base64", "sign": "k_mul_base64"}
res = {}
for w in ("mul_base", "mul", "sign"):
    c = {}
    for f in sorted(glob.glob(f"{src}/{w}_p*/*/*counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if DOM[w] in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                c["_kernel"] = r["Kernel_Name"].split("(")[0]
                c["_vgpr"], c["_lds"], c["_wg"] = r.get("VGPR_Count"), r.get("LDS_Block_Size"), r.get("Workgroup_Size")
        for k, v in agg.items():
            c[k] = sum(v) / len(v)
    if len(c) < 4:
        continue
    g = lambda k: c.get(k, float("nan"))
    simd_cycles = g("GRBM_GUI_ACTIVE") / 8 * 1024 if "GRBM_GUI_ACTIVE" in c else g("SQ_BUSY_CYCLES") / 8 * 4      # 1024 SIMDs x the kernel's duration in shader cycles
    d = {"kernel": c.get("_kernel"), "vgpr": c.get("_vgpr"), "lds_bytes": c.get("_lds"), "workgroup": c.get("_wg"),
         "duration_cycles": g("GRBM_GUI_ACTIVE") / 8,
         "VALU_busy_share": g("SQ_ACTIVE_INST_VALU") * 4 / simd_cycles,
         "LDS_inst_in_flight_share_of_simd_cycles": g("SQ_ACTIVE_INST_LDS") * 4 / simd_cycles,
         "scalar_inst_share": g("SQ_ACTIVE_INST_SCA") * 4 / simd_cycles, "misc_inst_share": g("SQ_ACTIVE_INST_MISC") * 4 / simd_cycles,
         "vmem_inst_share": g("SQ_ACTIVE_INST_VMEM") * 4 / simd_cycles,
         "any_inst_share": g("SQ_ACTIVE_INST_ANY") * 4 / simd_cycles,
         "valu_insts_per_wave": g("SQ_INSTS_VALU") / g("SQ_WAVES"), "lds_insts_per_wave": g("SQ_INSTS_LDS") / g("SQ_WAVES"),
         "salu_insts_per_wave": g("SQ_INSTS_SALU") / g("SQ_WAVES"), "branch_insts_per_wave": g("SQ_INSTS_BRANCH") / g("SQ_WAVES"),
         "wave_cycles_per_wave": g("SQ_WAVE_CYCLES") * 4 / g("SQ_WAVES"), "wait_inst_lds_cycles_per_wave": g("SQ_WAIT_INST_LDS") * 4 / g("SQ_WAVES"),
         "wait_inst_any_cycles_per_wave": g("SQ_WAIT_INST_ANY") * 4 / g("SQ_WAVES"), "wait_any_cycles_per_wave": g("SQ_WAIT_ANY") * 4 / g("SQ_WAVES"),
         "lds_bank_conflict_cycles_share_of_lds_active": g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE")),
         "lds_idx_active_share_of_cu_cycles": g("SQ_LDS_IDX_ACTIVE") / (g("GRBM_GUI_ACTIVE") / 8 * 256) if "GRBM_GUI_ACTIVE" in c else None,
         "ifetch_per_wave": g("SQ_IFETCH") / g("SQ_WAVES")}
    res[w] = {"derived": d, "raw_mean_per_dispatch": {k: v for k, v in c.items() if not k.startswith("_")}}
name = "issue_breakdown" + (("_" + tag) if tag else "")
json.dump(res, open(os.path.join(dst, name + ".json"), "w"), indent=1)
keys = list(next(iter(res.values()))["derived"].keys()) if res else []
with open(os.path.join(dst, name + ".md"), "w") as f:
    f.write("| quantity | " + " | ".join(res) + " |\n|---|" + "---|" * len(res) + "\n")
    for k in keys:
        f.write(f"| {k} | " + " | ".join((f"{res[w]['derived'][k]:.4g}" if isinstance(res[w]['derived'][k], float) else str(res[w]['derived'][k])) for w in res) + " |\n")
print(open(os.path.join(dst, name + ".md")).read())
os.makedirs("gpurun_out/profiles_" + rnd, exist_ok=True)
import shutil
for ext in (".md", ".json"):
    shutil.copyfile(os.path.join(dst, name + ext), os.path.join("gpurun_out/profiles_" + rnd, name + ext))
