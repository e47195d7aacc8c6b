#!/bin/bash
# Run ON THE GPU BOX.  Round-5 review item 6: 28 % of the fixed-base kernel's LDS-active cycles are bank conflicts.  They come from the thirty
# ds_bpermute_b32 of a window, not from its nine row reads (which are conflict-free by the lane-group tables of MI355X_MICROARCH.md "LDS"):
# ds_bpermute banks by SOURCE LANE mod 32, a lane group of 32 requesters names sources in 0..63 (entry index | sign << 5), and two requesters
# that want the same entry with opposite signs meet in one bank — with random digits almost every group has such a pair, so nearly every
# bpermute takes two passes per group.  The variant (tools/_build/libkyb_ab_requester_neg.so = the shipped sources + -DKYB_AB_REQUESTER_NEG,
# profiles/r06/ab_base_requester_neg.patch) keeps POSITIVE entries in all 64 lanes, pulls from (own half | index) — 32 sources, 32 banks, never
# a conflict — and lets the requesting lane swap ypx / ymx and negate xy2d (20 more v_cndmask per window).  Same box, interleaved; every run checks
# 64 outputs against the oracle.  Then the LDS / VALU counters of both builds.  -> profiles/r06/ab_base_requester_neg.log
set -u
cd "$GRAFT_REPO_ROOT"
V=$PWD/tools/_build/libkyb_ab_requester_neg.so
for i in 1 2 3; do
  for lib in shipped requester_neg; do
    for w in mul_base sign; do
      if [ $lib = requester_neg ]; then export KYB_HIP_LIB=$V; else unset KYB_HIP_LIB; fi
      python bench.py --workload $w --steps 20 --warmup 5 --only --no-cpu-baseline --check 64 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']
print('$lib', '$w', 'ms_per_step', l['ms_per_step'], 'value', l['value'], 'kernel_ms', r.get('avg_launch_ms'), 'executed_frac', r.get('executed_frac'), 'issue_share', r.get('issue_share'), 'parity_checked', l.get('parity_checked_items'))"
    done
  done
done
unset KYB_HIP_LIB
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in shipped requester_neg; do
  if [ $lib = requester_neg ]; then export KYB_HIP_LIB=$V; else unset KYB_HIP_LIB; fi
  out=gpurun_out/prof_ab_$lib; rm -rf $out; mkdir -p $out
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
             "GRBM_GUI_ACTIVE SQ_CYCLES SQ_INSTS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU2 SQ_INSTS_VSKIPPED"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 bench.py --workload mul_base --steps 2 --warmup 1 --only --no-cpu-baseline --check 64 > /dev/null 2> $out/p$i.err
  done
  python3 - "$out" "$lib" <<'PY'
import collections, csv, glob, sys
out, lib = sys.argv[1], sys.argv[2]
c = {}
for f in sorted(glob.glob(f"{out}/p*/*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_mul_base64" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            c["_vgpr"] = r.get("VGPR_Count")
    for k, v in agg.items():
        c[k] = sum(v) / len(v)
g = lambda k: c.get(k, float("nan"))
simd = g("GRBM_GUI_ACTIVE") / 8 * 1024
print(f"counters {lib}: vgpr {c.get('_vgpr')}  kernel cycles {g('GRBM_GUI_ACTIVE') / 8:.4g}  VALU busy {g('SQ_ACTIVE_INST_VALU') * 4 / simd:.4f}  LDS instruction active {g('SQ_ACTIVE_INST_LDS') * 4 / simd:.4f}  "
      f"VALU insts/wave {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.0f}  LDS insts/wave {g('SQ_INSTS_LDS') / g('SQ_WAVES'):.0f}  LDS array active cycles {g('SQ_LDS_IDX_ACTIVE'):.4g}  "
      f"bank-conflict cycles {g('SQ_LDS_BANK_CONFLICT'):.4g}  conflict share of LDS-active {g('SQ_LDS_BANK_CONFLICT') / max(1.0, g('SQ_LDS_IDX_ACTIVE')):.4f}")
PY
done
