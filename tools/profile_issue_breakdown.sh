#!/bin/bash
# Run ON THE GPU BOX (via gpurun): where the issue cycles of the fixed-base kernel go (round-4 review item 5: k_mul_base64 VALUBusy 92.8 % against
# the ladder's 98.9 %, "no counter in profiles/ says why").  Four --pmc passes per workload (counters only, no trace domain), the ladder beside the
# fixed base for comparison; raw output under gpurun_out/prof_issue/, table by tools/summarise_issue_breakdown.py -> profiles/<round>/issue_breakdown.*
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_issue; rm -rf $out; mkdir -p $out
EXTRA="${KYB_BENCH_EXTRA:-}"
for w in ${WORKLOADS:-mul_base mul sign}; do
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
             "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_IFETCH" \
             "GRBM_GUI_ACTIVE SQ_CYCLES SQ_INSTS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU2 SQ_INSTS_VSKIPPED"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $out/${w}_p$i -- python3 bench.py --workload $w --steps 2 --warmup 1 --only --no-cpu-baseline --check 64 $EXTRA > /dev/null 2> $out/${w}_p$i.err
  done
  echo "issue counters collected for $w"
done
