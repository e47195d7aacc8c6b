#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for the five
# bench workloads; raw output under gpurun_out/prof/, summarised afterwards by tools/summarise_profiles.py
# (PMC passes are collected on their own, never combined with --sys-trace etc.).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof; rm -rf $out; mkdir -p $out
for w in mul mul_enc mul_base sign verify; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/${w}_trace -- python3 bench.py --workload $w --steps 20 --warmup 5 --only --no-cpu-baseline --check 64 > $out/${w}_trace.json 2> $out/${w}_trace.err
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/${w}_pmc_sq -- python3 bench.py --workload $w --steps 2 --warmup 1 --only --no-cpu-baseline --check 64 > /dev/null 2> $out/${w}_pmc_sq.err
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $out/${w}_pmc_fetch -- python3 bench.py --workload $w --steps 2 --warmup 1 --only --no-cpu-baseline --check 64 > /dev/null 2> $out/${w}_pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/${w}_pmc_write -- python3 bench.py --workload $w --steps 2 --warmup 1 --only --no-cpu-baseline --check 64 > /dev/null 2> $out/${w}_pmc_write.err
  echo "profiled $w"
done
# then: python tools/summarise_profiles.py <round> (here or in the build container), and tools/bench_lines.sh for the plain lines
