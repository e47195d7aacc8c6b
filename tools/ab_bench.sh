#!/bin/bash
# A/B timing of library variants on one GPU box: interleaved rounds, same process settings.
# usage: tools/ab_bench.sh "<lib1> <lib2> ..." "<bench args>" rounds
libs="$1"; args="$2"; rounds="${3:-2}"
for r in $(seq 1 $rounds); do
  for l in $libs; do
    KYB_HIP_LIB=$PWD/kyber-rs_amd/$l python bench.py $args --no-cpu-baseline --check 256 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$l', 'round $r', d['config']['items_per_gpu'], 'value %.4g'%d['value'], 'kern_ms %.4f'%d['roofline']['avg_kernel_ms'], 'frac', d['roofline']['frac'])
"
  done
done
