#!/bin/bash
# tools/ab_libs.sh "<lib...>" "<workload...>" rounds "<extra bench args>"
libs="$1"; wls="$2"; rounds="${3:-2}"; extra="${4:-}"
for w in $wls; do for r in $(seq 1 $rounds); do for l in $libs; do
  KYB_HIP_LIB=$PWD/kyber-rs_amd/$l python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --check 256 $extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$w', '$l', 'round $r', 'value %.4g'%d['value'], 'step_ms %.4f'%d['roofline']['step']['avg_step_ms'], d['roofline']['kernel'], 'launch_ms %.4f'%d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])
"
done; done; done
