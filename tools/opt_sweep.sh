#!/bin/bash
# sweep engine options for one workload: tools/opt_sweep.sh <workload> "<opt list A>" "<opt list B>" ...
w=$1; shift
for r in 1 2; do
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --check 256 $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$w [$o] round $r value %.4g'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'frac', d['roofline']['frac'])
"
done; done
