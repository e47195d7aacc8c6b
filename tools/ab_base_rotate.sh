#!/bin/bash
# A/B on ONE box, interleaved: the fixed-base kernel with the rotating issue priority (the shipped library) against a build without it
# (make OUT=tools/_build/libkyb_norot.so OBJDIR=_obj_norot EXTRA=-DKYB_BASE64_ROTATE_PRIO=0) — profiles/r05/ab_base_rotate_prio.log
for i in 1 2 3; do
  for lib in rot norot; do
    for w in mul_base sign; do
      if [ $lib = norot ]; then export KYB_HIP_LIB=$PWD/tools/_build/libkyb_norot.so; else unset KYB_HIP_LIB; fi
      python bench.py --workload $w --steps 20 --warmup 5 --only --no-cpu-baseline --check 64 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']
print('$lib', '$w', 'ms_per_step', l['ms_per_step'], 'value', l['value'], 'kernel_ms', r.get('kernel_ms_avg', r.get('kernel_ms')), 'frac', r.get('frac'), 'executed_frac', r.get('executed_frac'))"
    done
  done
done
