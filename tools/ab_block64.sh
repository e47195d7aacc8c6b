for r in 1 2; do for b in 1024 512; do for w in mul_base sign; do python bench.py --workload $w --opt mul_base.block64=$b --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$w block64 $b value %.4g'%d['value'], 'launch_ms %.4f'%d['roofline']['avg_launch_ms'], 'exec_frac', d['roofline'].get('executed_frac'))
"; done; done; done
