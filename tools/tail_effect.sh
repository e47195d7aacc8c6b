for w in 3 2; do for n in 1048576 983040 786432; do python bench.py --workload mul --n $n --opt mul.ladder_waves=$w --no-cpu-baseline --steps 10 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('waves $w n $n value %.4g'%d['value'], 'ladder_ms %.4f'%d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])
"; done; done
