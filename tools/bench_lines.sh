#!/bin/bash
# Run ON THE GPU BOX: the bench lines of every workload for the current build (profiles/<round>/bench_lines.jsonl)
: > gpurun_out/bench_lines.jsonl
python3 bench.py --workload mul 2>/dev/null | tail -1 >> gpurun_out/bench_lines.jsonl
for w in mul_base sign verify; do python3 bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/bench_lines.jsonl; done
python3 bench.py --workload sign --keyed --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/bench_lines.jsonl
echo "bench lines done"
