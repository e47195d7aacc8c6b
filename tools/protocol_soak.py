#!/usr/bin/env python3
"""Repeat the call-by-call protocol programs (tests/cpp/test_{vss_round,dkg_finish,dss_round}.cpp) at many shapes for a while: every run's eager,
recorded and batch-aware transcripts must be identical, and identical to the CPU port's.  Thousands of one-item engine calls per run — the
one-item kernels (an item's scalar in four pieces on four workgroups, arrival counters in device scratch) under the call pattern that uses them.

  python tools/protocol_soak.py [seconds = 120]      (GPU)"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_vss_round as R

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(5)
bins = {p: (R.build(p), R.build(p, cpu_port=True)) for p in ("test_vss_round", "test_dkg_finish", "test_dss_round")}
t0, runs, lines = time.time(), 0, 0
while time.time() - t0 < budget:
    prog = rng.choice(list(bins))
    n = rng.randint(2, 40)
    t = rng.randint(1, n)
    gpu, cpu = bins[prog]
    out, _ = R.run_program(gpu, n, t)
    ref, _ = R.run_program(cpu, n, t, "eager")
    assert out["E"] == out["D"] == ref["E"], (prog, n, t)
    assert not out["B"] or out["B"] == out["E"], (prog, n, t)
    runs += 1; lines += len(out["E"])
print(f"protocol_soak: {runs} runs ({lines} transcript lines) in {time.time() - t0:.0f} s, every form equal to the CPU port's")
