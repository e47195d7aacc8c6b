#!/usr/bin/env python3
"""One host-pointer kyb_mul_batch of 2^20 items from page-locked and from pageable memory, for a timeline:

  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/hp -- python3 tools/host_pipeline_trace.py [pinned|pageable] [reps]

and, without the profiler, the wall time per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import kyber_rs_amd

mode = sys.argv[1] if len(sys.argv) > 1 else "pinned"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
opts = [a for a in sys.argv[3:] if "=" in a]
eng = kyber_rs_amd.Engine(0)
for o in opts:
    k, v = o.split("=")
    eng.set_option(k, int(v))
n = 1 << 20
rng = np.random.default_rng(1)
s = rng.integers(0, 256, (n, 32), dtype=np.uint8); s[:, 31] &= 0x0f
ext = eng.mul_base(s[::-1].copy(), ext_only=True)
if mode == "pinned":
    ps = eng.pinned_array((n, 32), np.uint8); ps[:] = s
    pe = eng.pinned_array((n, 40), np.int32); pe[:] = ext
    po = eng.pinned_array((n, 32), np.uint8)
else:
    ps, pe, po = s, ext, np.empty((n, 32), dtype=np.uint8)
eng.mul_into(ps, pe, po)
ts = []
for _ in range(reps):
    t = time.perf_counter(); eng.mul_into(ps, pe, po); ts.append(time.perf_counter() - t)
print(f"{mode}: " + " ".join(f"{x*1e3:.2f}" for x in ts) + f" ms per call -> {n/min(ts):.3e} items/s best, options {opts}")
