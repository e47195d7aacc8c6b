"""Latency of small batches through the host-pointer C ABI (what unmodified, one-call-at-a-time protocol code sees)
next to the single-thread CPU port: where the break-even batch size lies."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import kyber_rs_amd
import oracle_lib
import synth

eng = kyber_rs_amd.Engine(0)
# --opt key=value ...: engine options (e.g. coop.max_items=0 coop.base_max_items=0 to time the batch kernels alone)
for kv in sys.argv[1:]:
    if kv.startswith("--opt"):
        continue
    key, val = kv.split("=")
    eng.set_option(key, int(val))
    print(f"# option {key} = {val}")
orc = oracle_lib.Oracle()
N = 1 << 15
s = synth.scalars(N, 3)
enc, ext = eng.mul_base(s, want_ext=True)
k = synth.scalars(N, 4)
msgs = synth.messages(N, 3)
sig = eng.schnorr_sign(s, k, msgs)
pub = enc


def med(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2]


print("n, op, gpu_ms, gpu_items_per_s, cpu1_ms")
for n in (1, 8, 64, 512, 4096, 32768):
    reps = 30 if n <= 512 else 8
    ops = {
        "mul_base": (lambda: eng.mul_base(s[:n]), lambda: orc.mul_base_batch(s[:n])),
        "mul": (lambda: eng.mul(s[:n], pts_ext=ext[:n]), lambda: orc.mul_batch(s[:n], ext[:n])),
        "sign": (lambda: eng.schnorr_sign(s[:n], k[:n], msgs[:n]), lambda: orc.schnorr_sign_batch(s[:n], k[:n], msgs[:n])),
        "verify": (lambda: eng.verify(pub[:n], msgs[:n], sig[:n], flavor=1), lambda: orc.verify_batch(1, pub[:n], msgs[:n], sig[:n])),
    }
    for name, (g, c) in ops.items():
        tg = med(g, reps)
        tc = med(c, 3) if n <= 4096 else float("nan")
        print(f"{n}, {name}, {tg * 1e3:.3f}, {n / tg:.3e}, {tc * 1e3:.3f}", flush=True)
