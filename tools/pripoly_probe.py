"""PriPoly::shares (poly.rs:144-152) through kyb_pripoly_eval_batch against the single-thread CPU port: a dealer's n private shares of a
threshold-t polynomial, host-pointer calls (the coefficients and the shares cross PCIe inside the timed call)."""
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
orc = oracle_lib.Oracle()
print("n, t, gpu_ms_per_call, cpu_port_1_thread_ms (n Horner chains of t sc_mul_add), ratio")
for n, t in ((16, 11), (64, 43), (256, 171), (1024, 683), (4096, 2731), (1024, 16), (65536, 16)):
    coeffs = synth.scalars(t, t)
    idx = np.arange(n, dtype=np.uint32)
    got = eng.pripoly_eval(coeffs, idx)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); eng.pripoly_eval(coeffs, idx); ts.append(time.perf_counter() - t0)
    gpu = sorted(ts)[len(ts) // 2] * 1e3
    m = min(n, 64)
    t0 = time.perf_counter()
    want = [orc.pripoly_eval(coeffs, int(i)) for i in idx[:m]]
    cpu = (time.perf_counter() - t0) / m * n * 1e3
    assert all(bytes(got[i]) == want[i] for i in range(m))
    print(f"{n}, {t}, {gpu:.3f}, {cpu:.1f}, {cpu / gpu:.0f}x", flush=True)
