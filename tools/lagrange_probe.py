"""kyb_lagrange_coeffs_batch: call time for share sets of DKG sizes (host-pointer calls, medians), against Python integers on a sample."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import kyber_rs_amd
import synth

eng = kyber_rs_amd.Engine(0)
L = synth.L
rng = np.random.default_rng(9)
print("m sets x t indices (index range): ms per call")
for m, t, hi in ((1, 43, 64), (1, 171, 256), (1, 683, 1024), (64, 683, 1024), (1024, 683, 1024), (1, 683, 1 << 20), (1, 683, 0xffffffff)):
    idx = np.stack([np.sort(rng.choice(hi, t, replace=False)) for _ in range(m)]).astype(np.uint32)
    lam = eng.lagrange_coeffs(idx)
    xs = [int(v) + 1 for v in idx[m - 1]]
    for i in (0, t - 1):
        num = den = 1
        for j in range(t):
            if j != i:
                num = num * xs[j] % L
                den = den * (xs[j] - xs[i]) % L
        assert int.from_bytes(bytes(lam[m - 1, i]), "little") == num * pow(den, L - 2, L) % L
    ts = []
    for _ in range(11):
        a = time.perf_counter(); eng.lagrange_coeffs(idx); ts.append(time.perf_counter() - a)
    print(f"{m} x {t} (< {hi}): {sorted(ts)[5] * 1e3:.3f}", flush=True)
