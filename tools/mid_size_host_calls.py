"""Host-pointer calls of DKG-round size (2,048 .. 32,768 items) under different `host.zero_copy_kib` settings: up to the threshold the
kernels work on the context's page-locked buffer (one memcpy by the caller thread each way), above it the arrays travel by hipMemcpyAsync
from pageable memory.  Median wall time per call in ms; device-resident time of the same call beside it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import kyber_rs_amd
import synth

eng = kyber_rs_amd.Engine(0)
NMAX = 32768
s = synth.scalars(NMAX, 81)
k = synth.scalars(NMAX, 82, b"k")
enc, ext = eng.mul_base(s, want_ext=True)
msgs = kyber_rs_amd.pack_messages(synth.messages(NMAX, 83))      # blob + offsets once: no Python loop over the messages inside a timed call
sigs = eng.schnorr_sign(s, k, msgs)


def med(fn, reps=15):
    fn(); fn()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e3


print("n, zero_copy_kib, mul_base, mul, mul_enc, sign, verify   (ms per host-pointer call)")
for n in (2048, 4096, 8192, 16384, 32768):
    ref = None
    for kib in (512, 2048, 8192, 32768):
        eng.set_option("host.zero_copy_kib", kib)
        row = [med(lambda: eng.mul_base(s[:n])), med(lambda: eng.mul(k[:n], pts_ext=ext[:n])), med(lambda: eng.mul(k[:n], pts_enc=enc[:n])),
               med(lambda: eng.schnorr_sign(s[:n], k[:n], msgs[:n])), med(lambda: eng.verify(enc[:n], msgs[:n], sigs[:n], 1))]
        got = (eng.mul(k[:n], pts_ext=ext[:n]).tobytes(), eng.verify(enc[:n], msgs[:n], sigs[:n], 1).tobytes(), eng.schnorr_sign(s[:n], k[:n], msgs[:n]).tobytes())
        if ref is None:
            ref = got
        assert got == ref
        print(f"{n}, {kib}, " + ", ".join(f"{v:.3f}" for v in row), flush=True)
eng.set_option("host.zero_copy_kib", 4096)
