"""Independent device-pointer calls on different streams overlap: N small variable-base batches issued on one
stream vs on N streams (each stream has its own scratch inside the engine)."""
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
dev = torch.device("cuda:0")
n = 64
K = 4
s = [torch.from_numpy(synth.scalars(n, 10 + i)).to(dev) for i in range(K)]
ext = [torch.empty((n, 40), dtype=torch.int32, device=dev) for _ in range(K)]
out = [torch.empty((n, 32), dtype=torch.uint8, device=dev) for _ in range(K)]
for i in range(K):
    eng.mul_base_dev(s[i], out_ext=ext[i])
eng.sync()
streams = [torch.cuda.Stream(device=dev) for _ in range(K)]


def run(multi):
    t0 = time.perf_counter()
    for i in range(K):
        st = streams[i] if multi else streams[0]
        eng.mul_dev(s[i], pts_ext=ext[i], out_enc=out[i], stream=st.cuda_stream)
    for st in streams:
        st.synchronize()
    return (time.perf_counter() - t0) * 1e3


for multi in (False, True, False, True):
    run(multi)
    ts = sorted(run(multi) for _ in range(20))
    print(f"{K} variable-base batches of {n} items on {'%d streams' % K if multi else '1 stream '}: {ts[len(ts) // 2]:.3f} ms")
ref = [o.clone() for o in out]
run(False)
assert all(torch.equal(a, b) for a, b in zip(ref, out))
