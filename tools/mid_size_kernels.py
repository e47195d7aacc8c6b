"""Per-kernel times of ONE mid-size device-resident call (HIP events around each launch, kyb_profile_*): where the 0.6 ms of a
8,192..32,768-item variable-base multiplication or verification go."""
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
N = 1 << 16
dev = "cuda:0"
s = torch.from_numpy(synth.scalars(N, 1)).to(dev)
k = torch.from_numpy(synth.scalars(N, 2)).to(dev)
out = torch.empty((N, 32), dtype=torch.uint8, device=dev)
ext = torch.empty((N, 40), dtype=torch.int32, device=dev)
pubs = torch.empty((N, 32), dtype=torch.uint8, device=dev)
msgs = torch.from_numpy(np.random.default_rng(3).integers(0, 256, 32 * N, dtype=np.uint8)).to(dev)
off = torch.arange(0, 32 * (N + 1), 32, dtype=torch.int32, device=dev)
sig = torch.empty((N, 64), dtype=torch.uint8, device=dev)
status = torch.empty((N,), dtype=torch.uint8, device=dev)
eng.mul_base_dev(s, out_enc=pubs, out_ext=ext)
eng.sign_dev(s, k, msgs, off, sig)
eng.sync()
for n in (4096, 8192, 16384, 32768, 65536):
    for what, fn in (("mul_base", lambda: eng.mul_base_dev(s[:n], out_enc=out[:n])), ("sign", lambda: eng.sign_dev(s[:n], k[:n], msgs, off[: n + 1], sig[:n])),
                     ("mul", lambda: eng.mul_dev(s[:n], pts_ext=ext[:n], out_enc=out[:n])), ("mul_enc", lambda: eng.mul_dev(s[:n], pts_enc=pubs[:n], out_enc=out[:n])),
                     ("verify", lambda: eng.verify_dev(pubs[:n], msgs, off[: n + 1], sig[:n], status[:n], 1))):
        for _ in range(3):
            fn()
        eng.sync()
        t0 = time.perf_counter(); fn(); eng.sync(); wall = (time.perf_counter() - t0) * 1e3
        eng.profile_begin(16)
        fn(); eng.sync()
        recs = eng.profile_read(16)
        eng.profile_begin(0)
        print(f"n={n} {what}: call {wall:.3f} ms; " + " ".join(f"{name}={ms:.3f}" for name, ms in recs), flush=True)
