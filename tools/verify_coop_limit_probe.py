"""k_verify_coop (one launch, three wavefronts per signature) against the kernel sequence of the one-item-per-wavefront regime, per batch size:
where coop.verify_max_items belongs."""
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
N = 4096
dev = "cuda:0"
s = torch.from_numpy(synth.scalars(N, 1)).to(dev)
k = torch.from_numpy(synth.scalars(N, 2)).to(dev)
pubs = torch.empty((N, 32), dtype=torch.uint8, device=dev)
msgs = torch.from_numpy(np.random.default_rng(3).integers(0, 256, 32 * N, dtype=np.uint8)).to(dev)
off = torch.arange(0, 32 * (N + 1), 32, dtype=torch.int32, device=dev)
sig = torch.empty((N, 64), dtype=torch.uint8, device=dev)
status = torch.empty((N,), dtype=torch.uint8, device=dev)
eng.mul_base_dev(s, out_enc=pubs)
eng.sign_dev(s, k, msgs, off, sig)
eng.sync()


def t(fn, reps=21):
    fn(); fn(); eng.sync()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); eng.sync(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e3


print("n: one launch / kernel sequence   (ms per device-resident verify call)")
for n in (256, 512, 640, 768, 896, 1024, 1280, 1536):
    row = []
    for vm in (1 << 20, 0):
        eng.set_option("coop.verify_max_items", vm)
        row.append(t(lambda: eng.verify_dev(pubs[:n], msgs, off[: n + 1], sig[:n], status[:n], 1)))
    print(f"{n}: {row[0]:.3f} / {row[1]:.3f}", flush=True)
