"""Where the one-item-per-wavefront kernels stop paying: device-resident call time of mul / mul_base / sign / verify with the
cooperative kernels forced on and forced off, per batch size.  Sets `coop.max_items` / `coop.base_max_items` defaults."""
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
ext2 = torch.empty((N, 40), dtype=torch.int32, device=dev)
pubs = torch.empty((N, 32), dtype=torch.uint8, device=dev)
msgs = torch.from_numpy(np.random.default_rng(3).integers(0, 256, 32 * N, dtype=np.uint8)).to(dev)
off = torch.arange(0, 32 * (N + 1), 32, dtype=torch.int32, device=dev)
sig = torch.empty((N, 64), dtype=torch.uint8, device=dev)
status = torch.empty((N,), dtype=torch.uint8, device=dev)
eng.set_option("coop.max_items", 0)
eng.set_option("coop.base_max_items", 0)
eng.mul_base_dev(s, out_enc=pubs, out_ext=ext)
eng.sign_dev(s, k, msgs, off, sig)
eng.sync()


def t(fn, reps=15):
    fn(); eng.sync()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); eng.sync(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e3


print("n, coop, mul_base_ms, mul_ms, sign_ms, verify_ms, decode_ms", flush=True)
for n in (512, 1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384, 32768):
    for coop in (2, 1, 0):                     # 2: as 1, and verification by the single-launch kernel (three wavefronts per signature)
        eng.set_option("coop.max_items", 1 << 20 if coop else 0)
        eng.set_option("coop.ladder_max_items", 1 << 20)
        eng.set_option("coop.ladder_enc_max_items", 1 << 20)
        eng.set_option("coop.base_max_items", 1 << 20 if coop else 0)
        eng.set_option("coop.verify_max_items", 1 << 20 if coop == 2 else 0)
        a = t(lambda: eng.mul_base_dev(s[:n], out_enc=out[:n]))
        b = t(lambda: eng.mul_dev(s[:n], pts_ext=ext[:n], out_enc=out[:n]))
        c = t(lambda: eng.sign_dev(s[:n], k[:n], msgs, off[: n + 1], sig[:n]))
        d = t(lambda: eng.verify_dev(pubs[:n], msgs, off[: n + 1], sig[:n], status[:n], 1))
        e = t(lambda: eng.decode_dev(pubs[:n], ext2[:n]))
        print(f"{n}, {coop}, {a:.3f}, {b:.3f}, {c:.3f}, {d:.3f}, {e:.3f}", flush=True)
