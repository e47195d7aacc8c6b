"""Where the role-split launches of the DKG-sized paths (ladder.y_only = 2) take over from the one-item-per-wavefront kernels: device-resident
mul_enc and verify calls per batch size, routing as shipped against the small-batch kernels switched off, interleaved on one box."""
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
N = 8192
dev = "cuda:0"
s = torch.from_numpy(synth.scalars(N, 1)).to(dev)
k = torch.from_numpy(synth.scalars(N, 2)).to(dev)
out = torch.empty((N, 32), dtype=torch.uint8, device=dev)
pubs = torch.empty((N, 32), dtype=torch.uint8, device=dev)
ext = torch.empty((N, 40), dtype=torch.int32, device=dev)
msgs = torch.from_numpy(np.random.default_rng(3).integers(0, 256, 32 * N, dtype=np.uint8)).to(dev)
off = torch.arange(0, 32 * (N + 1), 32, dtype=torch.int32, device=dev)
sig = torch.empty((N, 64), dtype=torch.uint8, device=dev)
status = torch.empty((N,), dtype=torch.uint8, device=dev)
sig2 = torch.empty((N, 64), dtype=torch.uint8, device=dev)
eng.mul_base_dev(s, out_enc=pubs, out_ext=ext)
eng.sign_dev(s, k, msgs, off, sig)
eng.sync()
KEYS = ("coop.max_items", "coop.base_max_items", "coop.verify_max_items", "coop.ladder_max_items", "coop.ladder_enc_max_items")
saved = {kk: eng.get_option(kk) for kk in KEYS}


def t(fn, reps=21):
    fn(); fn(); eng.sync()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); eng.sync(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e3


print("n, routing: mul_ms, mul_enc_ms, verify_ms, mul_base_ms, sign_ms", flush=True)
for n in (64, 512, 1024, 1536, 2048, 2304, 2560, 2816, 3072, 3584, 4096, 5120, 6144):
    for rnd in range(2):
        for name, opts in (("shipped", saved), ("batch kernels", {kk: 0 for kk in KEYS})):
            for kk, v in opts.items():
                eng.set_option(kk, v)
            m = t(lambda: eng.mul_dev(s[:n], pts_ext=ext[:n], out_enc=out[:n]))
            a = t(lambda: eng.mul_dev(s[:n], pts_enc=pubs[:n], out_enc=out[:n]))
            b = t(lambda: eng.verify_dev(pubs[:n], msgs, off[: n + 1], sig[:n], status[:n], 1))
            c = t(lambda: eng.mul_base_dev(s[:n], out_enc=out[:n]))
            d = t(lambda: eng.sign_dev(s[:n], k[:n], msgs, off[: n + 1], sig2[:n]))
            print(f"{n}, {name}: {m:.3f}, {a:.3f}, {b:.3f}, {c:.3f}, {d:.3f}", flush=True)
for kk, v in saved.items():
    eng.set_option(kk, v)
