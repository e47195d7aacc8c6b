"""Two lanes per item against one (k_mul_ladder_pair / k_mul_ladder): device-resident call time of mul and verify per batch size with
`ladder.pair_max_items` forced on and off, next to the one-item-per-wavefront kernels where they apply.  Sets the option's default."""
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
N = 1 << 19
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
coop_mul, coop_base = eng.get_option("coop.max_items"), eng.get_option("coop.base_max_items")
eng.set_option("coop.ladder_max_items", 1 << 20)


def t(fn, reps=15):
    fn(); eng.sync()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); eng.sync(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e3


print("n, mul_one_lane_ms, mul_two_lanes_ms, mul_coop_ms, verify_one_lane_ms, verify_two_lanes_ms", flush=True)
for n in (2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768, 49152, 65536, 98304, 131072, 196608, 262144, 524288):
    row = []
    for what in ("mul", "verify"):
        eng.set_option("coop.max_items", 0); eng.set_option("coop.base_max_items", 0)
        fn = (lambda: eng.mul_dev(s[:n], pts_ext=ext[:n], out_enc=out[:n])) if what == "mul" else (lambda: eng.verify_dev(pubs[:n], msgs, off[: n + 1], sig[:n], status[:n], 1))
        eng.set_option("ladder.pair_max_items", 0)
        a = t(fn)
        eng.set_option("ladder.pair_max_items", 1 << 24)
        b = t(fn)
        row += [a, b]
        if what == "mul":
            eng.set_option("coop.max_items", 1 << 20)
            row.append(t(fn) if n <= 16384 else float("nan"))
    print(f"{n}, " + ", ".join(f"{v:.3f}" for v in row), flush=True)
eng.set_option("coop.max_items", coop_mul); eng.set_option("coop.base_max_items", coop_base)
