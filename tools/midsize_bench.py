"""Device-resident call time of mid-size batches (the chip is not full): kernel-selection policy check."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

# kernel variants are selected here: the cross-check build (csrc/Makefile CROSSCHECK=1) — the product library has no such options
os.environ.setdefault("KYB_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber-rs_amd", "libkyber_ed25519_hip_crosscheck.so"))
import kyber_rs_amd
import synth

eng = kyber_rs_amd.Engine(0)
N = 1 << 19
s_np = np.random.default_rng(1).integers(0, 256, (N, 32), dtype=np.uint8); s_np[:, 31] &= 0x0F
s = torch.from_numpy(s_np).to("cuda:0")
out = torch.empty((N, 32), dtype=torch.uint8, device="cuda:0")
ext = torch.empty((N, 40), dtype=torch.int32, device="cuda:0")
eng.mul_base_dev(s, out_ext=ext); eng.sync()


def t(fn, reps=20):
    fn(); eng.sync()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); eng.sync(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e3


print("n, small_chunks, mul_base_ms, mul_ms")
for n in (4096, 65536, 131072, 262144):
    for sc in (0, 2):
        eng.set_option("mul_base.small_chunks", sc)
        a = t(lambda: eng.mul_base_dev(s[:n], out_enc=out[:n]))
        b = t(lambda: eng.mul_dev(s[:n], pts_ext=ext[:n], out_enc=out[:n]), 8) if sc == 0 else float("nan")
        print(f"{n}, {sc}, {a:.3f}, {b:.3f}", flush=True)

print("n, finish.min_items, mul_base_ms, sign_ms, mul_ms")
k = torch.from_numpy(synth.scalars(4096, 9)).to("cuda:0")
msgs = torch.zeros(4096 * 32, dtype=torch.uint8, device="cuda:0")
off = torch.arange(0, 32 * 4097, 32, dtype=torch.int32, device="cuda:0")
sig = torch.empty((4096, 64), dtype=torch.uint8, device="cuda:0")
eng.set_option("mul_base.small_chunks", 2)
for n in (1, 8, 32, 64, 512, 4095):
    for fm in (4096, 64, 8, 1):
        eng.set_option("finish.min_items", fm)
        a = t(lambda: eng.mul_base_dev(s[:n], out_enc=out[:n]))
        b = t(lambda: eng.sign_dev(s[:n], k[:n], msgs, off[: n + 1], sig[:n]))
        c = t(lambda: eng.mul_dev(s[:n], pts_ext=ext[:n], out_enc=out[:n]), 8)
        print(f"{n}, {fm}, {a:.3f}, {b:.3f}, {c:.3f}", flush=True)
eng.set_option("finish.min_items", 1)
