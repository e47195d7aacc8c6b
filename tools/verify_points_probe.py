"""kyb_verify_points_batch (public keys as points: schnorr::verify / eddsa::verify) against kyb_verify_batch (32-byte keys: verify_with_checks),
device-resident, with the kernels of one call of each."""
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
dev = "cuda:0"
N = 1 << 20
s = torch.from_numpy(synth.scalars(N, 1)).to(dev)
k = torch.from_numpy(synth.scalars(N, 2, b"k")).to(dev)
pubs = torch.empty((N, 32), dtype=torch.uint8, device=dev)
ext = torch.empty((N, 40), dtype=torch.int32, device=dev)
msgs = torch.from_numpy(np.random.default_rng(3).integers(0, 256, 32 * N, dtype=np.uint8)).to(dev)
off = torch.arange(0, 32 * (N + 1), 32, dtype=torch.int32, device=dev)
sig = torch.empty((N, 64), dtype=torch.uint8, device=dev)
st1 = torch.empty((N,), dtype=torch.uint8, device=dev)
st2 = torch.empty((N,), dtype=torch.uint8, device=dev)
eng.mul_base_dev(s, out_enc=pubs, out_ext=ext)
eng.sign_dev(s, k, msgs, off, sig)
sig[::5, 40] ^= 3
eng.sync()


def t(fn, reps):
    for _ in range(3):
        fn()
    eng.sync()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); eng.sync(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e3


print("n, verify_bytes_ms, verify_points_ms, ratio, kernels of one points call")
for n in (4096, 8192, 32768, 1 << 18, 1 << 20):
    fb = lambda: eng.verify_dev(pubs[:n], msgs, off[: n + 1], sig[:n], st1[:n], 1)
    fp = lambda: eng.verify_points_dev(ext[:n], msgs, off[: n + 1], sig[:n], st2[:n], 1)
    a, b = t(fb, 15), t(fp, 15)
    assert torch.equal(st1[:n], st2[:n]) and int((st1[:n] != 0).sum()) == (n + 4) // 5
    eng.profile_begin(16); fp(); eng.sync(); recs = eng.profile_read(16); eng.profile_begin(0)
    print(f"{n}, {a:.3f}, {b:.3f}, {b / a:.3f}, " + " ".join(f"{nm}={ms:.3f}" for nm, ms in recs), flush=True)
