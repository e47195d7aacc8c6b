"""ladder.y_only = 0 (decode first) / 1 (decode on a side stream beside the ladder) / 2 (ladder and decode workgroups in ONE launch): per-kernel times
and call times of device-resident mul_enc and verify calls, the modes interleaved on one box; every mode's bytes are compared with mode 0's."""
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
N = 1 << 15
dev = "cuda:0"
s = torch.from_numpy(synth.scalars(N, 1)).to(dev)
k = torch.from_numpy(synth.scalars(N, 2)).to(dev)
out = torch.empty((N, 32), dtype=torch.uint8, device=dev)
pubs = torch.empty((N, 32), dtype=torch.uint8, device=dev)
msgs = torch.from_numpy(np.random.default_rng(3).integers(0, 256, 32 * N, dtype=np.uint8)).to(dev)
off = torch.arange(0, 32 * (N + 1), 32, dtype=torch.int32, device=dev)
sig = torch.empty((N, 64), dtype=torch.uint8, device=dev)
status = torch.empty((N,), dtype=torch.uint8, device=dev)
eng.mul_base_dev(s, out_enc=pubs)
eng.sign_dev(s, k, msgs, off, sig)
eng.sync()
# every eleventh signature damaged in one of four ways, so that the statuses are not all zero
sg = sig.cpu().numpy().copy()
pb = pubs.cpu().numpy().copy()
for i in range(0, N, 11):
    kind = (i // 11) % 4
    if kind == 0: sg[i, 3] ^= 1            # R changed (decodes or not)
    elif kind == 1: sg[i, 40] ^= 1         # s changed
    elif kind == 2: pb[i, 5] ^= 1          # key changed
    else: sg[i, 63] |= 0xf0                # s >= L
sig.copy_(torch.from_numpy(sg).to(dev)); pubs2 = torch.from_numpy(pb).to(dev)
ref = {}
for rnd in range(2):
    for mode in (0, 1, 2):
        eng.set_option("ladder.y_only", mode)
        for n in (4096, 8192, 16384, 32768):
            for what, fn, res in (("mul_enc", lambda: eng.mul_dev(s[:n], pts_enc=pubs[:n], out_enc=out[:n]), out),
                                  ("verify", lambda: eng.verify_dev(pubs2[:n], msgs, off[: n + 1], sig[:n], status[:n], 1), status)):
                for _ in range(3):
                    fn()
                eng.sync()
                ts = []
                for _ in range(11):
                    t0 = time.perf_counter(); fn(); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
                eng.profile_begin(16)
                fn(); eng.sync()
                recs = eng.profile_read(16)
                eng.profile_begin(0)
                got = res[:n].cpu().numpy().tobytes()
                assert ref.setdefault((what, n), got) == got, (what, n, mode)
                print(f"y_only {mode} n={n} {what}: call {sorted(ts)[5]:.3f} ms; " + " ".join(f"{name}={ms:.3f}" for name, ms in recs), flush=True)
st = np.frombuffer(ref[("verify", 32768)], dtype=np.uint8)
print("statuses of the 32768-item verification:", {int(v): int(c) for v, c in zip(*np.unique(st, return_counts=True))})
eng.set_option("ladder.y_only", 2)
