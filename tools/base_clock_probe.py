"""In-kernel clock and duration of the fixed-base kernel (k_mul_base64) under sustained load, against a cold start: does the 2.1 GHz the
bench reads for it (the ladder: 2.36 GHz) come from the clock ramp of a short run or from the kernel's own power draw?"""
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

eng = kyber_rs_amd.Engine(0)
dev = torch.device("cuda", 0)
n = 1 << 20
rng = np.random.default_rng(7)
sc_np = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
sc_np[:, 31] &= 0x0f
sc = torch.from_numpy(sc_np).to(dev)
out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
tstream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(tstream)
st = tstream.cuda_stream
stamps = torch.zeros(8, dtype=torch.int64, device=dev)
for block in (1024, 768):
    eng.set_option("mul_base.block64", block)
    for seconds in (0.0, 0.05, 0.5, 2.5):
        torch.cuda.synchronize()
        time.sleep(1.0)                                  # let the clock fall back
        t0 = time.time()
        launches = 0
        while True:
            eng.mul_base_dev(sc, out_enc=out, stream=st)
            launches += 1
            if time.time() - t0 >= seconds:
                break
        torch.cuda.synchronize()
        stamps.zero_()
        torch.cuda.synchronize()
        eng.wave_stamps(stamps)
        eng.profile_begin(8)
        eng.mul_base_dev(sc, out_enc=out, stream=st)
        torch.cuda.synchronize()
        prof = dict(eng.profile_read(8))
        eng.profile_begin(0)
        eng.wave_stamps(None)
        a = stamps.cpu().numpy()
        ghz = (a[2] - a[0]) / (a[3] - a[1]) * 0.1 if a[3] != a[1] else float("nan")
        print(f"block {block}: after {seconds:4.2f} s of back-to-back launches ({launches:4d}): k_mul_base {prof.get('k_mul_base', float('nan')):.4f} ms, in-kernel clock {ghz:.3f} GHz, waves stamped {a[4]}", flush=True)
eng.set_option("mul_base.block64", 1024)
