"""Kernel-only durations (the engine's HIP events) of small batches through the cooperative and the batch kernels,
next to the wall time of the host-pointer call that contains them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kyber_rs_amd, synth
eng = kyber_rs_amd.Engine(0)
s = synth.scalars(8192, 3)
_, ext = eng.mul_base(s, want_ext=True)
k = synth.scalars(8192, 4)
msgs = [bytes([i & 255]) * 32 for i in range(1024)]
pubs = eng.mul_base(s[:1024])
sigs = eng.schnorr_sign(s[:1024], k[:1024], msgs)
print("n, path, op, call_us, kernels")
for coop in (1, 0):
    eng.set_option("coop.max_items", 1 << 20 if coop else 0)
    eng.set_option("coop.ladder_max_items", 1 << 20)
    eng.set_option("coop.ladder_enc_max_items", 1 << 20)
    eng.set_option("coop.base_max_items", 1 << 20 if coop else 0)
    for n in (1, 64, 1024, 2048, 4096):
        for op, fn in (("mul_base", lambda: eng.mul_base(s[:n])), ("mul", lambda: eng.mul(s[:n], pts_ext=ext[:n])),
                       ("sign", lambda: eng.schnorr_sign(s[:n], k[:n], msgs[:n])), ("verify", lambda: eng.verify(pubs[:n], msgs[:n], sigs[:n], 1))):
            if op in ("sign", "verify") and n > 1024:
                continue
            fn(); fn()
            ts = []
            for _ in range(7):
                eng.profile_begin(16)
                t = time.perf_counter(); fn(); dt = time.perf_counter() - t
                prof = eng.profile_read(16)
                ts.append((dt, prof))
            eng.profile_begin(0)
            dt, prof = sorted(ts, key=lambda x: x[0])[len(ts) // 2]
            print(f"{n}, {'coop' if coop else 'batch'}, {op}, {dt * 1e6:.1f}, " + " ".join(f"{k}={v * 1e3:.1f}us" for k, v in prof), flush=True)
