import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
# kernel variants are selected here: the cross-check build (csrc/Makefile CROSSCHECK=1) — the product library has no such options
os.environ.setdefault("KYB_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber-rs_amd", "libkyber_ed25519_hip_crosscheck.so"))
import kyber_rs_amd, synth
eng = kyber_rs_amd.Engine(0)
s = synth.scalars(683, 3)
_, ext = eng.mul_base(s, want_ext=True)
idx = np.full(1, 512, dtype=np.uint32)
for t in (683, 171, 43):
    for segs in (1, 2, 4, 8, 16, 32):
        if segs > t: continue
        eng.set_option("poly.segments", segs)
        eng.pubpoly_eval(ext[:t], idx)
        best = None
        for _ in range(5):
            eng.profile_begin(4)
            a = time.perf_counter(); eng.pubpoly_eval(ext[:t], idx); dt = time.perf_counter() - a
            prof = eng.profile_read(4)
            if best is None or dt < best[0]: best = (dt, prof)
        print(t, segs, f"{best[0]*1e6:.0f} us", best[1], flush=True)
