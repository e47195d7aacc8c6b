"""Many host threads, each with its own context, issuing ONE-item calls at the same time: what a server that handles requests
concurrently gets out of one GPU (ctypes releases the GIL during a call)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kyber_rs_amd, synth

base = kyber_rs_amd.Engine(0)
s = synth.scalars(64, 3)
enc, ext = base.mul_base(s, want_ext=True)
msgs = [b"m" * 32] * 64
sigs = base.schnorr_sign(s, np.roll(s, 1, axis=0).copy(), msgs)
DUR = 1.5
print("threads, op, calls_per_s, mean_call_us")
for nt in (1, 4, 16, 64):
    engines = [kyber_rs_amd.Engine(0, private=True) for _ in range(nt)]
    for op in ("mul_base", "mul", "verify"):
        counts = [0] * nt
        stop = time.perf_counter() + DUR
        def work(i):
            e = engines[i]
            fn = {"mul_base": lambda: e.mul_base(s[:1]), "mul": lambda: e.mul(s[:1], pts_ext=ext[:1]), "verify": lambda: e.verify(enc[:1], msgs[:1], sigs[:1], 1)}[op]
            fn()
            while time.perf_counter() < stop:
                fn(); counts[i] += 1
        th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        t0 = time.perf_counter()
        for t_ in th: t_.start()
        for t_ in th: t_.join()
        dt = time.perf_counter() - t0
        tot = sum(counts)
        print(f"{nt}, {op}, {tot / dt:.0f}, {dt * nt / max(tot, 1) * 1e6:.0f}", flush=True)
    for e in engines: e.close()
