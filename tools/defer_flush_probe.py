#!/usr/bin/env python3
"""Where the time of a deferred flush goes (profiles/r05/defer_flush_probe.log): the engine calls a flush of the DKG-finish shape makes, timed one by
one through the host-pointer ABI — kyb_sum_batch of 43 groups of 64 points (dist_key_share, dkg.rs:905-953), with and without the encodings —
next to the same call on page-locked arrays and on device-resident arrays."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kyber_rs_amd, synth

def med(fn, reps=30):
    fn(); fn()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return round(sorted(ts)[len(ts) // 2] * 1e3, 4)

eng = kyber_rs_amd.Engine(0)
eng.set_option("ext.projective", 1)
for m, t in ((43, 64), (43, 8), (171, 256), (1, 64)):
    pts = eng.mul_base(synth.scalars(m * t, 5), ext_only=True).reshape(m, t, 40)
    P = kyber_rs_amd._ptr
    enc = np.zeros((m, 32), np.uint8); ext = np.zeros((m, 40), np.int32)
    lib = eng.lib
    row = {"m": m, "t": t}
    row["sum enc+ext ms"] = med(lambda: lib.kyb_sum_batch(P(pts), m, t, P(enc), P(ext)))
    row["sum ext only ms"] = med(lambda: lib.kyb_sum_batch(P(pts), m, t, None, P(ext)))
    row["sum enc only ms"] = med(lambda: lib.kyb_sum_batch(P(pts), m, t, P(enc), None))
    pp = eng.pinned_array(pts.shape, np.int32); pp[:] = pts
    pe = eng.pinned_array(enc.shape, np.uint8); px = eng.pinned_array(ext.shape, np.int32)
    row["pinned enc+ext ms"] = med(lambda: lib.kyb_sum_batch(P(pp), m, t, P(pe), P(px)))
    row["add_batch(m*t pairs) ms"] = med(lambda: eng.add(pts.reshape(-1, 40), pts.reshape(-1, 40)))
    row["encode(m) ms"] = med(lambda: eng.encode(ext))
    print(row, flush=True)
