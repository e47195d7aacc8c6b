#!/usr/bin/env python3
"""One-item host-pointer calls: wall time of the call against the time its kernels take on the device (the engine's own event profiler), i.e. how
much of a batch-of-1 call is launch / completion overhead and how much is the serial arithmetic of one item (profiles/r05/one_item_breakdown.log)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kyber_rs_amd, synth

eng = kyber_rs_amd.Engine(0)
s, k = synth.scalars(4, 5), synth.scalars(4, 6, b"k")
enc, ext = eng.mul_base(s, want_ext=True)
msgs = synth.messages(4, 7)
sig = eng.schnorr_sign(s, k, msgs)
ops = {"mul_base -> enc": lambda: eng.mul_base(s[:1]), "mul(ext) -> enc": lambda: eng.mul(k[:1], pts_ext=ext[:1]), "sign": lambda: eng.schnorr_sign(s[:1], k[:1], msgs[:1]),
       "encode": lambda: eng.encode(ext[:1]), "decode": lambda: eng.decode(enc[:1]), "add": lambda: eng.add(ext[:1], ext[1:2]), "verify": lambda: eng.verify(enc[:1], msgs[:1], sig[:1], 1)}
for proj in (0, 1):
    eng.set_option("ext.projective", proj)
    if proj:
        ops = {"mul_base -> ext (projective)": lambda: eng.mul_base(s[:1], ext_only=True), "mul(ext) -> ext (projective)": lambda: eng.mul(k[:1], pts_ext=ext[:1], ext_only=True)}
    for name, fn in ops.items():
        for _ in range(20): fn()
        ts = []
        for _ in range(200):
            a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
        wall = sorted(ts)[100] * 1e6
        eng.profile_begin(64)
        for _ in range(8): fn()
        recs = eng.profile_read()
        per = {}
        for nm, ms in recs:
            per.setdefault(nm, []).append(ms)
        kern = sum(sum(v) for v in per.values()) / 8 * 1e3
        print(f"{name:32s} wall {wall:7.1f} us   kernels {kern:7.1f} us   ({', '.join(f'{n} x{len(v)//8}: {sorted(v)[len(v)//2]*1e3:.1f}' for n, v in per.items())})", flush=True)
