#!/usr/bin/env python3
"""Fixed base and signing at small and mid sizes (host-pointer calls, as a binding makes them): the one-item-per-wavefront kernels (k_mul_base_coop,
k_sign_coop) against the mid-size form — four wavefronts per 64 items, a quarter of the 43 windows each (k_mul_base64_quarters) and one inversion
per wavefront behind it (k_finish_wave).  Where the hand-over coop.base_max_items belongs.  Both sides give the same bytes.

  python tools/base_quarters_probe.py        (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import kyber_rs_amd
import synth

eng = kyber_rs_amd.Engine(0, crosscheck=True)      # mul_base.quarters is a selector of the cross-check build (same kernels)


def med(fn, n=60):
    fn(); fn(); ts = []
    for _ in range(n):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e6


keep = {k: eng.get_option(k) for k in ("coop.base_max_items", "coop.verify_max_items")}
print("items, mul_base one item per wavefront [us], mul_base in quarters [us], sign one per wavefront [us], sign in quarters [us]")
for n in (64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096, 4608, 8192):
    s, k = synth.raw256(n, 7), synth.scalars(n, 8, b"k")
    msgs = kyber_rs_amd.pack_messages(synth.messages(n, 9))      # 32-byte digests, as DSS signs them, packed once
    x = s.copy(); x[:, 31] &= 0x7f
    eng.set_option("coop.base_max_items", 1 << 20)
    eng.set_option("mul_base.quarters", 0)             # (with the mid-size form on, the fixed base leaves the one-item kernels at 5 wavefronts per CU whatever coop.base_max_items says)
    a, ea = med(lambda: eng.mul_base(s)), eng.mul_base(s)
    c, ec = med(lambda: eng.schnorr_sign(x, k, msgs)), eng.schnorr_sign(x, k, msgs)
    eng.set_option("mul_base.quarters", 1)
    eng.set_option("coop.base_max_items", 0)
    eng.set_option("coop.verify_max_items", 0)
    b, eb = med(lambda: eng.mul_base(s)), eng.mul_base(s)
    d, ed = med(lambda: eng.schnorr_sign(x, k, msgs)), eng.schnorr_sign(x, k, msgs)
    eng.set_option("coop.verify_max_items", keep["coop.verify_max_items"])
    assert np.array_equal(ea, eb) and np.array_equal(ec, ed)
    print("%5d, %.1f, %.1f, %.1f, %.1f" % (n, a, b, c, d), flush=True)
for k_, v in keep.items():
    eng.set_option(k_, v)
