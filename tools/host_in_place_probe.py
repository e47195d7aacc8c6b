"""DKG-sized host-pointer calls on pageable arrays, on page-locked arrays copied into the context's buffer (host.in_place = 0) and on page-locked
arrays used where they lie (host.in_place = 1): medians of 25 calls, interleaved on one box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import kyber_rs_amd
import synth

eng = kyber_rs_amd.Engine(0)
lib, P, ck = eng.lib, kyber_rs_amd._ptr, kyber_rs_amd._check
NMAX = 16384
s = synth.scalars(NMAX, 81)
k = synth.scalars(NMAX, 82, b"k")
enc, ext = eng.mul_base(s, want_ext=True)
msgs = kyber_rs_amd.pack_messages(synth.messages(NMAX, 83))
sigs = eng.schnorr_sign(s, k, msgs)


def pin(a):
    b = eng.pinned_array(a.shape, a.dtype)
    b[...] = a
    return b


def med(fn, reps=25):
    fn(); fn()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e3


print("n, arrays, host.in_place: mul, mul_enc, verify, mul_base   (ms per host-pointer call, median of 25)")
for n in (1024, 4096, 8192, 16384):
    pg = dict(k=k[:n].copy(), enc=enc[:n].copy(), ext=ext[:n].copy(), sig=sigs[:n].copy(), blob=msgs.blob[: int(msgs.off[n])].copy(), off=msgs.off[: n + 1].copy(),
              out=np.empty((n, 32), np.uint8), st=np.empty((n,), np.uint8))
    pl = {kk: pin(v) for kk, v in pg.items()}
    ref = None
    for rnd in range(2):
        for name, a, ip in (("pageable", pg, 1), ("page-locked", pl, 0), ("page-locked", pl, 1)):
            eng.set_option("host.in_place", ip)
            row = [med(lambda: ck(lib.kyb_mul_batch(P(a["k"]), None, P(a["ext"]), n, P(a["out"]), None, None), "mul")),
                   med(lambda: ck(lib.kyb_mul_batch(P(a["k"]), P(a["enc"]), None, n, P(a["out"]), None, None), "mul_enc")),
                   med(lambda: ck(lib.kyb_verify_batch(P(a["enc"]), P(a["blob"]), P(a["off"]), P(a["sig"]), n, 1, P(a["st"])), "verify")),
                   med(lambda: ck(lib.kyb_mul_base_batch(P(a["k"]), n, P(a["out"]), None), "mul_base"))]
            got = (a["out"].tobytes(), a["st"].tobytes())
            ref = ref or got
            assert got == ref
            print(f"{n}, {name}, {ip}: " + ", ".join(f"{v:.3f}" for v in row), flush=True)
eng.set_option("host.in_place", 1)
