"""One-item (and 64-item) call time of every host-pointer entry point the trait surface maps to batch-of-1 calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kyber_rs_amd, synth
eng = kyber_rs_amd.Engine(0)
s = synth.scalars(64, 3)
enc, ext = eng.mul_base(s, want_ext=True)
ext2 = np.roll(ext, 1, axis=0).copy()
msgs = [b"m" * 32] * 64
sigs = eng.schnorr_sign(s, np.roll(s, 1, axis=0).copy(), msgs)


def t(fn, reps=60):
    fn(); fn()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e6


small = np.zeros((64, 32), dtype=np.uint8)      # x = index + 1 of PubPoly::eval (poly.rs:461-464): 513, 514, ...
small[:, 0] = np.arange(1, 65, dtype=np.uint8); small[:, 1] = 2


def ext_only_rows():
    eng.set_option("ext.projective", 1)
    for name, fn in (("mul_base -> ext only, projective", lambda n: eng.mul_base(s[:n], ext_only=True)),
                     ("mul(ext) -> ext only, projective", lambda n: eng.mul(s[:n], pts_ext=ext[:n], ext_only=True)),
                     ("mul(ext) by a 10-bit share index -> ext only, projective", lambda n: eng.mul(small[:n], pts_ext=ext[:n], ext_only=True))):
        print(f"{name}, {t(lambda: fn(1)):.1f}, {t(lambda: fn(64)):.1f}", flush=True)
    eng.set_option("ext.projective", 0)


print("op, n=1 us, n=64 us")
for name, fn in (("encode", lambda n: eng.encode(ext[:n])), ("decode", lambda n: eng.decode(enc[:n])), ("add", lambda n: eng.add(ext[:n], ext2[:n])),
                 ("equal", lambda n: eng.equal(ext[:n], ext2[:n])), ("mul_base", lambda n: eng.mul_base(s[:n])),
                 ("mul(ext)", lambda n: eng.mul(s[:n], pts_ext=ext[:n])), ("mul(ext) by a 10-bit share index", lambda n: eng.mul(small[:n], pts_ext=ext[:n])), ("mul(enc)", lambda n: eng.mul(s[:n], pts_enc=enc[:n])),
                 ("sign", lambda n: eng.schnorr_sign(s[:n], s[:n], msgs[:n])), ("verify", lambda n: eng.verify(enc[:n], msgs[:n], sigs[:n], 1)),
                 ("pubpoly_eval(t=8)", lambda n: eng.pubpoly_eval(ext[:8], np.arange(n, dtype=np.uint32))),
                 ("pubpoly_eval(t=683, index 512)", lambda n: eng.pubpoly_eval(np.tile(ext, (11, 1))[:683], np.full(n, 512, dtype=np.uint32))),
                 ("sum(t=8)", lambda n: eng.sum_points(np.tile(ext[None, :8], (n, 1, 1)))),
                 ("lincomb(t=8)", lambda n: eng.lincomb(np.tile(s[None, :8], (n, 1, 1)), pts_ext=np.tile(ext[None, :8], (n, 1, 1))))):
    print(f"{name}, {t(lambda: fn(1)):.1f}, {t(lambda: fn(64)):.1f}", flush=True)
ext_only_rows()
