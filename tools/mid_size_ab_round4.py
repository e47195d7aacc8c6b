"""Same-box A/B of this round's mid-size changes through the HOST-pointer calls (what bench.py's mid_size_calls reports): the two-lane ladder on
the y of wire encodings / the verification's hash-only A half (ladder.y_only) and four items per inversion in the finish (finish.four), each on and
off (ladder.y_only = 2: the decodes as workgroups of the ladder's own launch), interleaved, medians of 25 calls.  Kernel times of the same calls: tools/mid_size_kernels.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

# kernel variants are selected here: the cross-check build (csrc/Makefile CROSSCHECK=1) — the product library has no such options
os.environ.setdefault("KYB_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber-rs_amd", "libkyber_ed25519_hip_crosscheck.so"))
import kyber_rs_amd
import synth

eng = kyber_rs_amd.Engine(0)
NMAX = 32768
s = synth.scalars(NMAX, 81)
k = synth.scalars(NMAX, 82, b"k")
enc, ext = eng.mul_base(s, want_ext=True)
msgs = kyber_rs_amd.pack_messages(synth.messages(NMAX, 83))
sigs = eng.schnorr_sign(s, k, msgs)


def med(fn, reps=25):
    fn(); fn()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e3


print("n, ladder.y_only, finish.four: mul, mul_enc, verify   (ms per host-pointer call, median of 25)")
for n in (4096, 8192, 16384, 32768):
    for rnd in range(2):
        for y_only, four in ((0, 0), (1, 0), (0, 1), (1, 1), (2, 1)):
            eng.set_option("ladder.y_only", y_only)
            eng.set_option("finish.four", four)
            row = [med(lambda: eng.mul(k[:n], pts_ext=ext[:n])), med(lambda: eng.mul(k[:n], pts_enc=enc[:n])), med(lambda: eng.verify(enc[:n], msgs[:n], sigs[:n], 1))]
            print(f"{n}, {y_only}, {four}: " + ", ".join(f"{v:.3f}" for v in row), flush=True)
eng.set_option("ladder.y_only", 2)
eng.set_option("finish.four", 1)
