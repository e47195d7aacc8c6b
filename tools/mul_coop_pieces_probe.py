#!/usr/bin/env python3
"""One variable-base multiplication of FEW items (kyb_mul_batch, host pointers): k_mul_coop with an item's scalar cut into four pieces, each on a
single-wavefront workgroup of its own (kernels_coop.hip), against one wavefront per item — the figures behind the cut (144 / 66 / 31 / 15 bits) and
behind the hand-over 2 n <= coop.verify_max_items.  The option is a hand-over size, not a kernel selector: both sides give the same bytes.

  python tools/mul_coop_pieces_probe.py        (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kyber_rs_amd

eng = kyber_rs_amd.Engine(0)
rng = np.random.default_rng(3)


def med(fn, n=100):
    fn(); fn(); ts = []
    for _ in range(n):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e6


keep = eng.get_option("coop.verify_max_items")
print("items, four workgroups per item [us], one wavefront per item [us]   (32-byte encodings out)")
for n in (1, 4, 16, 32, 64, 96, 128, 192, 256, 384, 512, 768, 1024):
    s = rng.integers(0, 256, (n, 32), dtype=np.uint8); s[:, 31] &= 0x0f
    P = eng.mul_base(rng.integers(0, 256, (n, 32), dtype=np.uint8), ext_only=True)
    eng.set_option("coop.verify_max_items", 4 * n)
    a, ea = med(lambda: eng.mul(s, pts_ext=P)), eng.mul(s, pts_ext=P)
    eng.set_option("coop.verify_max_items", 0)
    b, eb = med(lambda: eng.mul(s, pts_ext=P)), eng.mul(s, pts_ext=P)
    assert np.array_equal(ea, eb)
    print("%5d, %.1f, %.1f" % (n, a, b), flush=True)
eng.set_option("coop.verify_max_items", keep)
# the chain of one item: ladder steps and doublings (a public 64-bit multiplier walks 64 steps on one wavefront: the rest is head, tail and launch)
s = rng.integers(0, 256, (1, 32), dtype=np.uint8); s[0, 31] &= 0x0f
P = eng.mul_base(rng.integers(0, 256, (1, 32), dtype=np.uint8), ext_only=True)
eng.set_option("ext.projective", 1)
four = med(lambda: eng.mul(s, pts_ext=P, ext_only=True), 200)
eng.set_option("coop.verify_max_items", 0)
one = med(lambda: eng.mul(s, pts_ext=P, ext_only=True), 200)
eng.set_option("coop.verify_max_items", keep)
sp = s.copy(); sp[0, 8:] = 0
short = med(lambda: eng.mul(sp, pts_ext=P, ext_only=True, public=True), 200)
eng.set_option("ext.projective", 0)
step = (one - short) / 192
print("one item, projective limbs out: four workgroups %.1f us, one wavefront %.1f us, one wavefront on a public 64-bit multiplier %.1f us" % (four, one, short))
print("  => a ladder step %.3f us, head + tail + launch %.1f us; the top piece's 241 doublings + 15 steps take %.1f us: a doubling %.3f us = %.2f of a step"
      % (step, short - 64 * step, four - (short - 64 * step), (four - (short - 64 * step) - 15 * step) / 241, (four - (short - 64 * step) - 15 * step) / 241 / step))
