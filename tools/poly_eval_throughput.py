"""Throughput of the one-evaluation-per-lane kernel k_poly_eval on a full chip (device-resident), for the record in DESIGN.md:
m polynomials of degree t - 1, k indices each (the verifier side of a large DKG), mads counted from the point operations executed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kyber_rs_amd, synth
eng = kyber_rs_amd.Engine(0)
for t, m, k, bits in ((43, 4096, 64, 10), (171, 1024, 256, 10), (683, 1024, 256, 10)):
    _, ext = eng.mul_base(synth.scalars(t, 5), want_ext=True)
    polys = np.ascontiguousarray(np.tile(ext[None], (m, 1, 1)))
    idx = (np.arange(m * k, dtype=np.uint32).reshape(m, k) * 7919) % ((1 << bits) - 1)
    if len(sys.argv) > 1:
        idx[:] = int(sys.argv[1])                                 # every evaluation at the same index (a DKG verifier's own)
    eng.pubpoly_eval_multi(polys[:2], idx[:2])
    eng.profile_begin(8)
    a = time.perf_counter(); out = eng.pubpoly_eval_multi(polys, idx); dt = time.perf_counter() - a
    prof = eng.profile_read(8)
    kern = sum(v for _, v in prof)
    n = m * k
    # per Horner step: `bits` doublings (4 S + 4 M = 620 mads) + ~bits/2 additions + 1 addition (9 M = 900 mads incl. the cached form)
    mads = t * (bits * 620 + (bits / 2 + 1) * 900)
    print(f"t={t} evaluations={n} ({bits}-bit indices): call {dt*1e3:.1f} ms, kernels {kern:.2f} ms {[(a_, round(b_, 2)) for a_, b_ in prof]} -> "
          f"{n / (kern * 1e-3):.3e} eval/s, ~{n * mads / (kern * 1e-3) / 1e12:.1f} T mad/s", flush=True)
