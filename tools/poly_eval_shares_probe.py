"""PubPoly::shares-like calls (ONE polynomial of t coefficients at n indices): kernel time of the automatic launch shape against
forced alternatives — checks the cost model of launch_poly_eval away from the shapes it was fitted on."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
# kernel variants are selected here: the cross-check build (csrc/Makefile CROSSCHECK=1) — the product library has no such options
os.environ.setdefault("KYB_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber-rs_amd", "libkyber_ed25519_hip_crosscheck.so"))
import kyber_rs_amd, synth
eng = kyber_rs_amd.Engine(0)
for n, t in ((700, 100), (1500, 100), (3000, 100), (5000, 683), (20000, 100), (20000, 683), (40000, 683), (60000, 683), (100000, 100), (3000, 24), (900, 2000)):
    _, ext = eng.mul_base(synth.scalars(t, 3), want_ext=True)
    idx = np.arange(n, dtype=np.uint32)
    ref = None
    row = []
    for bs in (0, 1, 2, 4, 16, 64):
        if bs > max(2, t // 4): continue
        eng.set_option("poly.batch_segments", bs)
        out = eng.pubpoly_eval(ext, idx)
        if ref is None: ref = out
        assert np.array_equal(out, ref)
        best = None
        for _ in range(3):
            eng.profile_begin(32)
            eng.pubpoly_eval(ext, idx)
            prof = eng.profile_read(32)
            k = sum(v for _, v in prof)
            if best is None or k < best[0]: best = (k, prof)
        names = sorted({n_ for n_, _ in best[1]})
        row.append(f"{bs}: {best[0]:.2f} ms" + (" [" + "+".join(x.replace("k_", "") for x in names) + "]" if bs == 0 else ""))
    print(f"n={n} t={t}  " + "  ".join(row), flush=True)
