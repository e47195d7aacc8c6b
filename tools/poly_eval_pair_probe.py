"""PubPoly::eval of m dealers' polynomials (t = 683) at one index each, m between the wavefront shape's range and the lane shape's: the
automatic choice of launch_poly_eval (with the two-lane ladder in its cost model) against forced segment counts of the segment-per-lane shape
and against the one-evaluation-per-wavefront shape.  Kernel time per call (HIP events)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
# kernel variants are selected here: the cross-check build (csrc/Makefile CROSSCHECK=1) — the product library has no such options
os.environ.setdefault("KYB_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber-rs_amd", "libkyber_ed25519_hip_crosscheck.so"))
import kyber_rs_amd, synth
eng = kyber_rs_amd.Engine(0)
t = 683
_, ext = eng.mul_base(synth.scalars(t, 3), want_ext=True)


def kernels(fn):
    best = None
    for _ in range(4):
        eng.profile_begin(32); fn(); prof = eng.profile_read(32)
        k = sum(v for _, v in prof)
        if best is None or k < best[0]:
            best = (k, prof)
    eng.profile_begin(0)
    agg = {}
    for n_, v in best[1]:
        agg[n_] = round(agg.get(n_, 0.0) + v, 3)
    return best[0], agg


for m in (96, 128, 192, 256, 384, 512, 768, 1024, 2048):
    polys = np.tile(ext[None, :, :], (m, 1, 1))
    idx = np.full((m, 1), m // 2 + 300, dtype=np.uint32)
    eng.set_option("poly.segments", 0)
    row = []
    ref = None
    for bs in (0, 1, 16, 32, 64, 128, 256):
        eng.set_option("poly.batch_segments", bs)
        out = eng.pubpoly_eval_multi(polys, idx)
        if ref is None:
            ref = out
        assert np.array_equal(out, ref)
        k, agg = kernels(lambda: eng.pubpoly_eval_multi(polys, idx))
        row.append((bs, round(k, 3), agg if bs == 0 else None))
    eng.set_option("poly.batch_segments", 0)
    print(f"m={m}: automatic {row[0][1]} ms {row[0][2]}; one per wavefront {row[1][1]}; forced segments per lane-shape " + " ".join(f"{b}:{v}" for b, v, _ in row[2:]), flush=True)
