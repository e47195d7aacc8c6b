"""Kernel time of the verifier's side of a DKG round (m dealers' polynomials of t coefficients, each at the node's own index)
per number of wavefronts (segments) an evaluation is spread over — sets the automatic choice in launch_poly_eval."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
# kernel variants are selected here: the cross-check build (csrc/Makefile CROSSCHECK=1) — the product library has no such options
os.environ.setdefault("KYB_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber-rs_amd", "libkyber_ed25519_hip_crosscheck.so"))
import kyber_rs_amd, synth
eng = kyber_rs_amd.Engine(0)
for m, t in ((1024, 683), (256, 171), (64, 43), (4096, 683), (2048, 171), (8192, 683), (16384, 171), (512, 342)):
    _, ext = eng.mul_base(synth.scalars(t, 3), want_ext=True)
    polys = np.tile(ext[None, :, :], (m, 1, 1))
    idx = np.full((m, 1), m // 2, dtype=np.uint32)
    ref = None
    eng.set_option("poly.segments", 0)
    for bs in (0, 4, 8, 16, 32, 64, 128):
        if bs > t // 2: continue
        eng.set_option("poly.batch_segments", bs)
        out = eng.pubpoly_eval_multi(polys, idx)
        if ref is None: ref = out
        assert np.array_equal(out, ref)
        best = None
        for _ in range(4):
            eng.profile_begin(32)
            a = time.perf_counter(); eng.pubpoly_eval_multi(polys, idx); dt = time.perf_counter() - a
            prof = eng.profile_read(32)
            k = sum(v for _, v in prof)
            if best is None or k < best[1]: best = (dt, k, prof)
        agg = {}
        for n_, v in best[2]: agg[n_] = agg.get(n_, 0.0) + v
        print(f"m={m} t={t} batch_segments={bs}: call {best[0]*1e3:.2f} ms, kernels {best[1]:.3f} ms {[(n_, round(v, 3)) for n_, v in agg.items()]}", flush=True)
    eng.set_option("poly.batch_segments", 1)
    for segs in (0, 1, 2, 3, 4, 6, 8, 12, 16, 32):
        eng.set_option("poly.segments", segs)
        out = eng.pubpoly_eval_multi(polys, idx)
        if ref is None: ref = out
        assert np.array_equal(out, ref)
        best = None
        for _ in range(4):
            eng.profile_begin(8)
            a = time.perf_counter(); eng.pubpoly_eval_multi(polys, idx); dt = time.perf_counter() - a
            prof = eng.profile_read(8)
            k = sum(v for _, v in prof)
            if best is None or k < best[1]: best = (dt, k, prof)
        print(f"m={m} t={t} segs={segs}: call {best[0]*1e3:.2f} ms, kernels {best[1]:.3f} ms {[(n_, round(v, 3)) for n_, v in best[2]]}", flush=True)
