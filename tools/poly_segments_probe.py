"""PubPoly::eval of ONE polynomial at ONE index (a verifier's verify_deal): call time by poly.segments (wavefronts per evaluation), per threshold."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
# kernel variants are selected here: the cross-check build (csrc/Makefile CROSSCHECK=1) — the product library has no such options
os.environ.setdefault("KYB_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber-rs_amd", "libkyber_ed25519_hip_crosscheck.so"))
import kyber_rs_amd, synth
eng = kyber_rs_amd.Engine(0)
print("t: ms per call for poly.segments = 0 (cost model), 1, 4, 6, 8, 11, 16, 22")
for t in (16, 22, 32, 43, 56, 64, 86, 128, 171, 342, 683):
    commits = eng.mul_base(synth.scalars(t, 5), ext_only=True)
    idx = np.array([int(os.environ.get("IDX", "37"))], dtype=np.uint32)
    ref = None
    row = []
    for seg in (0, 1, 4, 6, 8, 11, 16, 22):
        eng.set_option("poly.segments", seg)
        out = eng.pubpoly_eval(commits, idx)
        ref = out if ref is None else ref
        assert np.array_equal(out, ref)
        ts = []
        for _ in range(21):
            a = time.perf_counter(); eng.pubpoly_eval(commits, idx); ts.append(time.perf_counter() - a)
        row.append(sorted(ts)[10] * 1e3)
    print(f"{t}: " + " ".join(f"{v:.3f}" for v in row), flush=True)
eng.set_option("poly.segments", 0)
