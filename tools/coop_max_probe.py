"""coop.max_items: PubPoly::eval at n indices, linear combinations and sums — the one-item-per-wavefront kernels against the batch kernels per size."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kyber_rs_amd, synth
eng = kyber_rs_amd.Engine(0)
def t(fn, reps=11):
    fn(); fn(); ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e3
KEYS = ("coop.max_items", "coop.ladder_max_items", "coop.ladder_enc_max_items", "coop.decode_max_items")
saved = {k: eng.get_option(k) for k in KEYS}
T = 43
commits = eng.mul_base(synth.scalars(T, 5), ext_only=True)
print("n: eval(t=43, n indices) shipped/batch/coop | lincomb(m=n/43 x 43, secret) shipped/batch/coop | sum(m=n/43 x 43) shipped/batch/coop   [ms, host-pointer calls]")
for n in (344, 1032, 2064, 3096, 4128, 6192, 8256, 12384):
    m = n // T
    idx = (np.arange(n) % 1000).astype(np.uint32)
    sc = synth.scalars(m * T, 7).reshape(m, T, 32)
    pts = eng.mul_base(synth.scalars(m * T, 8), ext_only=True).reshape(m, T, 40)
    row = []
    for opts in (saved, {k: 0 for k in KEYS}, {k: 1 << 20 for k in KEYS}):
        for k, v in opts.items(): eng.set_option(k, v)
        row.append((t(lambda: eng.pubpoly_eval(commits, idx)), t(lambda: eng.lincomb(sc, pts_ext=pts)), t(lambda: eng.sum_points(pts))))
    print(f"{n}: eval " + "/".join(f"{r[0]:.3f}" for r in row) + " | lincomb " + "/".join(f"{r[1]:.3f}" for r in row) + " | sum " + "/".join(f"{r[2]:.3f}" for r in row), flush=True)
for k, v in saved.items(): eng.set_option(k, v)
