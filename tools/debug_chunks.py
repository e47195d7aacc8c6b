import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch-first":
    import torch
    print("torch first; cuda available:", torch.cuda.is_available())
import kyber_rs_amd, oracle_lib, synth
eng = kyber_rs_amd.Engine(0)
print(eng.device_info())
orc = oracle_lib.Oracle()
for n in (2048, 1 << 15, (1 << 17) + 3):
    rng = np.random.default_rng(n)
    s = rng.integers(0, 256, (n, 32), dtype=np.uint8); s[:, 31] &= 0x0f
    ps = rng.integers(0, 256, (n, 32), dtype=np.uint8); ps[:, 31] &= 0x0f
    t = time.time(); enc_b, ext_b = eng.mul_base(ps, want_ext=True); tb = time.time() - t
    want_b = orc.mul_base_batch(ps, nthreads=16)
    print(n, "mul_base ok:", np.array_equal(enc_b, want_b), "bad", int((enc_b != want_b).any(axis=1).sum()), "%.3fs" % tb)
    for sel in (0, 1):
        eng.set_option("mul.select", sel)
        t = time.time(); got = eng.mul(s, pts_ext=ext_b); tm = time.time() - t
        want = orc.mul_batch(s, ext_b, nthreads=16)
        bad = (got != want).any(axis=1)
        print(n, "mul select", sel, "ok:", not bad.any(), "bad", int(bad.sum()), "first bad", np.nonzero(bad)[0][:8], "%.3fs" % tm)
