"""verify with the keys given as POINTS (schnorr::verify's own signature) against verify from key bytes, device-resident, per batch size; kernel by kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import kyber_rs_amd, synth
eng = kyber_rs_amd.Engine(0)
N = 32768; dev = "cuda:0"
s = torch.from_numpy(synth.scalars(N, 1)).to(dev); k = torch.from_numpy(synth.scalars(N, 2)).to(dev)
pubs = torch.empty((N, 32), dtype=torch.uint8, device=dev); pext = torch.empty((N, 40), dtype=torch.int32, device=dev)
msgs = torch.from_numpy(np.random.default_rng(3).integers(0, 256, 32 * N, dtype=np.uint8)).to(dev)
off = torch.arange(0, 32 * (N + 1), 32, dtype=torch.int32, device=dev)
sig = torch.empty((N, 64), dtype=torch.uint8, device=dev); st = torch.empty((N,), dtype=torch.uint8, device=dev); st2 = torch.empty((N,), dtype=torch.uint8, device=dev)
eng.mul_base_dev(s, out_enc=pubs, out_ext=pext); eng.sign_dev(s, k, msgs, off, sig); eng.sync()
for n in (256, 768, 1024, 2048, 4096, 8192, 32768):
    for what, fn, res in (("verify(bytes)", lambda: eng.verify_dev(pubs[:n], msgs, off[: n + 1], sig[:n], st[:n], 1), st),
                          ("verify(points)", lambda: eng.verify_points_dev(pext[:n], msgs, off[: n + 1], sig[:n], st2[:n], 1), st2)):
        for _ in range(3): fn()
        eng.sync(); ts = []
        for _ in range(11):
            t0 = time.perf_counter(); fn(); eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        eng.profile_begin(16); fn(); eng.sync(); recs = eng.profile_read(16); eng.profile_begin(0)
        assert not res[:n].cpu().numpy().any()
        print(f"n={n} {what}: call {sorted(ts)[5]:.3f} ms; " + " ".join(f"{nm}={ms:.3f}" for nm, ms in recs), flush=True)
