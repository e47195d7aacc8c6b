#!/usr/bin/env python3
"""recover_pub_poly's curve work (poly.rs:607-634): t x t products over t shared points, constant-time ladders (kyb_lincomb_batch) against
window tables of the points (kyb_lincomb_public_batch), device-resident operands, kernel times from the engine's own events."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import kyber_rs_amd, oracle_lib, synth
import ctypes

eng = kyber_rs_amd.Engine(0); orc = oracle_lib.Oracle(); dev = torch.device("cuda", 0)
lib = eng.lib
for m, t in ((683, 683), (171, 171), (64, 1024), (2048, 683), (43, 43)):
    sc_np = synth.scalars(m * t, 9).reshape(m, t, 32)
    sc = torch.from_numpy(sc_np).to(dev)
    pts_np = orc.mul_base_ext_batch(synth.scalars(min(t, 64), 10, b"point"))
    pts_np = np.tile(pts_np, ((t + len(pts_np) - 1) // len(pts_np), 1))[:t].copy()
    pts = torch.from_numpy(pts_np).to(dev)
    out = [torch.empty((m, 32), dtype=torch.uint8, device=dev) for _ in range(2)]
    res = {}
    for name, fn, o in (("ladders", lib.kyb_lincomb_batch_dev, out[0]), ("point tables", lib.kyb_lincomb_public_batch_dev, out[1])):
        call = lambda: fn(ctypes.c_void_p(sc.data_ptr()), None, ctypes.c_void_p(pts.data_ptr()), 1, m, t, ctypes.c_void_p(o.data_ptr()), None, None, None)
        assert call() == 0; eng.sync()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter(); assert call() == 0; eng.sync(); ts.append((time.perf_counter() - t0) * 1e3)
        eng.profile_begin(64); call(); eng.sync(); prof = eng.profile_read(64); eng.profile_begin(0)
        ker = {}
        for k, ms in prof: ker[k] = ker.get(k, 0.0) + ms
        res[name] = (min(ts), ker)
    assert torch.equal(out[0], out[1])
    g = m // 2
    assert bytes(out[1][g].cpu().numpy()) == orc.lincomb(sc_np[g], pts_np)
    a, b = res["ladders"], res["point tables"]
    print(f"m={m} t={t}: ladders {a[0]:.3f} ms, point tables {b[0]:.3f} ms ({a[0] / b[0]:.2f}x)   kernels: " + " ".join(f"{k}={v:.3f}" for k, v in b[1].items()), flush=True)
