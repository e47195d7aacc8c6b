#!/usr/bin/env python3
"""marshal_binary of n points (kyb_encode_batch_dev; the same two kernels close every multiplication): one point per wavefront with the inversion
spread over its lanes (k_finish_coop) against one lane per point with an inversion shared by four (k_encode_batched / k_finish) — where the
hand-over belongs now that the cooperative inversion is safegcd over the lanes (round 6) and no longer a chain of squarings through LDS.
The option is a hand-over size, not a kernel selector: both sides give the same bytes.

  python tools/finish_crossover.py        (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import kyber_rs_amd
import synth

eng = kyber_rs_amd.Engine(0, crosscheck=True)      # finish.four is a selector of the cross-check build (same kernels)
N = 196608
sc = synth.scalars(16000, 5)
eng.set_option("ext.projective", 1)       # projective limbs (small-batch calls asked for limbs only): Z != 1, the encoder has to invert
host_ext = np.tile(np.concatenate([eng.mul_base(sc[i:i + 1000], ext_only=True) for i in range(0, 16000, 1000)]), (N // 16000 + 1, 1))[:N].copy()
eng.set_option("ext.projective", 0)
assert all(list(e[20:30]) != [1] + [0] * 9 for e in host_ext[::97])
ext = torch.from_numpy(host_ext).to("cuda:0")
out = torch.empty((N, 32), dtype=torch.uint8, device="cuda:0")
eng.sync()


def kernel_us(fn, reps=30):
    for _ in range(3):
        fn()
    eng.sync()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return sorted(ts)[len(ts) // 2]


keep = eng.get_option("coop.decode_max_items")
print("points, one per wavefront [us], one per lane and an inversion per wavefront (finish.four = 2) [us], per 4 points of a lane (finish.four = 1) [us]   (device-resident, HIP events around the call)")
for n in (256, 512, 1024, 1536, 2048, 3072, 4096, 8192, 16384, 65536, 131072, 196608):
    if n <= 16384:
        eng.set_option("coop.decode_max_items", n)
        a = kernel_us(lambda: eng.encode_dev(ext[:n], out[:n])); ea = out[:n].cpu().numpy().copy()
    else:
        a, ea = float("nan"), None                           # (far beyond its range)
    eng.set_option("coop.decode_max_items", 0)
    eng.set_option("finish.four", 2)
    w = kernel_us(lambda: eng.encode_dev(ext[:n], out[:n])); ew = out[:n].cpu().numpy().copy()
    eng.set_option("finish.four", 1)
    b = kernel_us(lambda: eng.encode_dev(ext[:n], out[:n])); eb = out[:n].cpu().numpy().copy()
    eng.set_option("finish.four", 2)
    assert (ea is None or np.array_equal(ea, eb)) and np.array_equal(ew, eb)
    print("%6d, %.1f, %.1f, %.1f" % (n, a, w, b), flush=True)
# unmarshal_binary alone (the square root is still a chain of cooperative squarings through LDS): its hand-over is coop.decode_max_items itself
enc = out.clone()
eng.set_option("coop.decode_max_items", 0)
eng.encode_dev(ext, enc); eng.sync()
dec = torch.empty((N, 40), dtype=torch.int32, device="cuda:0")
ok = torch.empty((N,), dtype=torch.uint8, device="cuda:0")
print("encodings, decode one per wavefront [us], one per lane [us]")
for n in (256, 512, 1024, 1536, 2048, 3072, 4096):
    eng.set_option("coop.decode_max_items", n)
    a = kernel_us(lambda: eng.decode_dev(enc[:n], dec[:n], ok[:n])); da = dec[:n].cpu().numpy().copy()
    eng.set_option("coop.decode_max_items", 0)
    b = kernel_us(lambda: eng.decode_dev(enc[:n], dec[:n], ok[:n])); db = dec[:n].cpu().numpy().copy()
    assert np.array_equal(da, db) and bool(ok[:n].all())
    print("%6d, %.1f, %.1f" % (n, a, b), flush=True)
eng.set_option("coop.decode_max_items", keep)
