#!/usr/bin/env python3
"""Variable base at mid sizes (device-resident calls): two lanes per item (k_mul_ladder_pair) against four (k_mul_ladder_quad) — where the hand-over
ladder.quad_max_items belongs.  Both sides give the same bytes (checked here against each other; against the oracle in tests/).

  python tools/ladder_quad_probe.py        (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import kyber_rs_amd
import synth

eng = kyber_rs_amd.Engine(0)
N = 65536
s = torch.from_numpy(synth.raw256(N, 11)).to("cuda:0")
b = torch.from_numpy(synth.scalars(N, 12)).to("cuda:0")
ext = torch.empty((N, 40), dtype=torch.int32, device="cuda:0")
out = torch.empty((N, 32), dtype=torch.uint8, device="cuda:0")
eng.mul_base_dev(b, out_ext=ext)
eng.sync()


def call_us(fn, reps=15):
    for _ in range(3):
        fn()
    eng.sync()
    ts = []
    for _ in range(reps):
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record(); fn(); b_.record(); b_.synchronize()
        ts.append(a_.elapsed_time(b_) * 1e3)
    return sorted(ts)[len(ts) // 2]


keep = {k: eng.get_option(k) for k in ("ladder.quad_max_items", "coop.max_items", "coop.ladder_max_items")}
print("items, one item per wavefront [us], four lanes per item [us], two lanes per item [us]   (kyb_mul_batch_dev -> encodings, HIP events around the call)")
for n in (1024, 1536, 2048, 2304, 2560, 2816, 3072, 3584, 4096, 8192, 12288, 16384, 20480, 32768):
    c = float("nan")
    if n <= 4096:
        eng.set_option("coop.max_items", 1 << 20); eng.set_option("coop.ladder_max_items", 1 << 20)
        c = call_us(lambda: eng.mul_dev(s[:n], pts_ext=ext[:n], out_enc=out[:n])); ec = out[:n].cpu().numpy().copy()
    eng.set_option("coop.max_items", 0)
    eng.set_option("ladder.quad_max_items", 1 << 20)
    q = call_us(lambda: eng.mul_dev(s[:n], pts_ext=ext[:n], out_enc=out[:n])); eq = out[:n].cpu().numpy().copy()
    eng.set_option("ladder.quad_max_items", 0)
    p = call_us(lambda: eng.mul_dev(s[:n], pts_ext=ext[:n], out_enc=out[:n])); ep = out[:n].cpu().numpy().copy()
    assert np.array_equal(eq, ep) and (n > 4096 or np.array_equal(ec, ep)), n
    print("%6d, %.1f, %.1f, %.1f" % (n, c, q, p), flush=True)
for k_, v in keep.items():
    eng.set_option(k_, v)
