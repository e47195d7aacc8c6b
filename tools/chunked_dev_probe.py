#!/usr/bin/env python3
"""Would a device-resident 2^20-item variable-base call gain from running as C chunks on C streams (the latency-bound k_mont_prep / k_finish of one
chunk under the ladder of another, as the host-pointer pipeline does)?  One batch on one stream against halves / quarters on two / four caller
streams through the device-pointer ABI (profiles/r05/chunked_dev_probe.log)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import kyber_rs_amd, synth
eng = kyber_rs_amd.Engine(0)
N = 1 << 20
s = torch.from_numpy(synth.scalars(N, 1)).cuda()
k = torch.from_numpy(synth.scalars(N, 2, b"k")).cuda()
ext = torch.empty((N, 40), dtype=torch.int32, device="cuda")
out = torch.empty((N, 32), dtype=torch.uint8, device="cuda")
ref = torch.empty((N, 32), dtype=torch.uint8, device="cuda")
eng.mul_base_dev(s, out_ext=ext); eng.sync()
streams = [torch.cuda.Stream() for _ in range(8)]
def run(chunks, offset_pattern=None):
    per = N // chunks
    for c in range(chunks):
        st = streams[c % len(streams)].cuda_stream
        lo, hi = c * per, (c + 1) * per
        eng.mul_dev(k[lo:hi], pts_ext=ext[lo:hi], out_enc=out[lo:hi], stream=st)
    torch.cuda.synchronize()
def t(fn, reps=10):
    fn(); fn()
    ts = []
    for _ in range(reps):
        a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
    return sorted(ts)[len(ts) // 2] * 1e3
eng.mul_dev(k, pts_ext=ext, out_enc=ref); eng.sync()
for chunks in (1, 2, 4, 8, 1, 2, 4):
    ms = t(lambda: run(chunks))
    print(f"{chunks} chunk(s) on {min(chunks, 8)} stream(s): {ms:.3f} ms per 2^20 items = {N / ms / 1e3:.4g} x10^6 /s, bytes equal: {bool(torch.equal(out, ref))}", flush=True)
