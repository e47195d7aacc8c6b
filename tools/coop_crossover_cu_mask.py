#!/usr/bin/env python3
"""Do the hand-over sizes between the kernel families move with the compute units the engine really has?  (round-4 review item 6: the defaults were
absolute item counts tuned on 256 CUs.)  Device-resident calls on a caller's stream created with hipExtStreamCreateWithCUMask — all 256 CUs, then 64
(the lowest 64 bits of the mask: 8 CUs on each of the 8 XCDs) with option device.cus = 64 — timed with the one-item-per-wavefront kernels forced on and forced off per batch size;
the crossover (first size at which the batch kernels win) is printed next to the engine's default threshold at that CU count
(profiles/r05/coop_crossover_cu_mask.log)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import kyber_rs_amd
import synth

eng = kyber_rs_amd.Engine(0)
hip = None
for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
    try:
        hip = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
        break
    except OSError:
        pass
assert hip is not None, "HIP runtime not found"
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]


def masked_stream(cus_total, keep_every):
    """the lowest cus_total / keep_every bits of the mask: bit i is compute unit i / 8 of XCD i mod 8 on this part (tools/microbench/cu_mask_map.hip,
    profiles/r05/cu_mask_map.log) — the same number of CUs on every XCD; an XCD WITHOUT a bit in the mask keeps all of its CUs, so sparse masks do nothing"""
    words = (cus_total + 31) // 32
    bits = (1 << (cus_total // keep_every)) - 1
    mask = (ctypes.c_uint32 * words)(*[(bits >> (32 * w)) & 0xffffffff for w in range(words)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask)
    assert rc == 0, f"hipExtStreamCreateWithCUMask failed: {rc}"
    return st.value, bin(bits).count("1")


N = 1 << 14
dev = "cuda:0"
s = torch.from_numpy(synth.scalars(N, 1)).to(dev)
k = torch.from_numpy(synth.scalars(N, 2)).to(dev)
out = torch.empty((N, 32), dtype=torch.uint8, device=dev)
ext = torch.empty((N, 40), dtype=torch.int32, device=dev)
ext2 = torch.empty((N, 40), dtype=torch.int32, device=dev)
pubs = torch.empty((N, 32), dtype=torch.uint8, device=dev)
eng.mul_base_dev(s, out_enc=pubs, out_ext=ext)
eng.sync()
torch.cuda.synchronize()
hw = eng.get_option("device.cus")
KEYS = ("coop.max_items", "coop.base_max_items", "coop.ladder_max_items", "coop.ladder_enc_max_items", "coop.decode_max_items", "coop.verify_max_items")

for keep_every in (1, 4):
    st, cus = masked_stream(hw, keep_every)
    eng.set_option("device.cus", cus if cus != hw else 0)
    defaults = {key: eng.get_option(key) for key in KEYS}

    def t(fn, reps=9):
        fn(); hip.hipStreamSynchronize(st)
        ts = []
        for _ in range(reps):
            a = time.perf_counter(); fn(); hip.hipStreamSynchronize(st); ts.append(time.perf_counter() - a)
        return sorted(ts)[len(ts) // 2] * 1e3

    ops = {"mul_base": (lambda n: eng.mul_base_dev(s[:n], out_enc=out[:n], stream=st), "coop.base_max_items"),
           "mul (limbs in)": (lambda n: eng.mul_dev(k[:n], pts_ext=ext[:n], out_enc=out[:n], stream=st), "coop.ladder_max_items"),
           "mul (bytes in)": (lambda n: eng.mul_dev(k[:n], pts_enc=pubs[:n], out_enc=out[:n], stream=st), "coop.ladder_enc_max_items"),
           "decode": (lambda n: eng.decode_dev(pubs[:n], ext2[:n], stream=st), "coop.decode_max_items")}
    print(f"# stream with {cus} of {hw} compute units (mask: the lowest {cus} bits), device.cus = {eng.get_option('device.cus')}, defaults {defaults}", flush=True)
    grid = sorted({max(1, int(round(cus * f))) for f in (1, 2, 3, 4, 5, 6, 8, 10, 11, 12, 13, 14, 16, 20, 24, 28, 32)})
    for name, (fn, key) in ops.items():
        rows, cross = [], None
        for n in grid:
            ms = []
            for coop in (1, 0):
                for kk in KEYS:      # (coop.verify_max_items also picks the several-wavefronts-per-item forms: it stays at its default when the family is forced ON)
                    eng.set_option(kk, 0 if not coop else (defaults[kk] if kk == "coop.verify_max_items" else 1 << 20))
                ms.append(t(lambda: fn(n)))
            rows.append((n, ms[0], ms[1]))
            if cross is None and ms[1] < ms[0]:
                cross = n
        eng.set_option("device.cus", cus if cus != hw else 0)          # thresholds back to the defaults of this CU count
        print(f"{name:16s} default hand-over {defaults[key]:5d} items ({defaults[key] / cus:.0f} per CU); batch kernels first win at {cross} items ({(cross or 0) / cus:.1f} per CU)")
        print("    n: one-item-per-wavefront ms / batch ms   " + "  ".join(f"{n}: {a:.3f}/{b:.3f}" for n, a, b in rows), flush=True)
eng.set_option("device.cus", 0)
