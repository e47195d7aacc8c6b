#!/usr/bin/env python3
"""Where the time of a ONE-ITEM call goes (round-5 review item 4: is there a per-call fixed cost left to remove — launch, completion, head, tail —
or is the floor the dependent chain of one item's arithmetic?).

For the one-item host-pointer calls `mul_base` (fixed base + encoding) and `mul` (variable base from limbs + encoding; four workgroups share the
scalar), on the CROSS-CHECK build (the product's sources + phase stamps, csrc/kernels_coop.hip KYB_PHASE):
  wall          time.perf_counter around the synchronous C-ABI call, median of 300 (stamps off)
  launch+wait   wall minus the kernel's duration as the engine's HIP events see it
  kernel        that duration (events on the launch stream)
  in-kernel     the constant 100 MHz clock written by the kernel itself at its phase boundaries (one call with the stamp buffer set):
                first wavefront starts -> scalar multiplication done -> partial results combined -> inversion begins -> inversion ends -> stored
What the events see beyond first-stamp -> last-stamp is the dispatch of the workgroups and the kernel's end-of-grid bookkeeping.
Usage: python tools/one_item_stamps.py  (GPU box) -> profiles/r06/one_item_stamps.log"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np   # noqa: E402
import torch         # noqa: E402

import kyber_rs_amd  # noqa: E402
import synth         # noqa: E402

TICK_US = 0.01       # s_memrealtime: 100 MHz


def main():
    eng = kyber_rs_amd.Engine(0, crosscheck=True)
    prod = kyber_rs_amd.Engine(0)
    s, k = synth.scalars(4, 5), synth.scalars(4, 6, b"k")
    enc, ext = eng.mul_base(s, want_ext=True)
    assert np.array_equal(enc, prod.mul_base(s))
    buf = torch.zeros(32, dtype=torch.int64, device="cuda:0")
    ops = {
        "mul_base -> enc (n = 1)": (lambda e: e.mul_base(s[:1]), "base"),
        "mul(ext) -> enc (n = 1)": (lambda e: e.mul(k[:1], pts_ext=ext[:1]), "mul"),
    }
    for name, (fn, kind) in ops.items():
        rows = {}
        for label, e in (("product", prod), ("crosscheck", eng)):
            for _ in range(30):
                fn(e)
            ts = []
            for _ in range(300):
                a = time.perf_counter(); fn(e); ts.append(time.perf_counter() - a)
            wall = statistics.median(ts) * 1e6
            e.profile_begin(64)
            for _ in range(8):
                fn(e)
            recs = e.profile_read()
            e.profile_begin(0)
            kern = sum(ms for _n, ms in recs) / 8 * 1e3
            rows[label] = (wall, kern, sorted({n for n, _ in recs}))
        print(f"{name}")
        for label, (wall, kern, names) in rows.items():
            print(f"  {label:10s} wall {wall:6.1f} us   kernel(s) by HIP events {kern:6.1f} us   launch + completion {wall - kern:5.1f} us   [{', '.join(names)}]")
        # one stamped call (cross-check build)
        best = None
        for _ in range(20):
            buf.zero_(); torch.cuda.synchronize()
            eng.lib.kyb_diag_phase_stamps(buf.data_ptr())
            fn(eng)
            eng.lib.kyb_diag_phase_stamps(None)
            v = [int(x) for x in buf.cpu().tolist()]
            if kind == "base":
                t = {"start": v[16], "mult": v[17], "combined": v[18], "inv0": v[30], "inv1": v[31], "stored": v[19]}
            else:
                last = max(range(4), key=lambda p: v[8 + p])                    # the piece that arrived last adds and finishes
                t = {"start": min(v[0:4]), "dbl": v[4 + last], "ladder": v[8 + last], "combined": v[12], "inv0": v[30], "inv1": v[31], "stored": v[13],
                     "pieces_done_at": [round((v[8 + p] - min(v[0:4])) * TICK_US, 2) for p in range(4)], "last_piece": last,
                     "piece_starts_after": [round((v[p] - min(v[0:4])) * TICK_US, 2) for p in range(4)]}
            total = (t["stored"] - t["start"]) * TICK_US
            if best is None or total < best[0]:
                best = (total, t)
        total, t = best
        us = lambda a, b: (t[b] - t[a]) * TICK_US      # noqa: E731
        if kind == "base":
            print(f"  in-kernel (100 MHz stamps, best of 20): first stamp -> stored {total:.2f} us = 11 of the 43 windows {us('start', 'mult'):.2f} + other wavefronts' sums in (3 additions) "
                  f"{us('mult', 'combined'):.2f} + to the inversion {us('combined', 'inv0'):.2f} + INVERSION {us('inv0', 'inv1'):.2f} + encode, store {us('inv1', 'stored'):.2f}")
        else:
            print(f"  in-kernel (100 MHz stamps, best of 20): first stamp -> stored {total:.2f} us = the last piece (#{t['last_piece']}): doublings {us('start', 'dbl'):.2f} + ladder, recovery "
                  f"{us('dbl', 'ladder'):.2f} + adding the four {us('ladder', 'combined'):.2f} + to the inversion {us('combined', 'inv0'):.2f} + INVERSION {us('inv0', 'inv1'):.2f} + encode, store "
                  f"{us('inv1', 'stored'):.2f};  pieces done after {t['pieces_done_at']} us, started after {t['piece_starts_after']} us")
        wall, kern, _ = rows["crosscheck"]
        print(f"  so of {wall:.1f} us: dependent arithmetic of the item {total:.1f}, dispatch + end of grid inside the event bracket {kern - total:.1f}, launch call + completion wait {wall - kern:.1f}")
    print("(a CPU core: 22 us fixed base, 63 us variable base — profiles/r05/one_item_breakdown.log; cpu_baseline of the bench line)")


if __name__ == "__main__":
    main()
