#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --memory-copy-trace csv directory: per stream/queue the busy intervals of the LAST call
(everything after the largest idle gap), kernels and copies, as a text timeline."""
import csv, glob, os, sys
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:28], r.get("Queue_Id", "?")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", "copy"))[:28], "dma"))
ev.sort()
if not ev:
    sys.exit("no events")
# the last call = events after the largest gap in the second half
gaps = [(ev[i + 1][0] - max(e[1] for e in ev[:i + 1]), i) for i in range(len(ev) // 2, len(ev) - 1)]
cut = max(gaps)[1] + 1 if gaps else 0
last = ev[cut:]
t0 = last[0][0]
print(f"{len(last)} events in the last call, span {(max(e[1] for e in last) - t0) / 1e6:.3f} ms")
for s, e, k, name, q in last:
    print(f"{(s - t0) / 1e6:9.3f} {(e - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} ms  {k} {q:>4} {name}")
busy_k = sum(e - s for s, e, k, *_ in last if k == "K") / 1e6
busy_c = sum(e - s for s, e, k, *_ in last if k == "C") / 1e6
print(f"kernel time (sum) {busy_k:.3f} ms, copy time (sum) {busy_c:.3f} ms")
