"""Several host threads, each with its own engine context, issuing DKG-sized host-pointer calls (4,096 items) at the same time.  One such
call leaves most of the chip idle (its kernels are one lane pair per item, one wavefront per SIMD at most); contexts are independent, so
the calls of different threads overlap on the GPU.  Items per second over all threads, per operation.  GPU_MAX_HW_QUEUES=16 is set here
(the host's decision, before the first HIP call) so that the streams of 16 contexts do not share four hardware queues."""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import kyber_rs_amd
import synth

N = 4096
base = kyber_rs_amd.Engine(0)
s = synth.scalars(N, 3)
k = synth.scalars(N, 4, b"k")
enc, ext = base.mul_base(s, want_ext=True)
msgs = kyber_rs_amd.pack_messages(synth.messages(N, 5))
sigs = base.schnorr_sign(s, k, msgs)
want_mul = base.mul(k, pts_ext=ext)
DUR = 1.5
print(f"threads, op, calls_per_s, items_per_s, mean_call_ms   ({N} items per call, host pointers)")
VARIANTS = [(nt, {}) for nt in (1, 2, 4, 8, 16)] + [
    (16, {"coop.share_by_load": 0}),                       # thresholds as set, whatever else is in flight (the behaviour before this option)
    (16, {"verify.overlap": 0}),
    (16, {"_shared_context": 1}),                          # ONE context for all 16 threads (a process-wide default context, as kyb_init gives): the calls are
                                                           # serialised on its mutex, so each one has the chip to itself and must keep the latency kernels —
                                                           # the in-flight count is taken AFTER the mutex (round 3 counted the queued threads too)
]
for nt, opts in VARIANTS:
    shared = opts.pop("_shared_context", 0) if "_shared_context" in opts else 0
    engines = [kyber_rs_amd.Engine(0, private=True) for _ in range(1 if shared else nt)]
    if shared:
        engines = engines * nt
        print(f"# {nt} threads sharing ONE context (mean_call_ms = time a call is served, queueing included)", flush=True)
    try:
        for e in engines:
            for k_, v_ in opts.items():
                e.set_option(k_, v_)
    except kyber_rs_amd.KyberHipError as err:          # a selector of the cross-check build (verify.overlap since round 5): nothing to compare in the product
        print(f"# {nt} threads with {opts}: skipped ({err})", flush=True)
        continue
    if opts:
        print(f"# {nt} threads with {opts}", flush=True)
    for op in ("mul_base", "mul", "sign", "verify"):
        counts = [0] * nt
        ok = [True] * nt
        stop = [0.0]

        def work(i):
            e = engines[i]
            fn = {"mul_base": lambda: e.mul_base(s), "mul": lambda: e.mul(k, pts_ext=ext), "sign": lambda: e.schnorr_sign(s, k, msgs),
                  "verify": lambda: e.verify(enc, msgs, sigs, 1)}[op]
            out = fn()
            if op == "mul":
                ok[i] = bool(np.array_equal(out, want_mul))
            while time.perf_counter() < stop[0]:
                fn()
                counts[i] += 1

        th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
        stop[0] = time.perf_counter() + DUR
        t0 = time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        dt = time.perf_counter() - t0
        assert all(ok)
        calls = sum(counts)
        print(f"{nt}, {op}, {calls / dt:.0f}, {calls * N / dt:.3e}, {dt * nt / max(calls, 1) * 1e3:.3f}", flush=True)
    for e in set(engines):
        e.close()
