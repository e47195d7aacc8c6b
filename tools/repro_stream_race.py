#!/usr/bin/env python3
"""Which side was round 5's red `test_cfg5_whole_2_24_on_one_gpu` on — a kernel at buffers beyond 2^31 bytes, or the ordering between
torch's null stream (which produced the operands) and the engine's own non-blocking stream (which consumed them)?

Three arms on the SAME 2^24-item shape (fixed base vs variable base on P = B, every output compared; a 2^12 sample against the oracle):

  engine_stream_forced   operands written by torch kernels on the null stream behind a deliberately slow null-stream kernel, the engine
                         called with stream = 0 (its own non-blocking stream) and nothing in between: the race made certain.
                         EXPECTED: mismatches (the ladder read points that were not written yet) — and the fixed-base result, whose
                         operand was complete, equals the oracle.
  engine_stream_as_r05   exactly what the round-5 test did (no slow kernel in front, stream = 0 right after `.repeat`): the race as it
                         was, won or lost by luck.  Reported as k failures of N trials.
  current_stream         the binding's default since round 6: the engine's kernels are queued on torch's current stream, the slow kernel
                         still in front.  EXPECTED: N of N trials equal — so the kernels are right at this size and the fault was the
                         ordering.

Prints one JSON line per arm; exit code 0 when the forced arm fails AND the ordered arm passes every trial."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np   # noqa: E402
import torch         # noqa: E402

import kyber_rs_amd  # noqa: E402
import oracle_lib    # noqa: E402


def slow_null_stream_work(dev, gib: int):
    """a few hundred milliseconds of null-stream work: whatever is queued behind it on the null stream starts late"""
    junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    for _ in range(gib):
        junk.random_()
    return junk


def trial(engine, oracle, base_row, s, enc_fixed, arm: str, slow_gib: int, check_oracle: bool):
    n = s.shape[0]
    dev = s.device
    bext = torch.zeros((n, 40), dtype=torch.int32, device=dev)          # what the ladder sees if it runs too early: X = Y = Z = T = 0
    enc_var = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    junk = slow_null_stream_work(dev, slow_gib) if slow_gib else None
    if arm == "engine_stream_ordered_by_caller":
        bext.copy_(base_row.expand(n, 40))
        torch.cuda.synchronize()                                         # what the C ABI asks of a caller that passes NULL
    elif arm == "engine_stream_as_r05":
        engine.mul_base_dev(s, out_enc=enc_var, stream=0)                # as in the round-5 test: the fixed-base kernel holds every CU ...
        bext = base_row.repeat(n, 1)                                     # ... while its line queues a fresh 2.68 GB null-stream write
    else:
        bext.copy_(base_row.expand(n, 40))                               # null stream, behind the slow kernel
    if arm == "current_stream":
        engine.mul_dev(s, pts_ext=bext, out_enc=enc_var)                 # default: torch's current stream
    else:
        engine.mul_dev(s, pts_ext=bext, out_enc=enc_var, stream=0)       # the engine's own non-blocking stream, unordered with the null stream
    engine.sync()
    torch.cuda.synchronize()
    bad = (enc_fixed != enc_var).any(dim=1)
    nbad = int(bad.sum().item())
    out = {"mismatching_items": nbad}
    if nbad:
        first = int(torch.nonzero(bad)[0].item())
        out["first_mismatch"] = first
        out["scalar_top_byte_of_first"] = int(s[first, 31].item())
    if check_oracle:
        idx = torch.randint(0, n, (1 << 12,), generator=torch.Generator().manual_seed(5)).to(dev)
        want = oracle.mul_base_batch(s[idx].cpu().numpy(), nthreads=min(16, len(os.sched_getaffinity(0))))
        out["fixed_base_equals_oracle_on_sample"] = bool(np.array_equal(enc_fixed[idx].cpu().numpy(), want))
        out["variable_base_equals_oracle_on_sample"] = bool(np.array_equal(enc_var[idx].cpu().numpy(), want))
    del bext, enc_var, junk
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=24)
    ap.add_argument("--trials", type=int, default=20)
    ap.add_argument("--slow-gib", type=int, default=6)
    args = ap.parse_args()
    n = 1 << args.log2n
    dev = torch.device("cuda:0")
    engine = kyber_rs_amd.Engine(0)
    oracle = oracle_lib.Oracle()
    g = torch.Generator(device=dev); g.manual_seed(24)
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    s[:, 31] &= 0x1F
    base_row = torch.from_numpy(oracle.base()).to(dev)
    enc_fixed = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    engine.mul_base_dev(s, out_enc=enc_fixed)
    engine.sync()
    torch.cuda.synchronize()
    print(json.dumps({"items": n, "points_bytes": n * 160, "beyond_2_31_bytes": n * 160 > 2**31, "trials": args.trials}), flush=True)

    ok = True
    t0 = time.time()
    # the engine stream's scratch grows on its first 2^24-item call, and growing frees device memory — hipFree waits for the whole device,
    # null stream included, which would hide the race in the first trial: grow it now, with operands that are complete
    warm = trial(engine, oracle, base_row, s, enc_fixed, "engine_stream_ordered_by_caller", 0, False)
    print(json.dumps({"arm": "engine_stream_ordered_by_caller (operands synchronised first)", **warm, "s": round(time.time() - t0, 1)}), flush=True)
    ok &= warm["mismatching_items"] == 0
    forced_fails = 0
    for i in range(3):
        forced = trial(engine, oracle, base_row, s, enc_fixed, "engine_stream_forced", args.slow_gib, True)
        print(json.dumps({"arm": "engine_stream_forced", "trial": i, **forced, "s": round(time.time() - t0, 1)}), flush=True)
        forced_fails += forced["mismatching_items"] > 0
        ok &= forced["fixed_base_equals_oracle_on_sample"]
    ok &= forced_fails == 3

    fails = 0
    for i in range(args.trials):
        r = trial(engine, oracle, base_row, s, enc_fixed, "engine_stream_as_r05", 0, False)
        fails += r["mismatching_items"] > 0
        torch.cuda.empty_cache()
    print(json.dumps({"arm": "engine_stream_as_r05", "failed_trials": fails, "of": args.trials, "s": round(time.time() - t0, 1)}), flush=True)

    fails = 0
    last = None
    for i in range(args.trials):
        last = trial(engine, oracle, base_row, s, enc_fixed, "current_stream", args.slow_gib, i == args.trials - 1)
        fails += last["mismatching_items"] > 0
    print(json.dumps({"arm": "current_stream", "failed_trials": fails, "of": args.trials, **{k: v for k, v in last.items() if k.endswith("sample")},
                      "s": round(time.time() - t0, 1)}), flush=True)
    ok &= fails == 0 and last["variable_base_equals_oracle_on_sample"]
    print(json.dumps({"verdict": "ordering, not the kernels" if ok else "NOT EXPLAINED BY ORDERING"}), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
