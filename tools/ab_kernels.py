#!/usr/bin/env python3
"""Same-box A/B of library builds (devices differ by several % and the clock is power-managed: only runs on ONE box,
interleaved, compare).

  python tools/ab_kernels.py libA.so libB.so [...] [--rounds 3] [--steps 20] [--workloads mul,mul_base]

Every (library, round) runs in its own process: ctypes on the raw C ABI (kyb_init, kyb_mul_batch_dev, kyb_mul_base_batch_dev:
signatures unchanged since round 1, so older builds load too), torch only for device memory and events.  Prints the
median step time per library and workload, and the ratio to the first library."""
import argparse
import ctypes
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib_spec, workloads, steps, n):
    import numpy as np
    import torch
    lib_path, *opts = lib_spec.split("@")          # path.so@key=value@key=value: engine options set after kyb_init
    lib = ctypes.CDLL(lib_path)
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    lib.kyb_init.argtypes = [ctypes.c_int]
    lib.kyb_mul_base_batch_dev.argtypes = [vp, sz, vp, vp, vp]
    lib.kyb_mul_batch_dev.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp]
    lib.kyb_schnorr_sign_batch_dev.argtypes = [vp, vp, vp, vp, sz, vp, vp]
    lib.kyb_verify_batch_dev.argtypes = [vp, vp, vp, vp, sz, ctypes.c_int, vp, vp]
    assert lib.kyb_init(0) == 0
    lib.kyb_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int]
    for o in opts:
        key, val = o.split("=")
        assert lib.kyb_set_option(key.encode(), int(val)) == 0, o
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    sc_np = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc_np[:, 31] &= 0x0f
    sc = torch.from_numpy(sc_np).to(dev)
    k = torch.from_numpy(np.roll(sc_np, 1, axis=0).copy()).to(dev)
    pts = torch.empty((n, 40), dtype=torch.int32, device=dev)
    out = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    st_t = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st_t)
    st = vp(st_t.cuda_stream)
    p = lambda t: vp(t.data_ptr())
    assert lib.kyb_mul_base_batch_dev(p(sc), n, None, p(pts), st) == 0
    ns = n // 4
    msgs = torch.from_numpy(rng.integers(0, 256, size=(32 * ns,), dtype=np.uint8)).to(dev)
    off = torch.arange(0, 32 * (ns + 1), 32, dtype=torch.int32, device=dev)
    sigs = torch.empty((n, 64), dtype=torch.uint8, device=dev)
    pubs = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    msgs_v = torch.from_numpy(rng.integers(0, 256, size=(32 * n,), dtype=np.uint8)).to(dev)
    off_v = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int32, device=dev)
    if "verify" in workloads:
        assert lib.kyb_schnorr_sign_batch_dev(p(sc), p(k), p(msgs_v), p(off_v), n, p(sigs), st) == 0
        assert lib.kyb_mul_base_batch_dev(p(sc), n, p(pubs), None, st) == 0
    steps_fn = {
        "mul": lambda: lib.kyb_mul_batch_dev(p(sc), None, p(pts), n, p(out), None, None, st),
        "mul_base": lambda: lib.kyb_mul_base_batch_dev(p(sc), n, p(out), None, st),
        "sign": lambda: lib.kyb_schnorr_sign_batch_dev(p(sc), p(k), p(msgs), p(off), ns, p(out), st),
        "verify": lambda: lib.kyb_verify_batch_dev(p(pubs), p(msgs_v), p(off_v), p(sigs), n, 1, p(out), st),
    }
    res = {}
    for w in workloads:
        fn = steps_fn[w]
        for _ in range(5):
            assert fn() == 0
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        res[w] = statistics.median(a.elapsed_time(b) for a, b in evs)
    print("AB_RESULT " + json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--workloads", default="mul,mul_base")
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    wl = a.workloads.split(",")
    if a.child:
        child(a.child, wl, a.steps, a.n)
        return
    acc = {lib: {w: [] for w in wl} for lib in a.libs}
    for r in range(a.rounds):
        for lib in a.libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), lib, "--child", os.path.abspath(lib.split("@")[0]) + lib[len(lib.split("@")[0]):], "--steps", str(a.steps), "--n", str(a.n),
                                  "--workloads", a.workloads], capture_output=True, text=True)
            line = [ln for ln in out.stdout.split("\n") if ln.startswith("AB_RESULT ")]
            if not line:
                print(f"{lib}: FAILED\n{out.stdout[-500:]}\n{out.stderr[-1500:]}")
                continue
            res = json.loads(line[0][10:])
            for w in wl:
                acc[lib][w].append(res[w])
            print(f"round {r} {os.path.basename(lib)}: " + "  ".join(f"{w} {res[w]:.4f} ms" for w in wl), flush=True)
    base = a.libs[0]
    for lib in a.libs:
        print(os.path.basename(lib) + ": " + "  ".join(
            f"{w} median {statistics.median(acc[lib][w]):.4f} ms (x{statistics.median(acc[lib][w]) / statistics.median(acc[base][w]):.4f} of {os.path.basename(base)})"
            for w in wl if acc[lib][w] and acc[base][w]))


if __name__ == "__main__":
    main()
