#!/usr/bin/env python3
"""Same-box A/B of one-item host-pointer calls between library builds (KYB_HIP_LIB), every output checked against the oracle:
  python tools/ab_one_item.py libA.so libB.so [--rounds 3]
Each (library, round) runs in its own process; prints the median wall time in microseconds per operation and library."""
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import kyber_rs_amd, oracle_lib, synth
    eng = kyber_rs_amd.Engine(0)
    orc = oracle_lib.Oracle()
    s, k = synth.scalars(8, 5), synth.scalars(8, 6, b"k")
    msgs = synth.messages(8, 7)
    enc, ext = eng.mul_base(s, want_ext=True)
    assert np.array_equal(enc, orc.mul_base_batch(s))
    eng.set_option("ext.projective", 1)
    proj = eng.mul_base(s, ext_only=True)                      # projective limbs (Z != 1): their marshal pays the inversion
    eng.set_option("ext.projective", 0)
    sig = eng.schnorr_sign(s, k, msgs)
    assert np.array_equal(sig, orc.schnorr_sign_batch(s, k, msgs))
    want_mul = orc.mul_batch(k, ext)
    ops = {
        "mul_base": (lambda i: eng.mul_base(s[i:i + 1]), lambda i: enc[i:i + 1]),
        "mul": (lambda i: eng.mul(k[i:i + 1], pts_ext=ext[i:i + 1]), lambda i: want_mul[i:i + 1]),
        "sign": (lambda i: eng.schnorr_sign(s[i:i + 1], k[i:i + 1], msgs[i:i + 1]), lambda i: sig[i:i + 1]),
        "encode_projective": (lambda i: eng.encode(proj[i:i + 1]), lambda i: enc[i:i + 1]),
        "verify": (lambda i: eng.verify(enc[i:i + 1], msgs[i:i + 1], sig[i:i + 1], 1), lambda i: np.zeros(1, np.uint8)),
        "mul_base_64_items": (lambda i: eng.mul_base(np.tile(s, (8, 1))), lambda i: np.tile(enc, (8, 1))),
        "mul_base_2048_items": (lambda i: eng.mul_base(np.tile(s, (256, 1))), lambda i: np.tile(enc, (256, 1))),
        "mul_256_items": (lambda i: eng.mul(np.tile(k, (32, 1)), pts_ext=np.tile(ext, (32, 1))), lambda i: np.tile(want_mul, (32, 1))),
        "mul_2048_items": (lambda i: eng.mul(np.tile(k, (256, 1)), pts_ext=np.tile(ext, (256, 1))), lambda i: np.tile(want_mul, (256, 1))),
        "mul_2816_items": (lambda i: eng.mul(np.tile(k, (352, 1)), pts_ext=np.tile(ext, (352, 1))), lambda i: np.tile(want_mul, (352, 1))),
    }
    out = {}
    for name, (fn, want) in ops.items():
        for i in range(8):
            assert np.array_equal(np.asarray(fn(i)), want(i)), (name, i)
        for _ in range(30):
            fn(0)
        ts = []
        for j in range(300):
            a = time.perf_counter(); fn(j & 7); ts.append(time.perf_counter() - a)
        out[name] = round(statistics.median(ts) * 1e6, 2)
    print("RESULT " + json.dumps(out), flush=True)


def main():
    libs = [a for a in sys.argv[1:] if a.endswith(".so")]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
    res = {lib: [] for lib in libs}
    for r in range(rounds):
        for lib in libs:
            env = dict(os.environ, KYB_HIP_LIB=os.path.abspath(lib))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            if not line:
                print(f"{lib}: FAILED\n{p.stdout[-1500:]}\n{p.stderr[-1500:]}")
                return 1
            res[lib].append(json.loads(line[0][7:]))
            print(os.path.basename(lib), "round", r, res[lib][-1], flush=True)
    for lib in libs:
        med = {k: round(statistics.median(x[k] for x in res[lib]), 2) for k in res[lib][0]}
        print("MEDIAN", os.path.basename(lib), med)
    return 0


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        sys.exit(main())
