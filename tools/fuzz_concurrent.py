"""Differential soak with SEVERAL host threads at once: each thread owns an engine context and issues random host-pointer operations of random
sizes (1 .. 6,500 items: every routing boundary, and the load-dependent thresholds move them while the others run); every output is compared
with oracle results computed beforehand.  What tools/fuzz_small_batches.py does for one caller, for the state that callers share: the GPU,
the process-wide call counter, allocations recycled between contexts.

  python tools/fuzz_concurrent.py [seconds] [threads] [seed]

Prints one summary line; exits non-zero at the first mismatch."""
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
import oracle_lib
import synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
orc = oracle_lib.Oracle()
NMAX = 6500
S = synth.raw256(NMAX, 100 + seed)
S[::3] = synth.scalars(len(S[::3]), 200 + seed)
K = synth.scalars(NMAX, 300 + seed, b"k")
P = orc.mul_base_ext_batch(synth.scalars(NMAX, 400 + seed, b"point"))
E = orc.encode_batch(P, nthreads=8)
bad_enc = next(bytes([v]) + bytes(31) for v in range(2, 60) if not orc.decode(bytes([v]) + bytes(31))[1])
E_BAD = E.copy()
E_BAD[::41] = np.frombuffer(bad_enc, dtype=np.uint8)
WANT_BASE = orc.mul_base_batch(S, nthreads=8)
WANT_MUL = orc.mul_batch(K, P, nthreads=8)
X = S.copy(); X[:, 31] &= 0x7f
MSG_LIST = synth.messages(NMAX, 500 + seed)
MSGS = kyber_rs_amd.pack_messages(MSG_LIST)
WANT_SIG = orc.schnorr_sign_batch(X, K, MSG_LIST, nthreads=8)
PUBS = orc.mul_base_batch(X, nthreads=8)
BAD_SIG = WANT_SIG.copy(); BAD_SIG[::4, 35] ^= 0x10; BAD_SIG[2::9, 0] ^= 1
WANT_ST = orc.verify_batch(1, PUBS, MSG_LIST, BAD_SIG, nthreads=8)
COEFFS = synth.scalars(40, 600 + seed)
WANT_SHARES = np.stack([np.frombuffer(orc.pripoly_eval(COEFFS, i), dtype=np.uint8) for i in range(NMAX)])
COMMITS = P[:12].copy()
WANT_EVAL = np.stack([np.frombuffer(orc.pubpoly_eval(COMMITS, i), dtype=np.uint8) for i in range(700)])
LC_SC = K[: 600 * 5].reshape(600, 5, 32)
LC_PT = P[: 600 * 5].reshape(600, 5, 40)
WANT_LC = np.stack([np.frombuffer(orc.lincomb(LC_SC[g], LC_PT[g]), dtype=np.uint8) for g in range(600)])
OPS = ("mul_base", "mul", "mul_enc", "sign", "verify", "encode", "decode", "shares", "eval", "lincomb")
SIZES = [1, 2, 63, 64, 65, 255, 256, 257, 511, 512, 513, 682, 683, 1023, 1024, 1025, 2047, 2048, 2049, 2688, 2689, 3072, 3073, 4096, 4097, 6144, 6145, NMAX]
stop_at = time.time() + budget
failed = []
counts = [dict.fromkeys(OPS, 0) for _ in range(nthreads)]


SHARED = kyber_rs_amd.Engine(0, private=True)          # every third thread works on this ONE context (calls serialised on its mutex: ADVICE r3)


def work(i):
    rng = np.random.default_rng(1000 * seed + i)
    own = i % 3 != 2
    eng = kyber_rs_amd.Engine(0, private=True) if own else SHARED
    try:
        while time.time() < stop_at and not failed:
            if own and rng.integers(0, 40) == 0:                           # contexts come and go while the others work
                eng.close()
                eng = kyber_rs_amd.Engine(0, private=True)
            if own and rng.integers(0, 6) == 0:
                eng.set_option("coop.share_by_load", int(rng.integers(0, 2)))
            op = OPS[int(rng.integers(0, len(OPS)))]
            n = int(rng.choice(SIZES)) if rng.integers(0, 3) else int(rng.integers(1, NMAX + 1))
            lo = int(rng.integers(0, NMAX - n + 1))
            sl = slice(lo, lo + n)
            counts[i][op] += 1
            if op == "mul_base":
                ok = np.array_equal(eng.mul_base(S[sl]), WANT_BASE[sl])
            elif op == "mul":
                ok = np.array_equal(eng.mul(K[sl], pts_ext=P[sl]), WANT_MUL[sl])
            elif op == "mul_enc":
                got, okf = eng.mul(K[sl], pts_enc=E_BAD[sl], want_ok=True)
                badp = (np.arange(lo, lo + n) % 41) == 0
                want = WANT_MUL[sl].copy(); want[badp] = 0; want[badp, 0] = 1
                ok = np.array_equal(okf.astype(bool), ~badp) and np.array_equal(got, want)
            elif op == "sign":
                ok = np.array_equal(eng.schnorr_sign(X[sl], K[sl], MSGS[sl]), WANT_SIG[sl])
            elif op == "verify":
                ok = np.array_equal(eng.verify(PUBS[sl], MSGS[sl], BAD_SIG[sl], 1), WANT_ST[sl])
            elif op == "encode":
                ok = np.array_equal(eng.encode(P[sl]), E[sl])
            elif op == "decode":
                ext, okf = eng.decode(E_BAD[sl])
                badp = (np.arange(lo, lo + n) % 41) == 0
                ok = np.array_equal(okf.astype(bool), ~badp) and np.array_equal(eng.encode(ext[~badp]), E[sl][~badp])
            elif op == "shares":
                ok = np.array_equal(eng.pripoly_eval(COEFFS, np.arange(lo, lo + n, dtype=np.uint32)), WANT_SHARES[sl])
            elif op == "eval":
                m = min(n, 700); a = lo % (700 - m + 1)
                ok = np.array_equal(eng.pubpoly_eval(COMMITS, np.arange(a, a + m, dtype=np.uint32)), WANT_EVAL[a:a + m])
            else:
                m = min(n, 600); a = lo % (600 - m + 1)
                ok = np.array_equal(eng.lincomb(LC_SC[a:a + m], pts_ext=LC_PT[a:a + m]), WANT_LC[a:a + m])
            if not ok:
                failed.append((i, op, n, lo))
    except Exception as ex:  # noqa: BLE001
        failed.append((i, "exception", repr(ex)))
    finally:
        if own:
            eng.close()


th = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
for t_ in th:
    t_.start()
last = time.time()
while any(t_.is_alive() for t_ in th):
    time.sleep(1.0)
    if time.time() - last > 60:
        print(f"... {sum(sum(c.values()) for c in counts)} calls so far", flush=True)
        last = time.time()
for t_ in th:
    t_.join()
SHARED.close()
total = {op: sum(c[op] for c in counts) for op in OPS}
if failed:
    print("MISMATCH", failed[:5], flush=True)
    sys.exit(1)
print(f"fuzz_concurrent: {sum(total.values())} calls from {nthreads} threads in {budget:.0f} s, seed {seed}, all equal to the oracle; per operation {total}", flush=True)
