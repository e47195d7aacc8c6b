"""Differential soak of the small-batch paths against the oracle: random operations, batch sizes on and around every routing
boundary, random option settings (projective hand-over, polynomial segments, single-/multi-wavefront kernels forced on and off),
inputs with quirk scalars, small-order / mixed-order / invalid points and corrupted signatures.

  python tools/fuzz_small_batches.py [seconds] [seed] [product | crosscheck]

product (default): the library that ships, with ITS options (hand-over sizes, projective hand-over, two-lane ladder on / off);
crosscheck: the cross-check build with the kernel-variant selectors in the mix as well (ladder.y_only, finish.four, poly.segments, ...).

Prints one summary line; exits non-zero at the first mismatch (with the case that produced it)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import json

import numpy as np

import kyber_rs_amd
import oracle_lib
import synth

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
CROSSCHECK = len(sys.argv) > 3 and sys.argv[3] == "crosscheck"
rng = np.random.default_rng(seed)
eng = kyber_rs_amd.Engine(0, crosscheck=CROSSCHECK)
orc = oracle_lib.Oracle()
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "kats.json")))
weak = [orc.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
L = synth.L
SIZES = [1, 2, 3, 63, 64, 65, 255, 256, 257, 511, 512, 513, 700, 1023, 1024, 1025, 1536, 1537, 1664, 1665, 2048, 2049, 2304, 2305, 2816, 2817, 3328, 3329, 3584, 3585, 4096, 4097, 4608, 4609, 6144, 6145]
NMAX = max(SIZES)
POOL_S = synth.raw256(NMAX, 1000 + seed)
POOL_S[::3] = synth.scalars(len(POOL_S[::3]), 2000 + seed)
for i, v in enumerate((0, 1, 2, L - 1, L, L + 1, 8 * L - 1, (1 << 255) - 1, 1 << 255, (1 << 256) - 1)):
    POOL_S[7 * i + 5] = np.frombuffer((v % (1 << 256)).to_bytes(32, "little"), dtype=np.uint8)
POOL_P = orc.mul_base_ext_batch(synth.scalars(NMAX, 3000 + seed, b"point"))
for i in range(0, NMAX, 97):
    POOL_P[i] = orc.add(POOL_P[i], weak[2 + (i // 97) % 3])
for i, w in enumerate(weak):
    POOL_P[11 + 13 * i] = w
POOL_P[211] = orc.null()
POOL_E = np.stack([np.frombuffer(orc.encode(p), dtype=np.uint8) for p in POOL_P])
bad_enc = next(bytes([v]) + bytes(31) for v in range(2, 60) if not orc.decode(bytes([v]) + bytes(31))[1])
POOL_E_BAD = POOL_E.copy()
POOL_E_BAD[::53] = np.frombuffer(bad_enc, dtype=np.uint8)
WANT_BASE = orc.mul_base_batch(POOL_S, nthreads=8)
K2 = np.roll(POOL_S, 17, axis=0).copy()
WANT_MUL = orc.mul_batch(K2, POOL_P, nthreads=8)
X = POOL_S.copy(); X[:, 31] &= 0x7f
KN = synth.scalars(NMAX, 4000 + seed, b"k")
MSGS = synth.messages(NMAX, 5000 + seed)
WANT_SIG = orc.schnorr_sign_batch(X, KN, MSGS, nthreads=8)
PUBS = orc.mul_base_batch(X, nthreads=8)
PUBS_EXT = orc.mul_base_ext_batch(X)
BAD_SIG = WANT_SIG.copy(); BAD_SIG[::4, 35] ^= 0x10
BAD_SIG[2::9, 0] ^= 1                                       # R corrupted: usually no longer a point
WANT_ST = {fl: orc.verify_batch(fl, PUBS, MSGS, BAD_SIG, nthreads=8) for fl in (0, 1)} if hasattr(orc, "verify_batch") else None
if WANT_ST is None or len(WANT_ST[0]) != NMAX:
    WANT_ST = {fl: np.array([orc.verify(fl, bytes(PUBS[i]), MSGS[i], bytes(BAD_SIG[i])) for i in range(NMAX)], dtype=np.uint8) for fl in (0, 1)}
COMMITS = POOL_P[:200].copy()


def set_random_options():
    o = {"ext.projective": int(rng.integers(0, 2)),
         "coop.verify_max_items": int(rng.choice([512, 512, 0, 4096])), "ladder.pair_max_items": int(rng.choice([32768, 32768, 0, 1 << 20])),
         "coop.ladder_max_items": int(rng.choice([3584, 2816, 1 << 20, 700])), "coop.ladder_enc_max_items": int(rng.choice([2048, 2048, 1 << 20, 700]))}
    variants = {"poly.segments": int(rng.choice([0, 0, 1, 2, 5, 32])), "poly.batch_segments": int(rng.choice([0, 0, 1, 1, 2, 3, 16, 200])),
                "verify.by_encoding": int(rng.integers(0, 2)), "verify.overlap": int(rng.integers(0, 2)), "ladder.y_only": int(rng.integers(0, 3)), "finish.four": int(rng.integers(0, 3)), "mul_base.quarters": int(rng.integers(0, 2))}
    if CROSSCHECK:          # (drawn in either mode: the same seed walks the same cases on both libraries)
        o.update(variants)
    if rng.integers(0, 4) == 0:              # the batch kernels of DKG-sized calls at these sizes (two-lane ladder, ladder.y_only, finish.four)
        o["coop.max_items"], o["coop.base_max_items"] = 0, 0
        o["coop.verify_max_items"] = int(rng.choice([0, 0, 512]))
    else:
        o["coop.max_items"], o["coop.base_max_items"] = 6144, 4608
    for k_, v in o.items():
        eng.set_option(k_, v)
    return o


counts = {}
t_end = time.time() + budget
t_note = time.time() + 60
cases = 0
while time.time() < t_end:
    if time.time() > t_note:                 # a long run shows that it is alive
        print(f"... {cases} cases so far", flush=True)
        t_note = time.time() + 60
    opts = set_random_options()
    n = int(rng.choice(SIZES)) if rng.integers(0, 3) else int(rng.integers(1, 900))
    lo = int(rng.integers(0, NMAX - n + 1))
    sl = slice(lo, lo + n)
    op = str(rng.choice(["mul_base", "mul_ext", "mul_enc", "sign", "verify", "decode", "encode", "eval", "eval_wire", "lincomb", "lincomb_pub", "dkg_round", "lagrange", "pripoly", "sum", "sum_wire"]))
    counts[op] = counts.get(op, 0) + 1
    cases += 1
    ctx = (op, n, lo, opts)
    try:
        if op == "mul_base":
            if rng.integers(0, 2):
                assert np.array_equal(eng.mul_base(POOL_S[sl]), WANT_BASE[sl])
            else:
                assert np.array_equal(eng.encode(eng.mul_base(POOL_S[sl], ext_only=True)), WANT_BASE[sl])
        elif op == "mul_ext" and n <= 64 and rng.integers(0, 2):
            # short public multipliers (share indices, the cofactor): the ladder skips the leading zeros when every scalar of the call is below 2^64
            bits = int(rng.choice([1, 4, 10, 32, 63, 64, 65]))
            sc = np.zeros((n, 32), dtype=np.uint8)
            sc[:, :9] = rng.integers(0, 256, (n, 9), dtype=np.uint8)
            for i in range(n):
                v = int.from_bytes(bytes(sc[i]), "little") & ((1 << bits) - 1)
                sc[i] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
            eng.set_option("mul.short_scalars", int(rng.integers(0, 4) == 0))      # the option, or (mostly) the explicit public call
            pub = bool(rng.integers(0, 4) != 0)
            want = orc.mul_batch(sc, POOL_P[sl])
            if opts["ext.projective"]:
                assert np.array_equal(eng.encode(eng.mul(sc, pts_ext=POOL_P[sl], ext_only=True, public=pub)), want)
            else:
                assert np.array_equal(eng.mul(sc, pts_ext=POOL_P[sl], public=pub), want)
            eng.set_option("mul.short_scalars", 0)
        elif op == "mul_ext":
            if rng.integers(0, 2):
                assert np.array_equal(eng.mul(K2[sl], pts_ext=POOL_P[sl]), WANT_MUL[sl])
            else:
                assert np.array_equal(eng.encode(eng.mul(K2[sl], pts_ext=POOL_P[sl], ext_only=True)), WANT_MUL[sl])
        elif op == "mul_enc":
            got, ok = eng.mul(K2[sl], pts_enc=POOL_E_BAD[sl], want_ok=True)
            for i in range(n):
                badp = ((lo + i) % 53) == 0
                assert bool(ok[i]) == (not badp)
                assert bytes(got[i]) == (bytes([1] + [0] * 31) if badp else bytes(WANT_MUL[lo + i]))
        elif op == "sign":
            keyed = rng.integers(0, 2)
            assert np.array_equal(eng.schnorr_sign(X[sl], KN[sl], MSGS[lo:lo + n], pubs=PUBS[sl] if keyed else None), WANT_SIG[sl])
        elif op == "verify":
            fl = int(rng.integers(0, 2))
            if rng.integers(0, 3) == 0:                        # the public keys as points (schnorr::verify): same statuses
                assert np.array_equal(eng.verify_points(PUBS_EXT[sl], MSGS[lo:lo + n], BAD_SIG[sl], fl), WANT_ST[fl][sl])
            else:
                assert np.array_equal(eng.verify(PUBS[sl], MSGS[lo:lo + n], BAD_SIG[sl], fl), WANT_ST[fl][sl])
        elif op == "decode":
            ext, ok = eng.decode(POOL_E_BAD[sl])
            for i in range(0, n, max(1, n // 40)):
                badp = ((lo + i) % 53) == 0
                assert bool(ok[i]) == (not badp)
                if not badp:
                    assert orc.encode(ext[i]) == bytes(POOL_E[lo + i])
        elif op == "encode":
            assert np.array_equal(eng.encode(POOL_P[sl]), POOL_E[sl])
        elif op == "eval":
            t = int(rng.choice([1, 2, 9, 48, 49, 150, 200]))
            m = min(n, 40)
            idx = rng.integers(0, 1 << int(rng.choice([1, 4, 10, 16, 32])), m, dtype=np.uint64).astype(np.uint32)
            idx[idx == 0xffffffff] = 7
            got = eng.pubpoly_eval(COMMITS[:t], idx)
            for i in range(0, m, max(1, m // 6)):
                assert bytes(got[i]) == orc.pubpoly_eval(COMMITS[:t], int(idx[i]))
        elif op == "eval_wire":
            t = int(rng.choice([1, 3, 20, 60]))
            m = max(1, min(n, 1200) // t)
            k_ = int(rng.choice([1, 2]))
            enc_in = POOL_E_BAD[lo:lo + m * t] if lo + m * t <= NMAX else POOL_E_BAD[:m * t]
            base_i = lo if lo + m * t <= NMAX else 0
            idx = rng.integers(0, 1 << int(rng.choice([1, 4, 10, 16])), (m, k_), dtype=np.uint64).astype(np.uint32)
            got, ok = eng.pubpoly_eval_multi_enc(enc_in.reshape(m, t, 32), idx)
            for g_ in range(0, m, max(1, m // 6)):
                pts = [orc.null() if ((base_i + g_ * t + j) % 53) == 0 else POOL_P[base_i + g_ * t + j] for j in range(t)]
                assert [bool(v) for v in ok[g_]] == [((base_i + g_ * t + j) % 53) != 0 for j in range(t)]
                assert bytes(got[g_, k_ - 1]) == orc.pubpoly_eval(np.stack(pts), int(idx[g_, k_ - 1]))
        elif op == "sum_wire":
            t = int(rng.choice([1, 2, 5, 16, 40]))
            m = max(1, min(n, 1200) // t)
            major = bool(rng.integers(0, 2))
            enc_in = POOL_E_BAD[:m * t].reshape(m, t, 32)
            arg = np.ascontiguousarray(enc_in.transpose(1, 0, 2)) if major else enc_in
            got, ok = eng.sum_points_enc(arg, item_major=major)
            okm = ok.T if major else ok
            for g_ in range(0, m, max(1, m // 8)):
                acc = orc.null()
                for j in range(t):
                    bad_ = ((g_ * t + j) % 53) == 0
                    assert bool(okm[g_, j]) == (not bad_)
                    if not bad_:
                        acc = orc.add(acc, POOL_P[g_ * t + j])
                assert bytes(got[g_]) == orc.encode(acc)
        elif op == "lincomb":
            t = int(rng.choice([1, 2, 3, 8, 33]))
            m = max(1, min(n, 2000) // t)
            sc = K2[lo:lo + m * t].reshape(m, t, 32) if lo + m * t <= NMAX else K2[:m * t].reshape(m, t, 32)
            pp = POOL_P[:m * t].reshape(m, t, 40)
            got = eng.lincomb(sc, pts_ext=pp)
            for g_ in range(0, m, max(1, m // 8)):
                assert bytes(got[g_]) == orc.lincomb(sc[g_], pp[g_])
        elif op == "lincomb_pub":
            # public multipliers: shared points in large enough shapes take the per-point window tables, the rest the short/plain ladders
            t = int(rng.choice([1, 2, 3, 16, 33, 130]))
            shared = bool(rng.integers(0, 2))
            m = max(1, int(rng.choice([n, 6 * n, 12000])) // t) if shared else max(1, min(n, 2000) // t)
            m = min(m, 12000 // t + 1)
            sc = np.ascontiguousarray(np.resize(K2, (m * t, 32))).reshape(m, t, 32).copy()
            if rng.integers(0, 3) == 0:                                        # short multipliers as well
                sc[:, :, int(rng.choice([1, 4, 8])):] = 0
            pp = POOL_P[lo % 64:lo % 64 + t] if shared else POOL_P[:m * t].reshape(m, t, 40)
            got = eng.lincomb(sc, pts_ext=pp, public=True)
            for g_ in sorted({0, m - 1, *range(0, m, max(1, m // 6))}):
                assert bytes(got[g_]) == orc.lincomb(sc[g_], pp if shared else pp[g_])
        elif op == "dkg_round":
            t = int(rng.choice([1, 2, 3, 20, 60]))
            m = max(1, min(n, 1500) // t)
            base_i = lo if lo + m * t <= NMAX else 0
            enc_in = POOL_E_BAD[base_i:base_i + m * t].reshape(m, t, 32)
            index = int(rng.integers(0, 1 << int(rng.choice([1, 4, 10, 16, 32])))) % 0xffffffff
            sums_wanted = bool(rng.integers(0, 2))
            ev, sums, ok = eng.dkg_verify_round_enc(enc_in, index, want_sums=sums_wanted)
            good = np.array([[((base_i + g_ * t + j) % 53) != 0 for j in range(t)] for g_ in range(m)])
            assert np.array_equal(ok.astype(bool), good)
            for g_ in sorted({0, m - 1, *range(0, m, max(1, m // 5))}):
                pts = [POOL_P[base_i + g_ * t + j] if good[g_, j] else orc.null() for j in range(t)]
                assert bytes(ev[g_]) == orc.pubpoly_eval(np.stack(pts), index)
            if sums_wanted:
                for j in sorted({0, t - 1, t // 2}):
                    acc = orc.null()
                    for g_ in range(m):
                        if good[g_, j]:
                            acc = orc.add(acc, POOL_P[base_i + g_ * t + j])
                    assert bytes(sums[j]) == orc.encode(acc)
        elif op == "pripoly":
            t = int(rng.choice([1, 2, 7, 8, 64, 200, 900]))
            k_ = max(1, min(n, 3000))
            m = int(rng.choice([1, 1, 3]))
            cf = np.resize(POOL_S[lo:] if lo + 8 < NMAX else POOL_S, (m * t, 32)).reshape(m, t, 32).copy()      # canonical, unreduced and quirk scalars
            idx = rng.integers(0, 1 << int(rng.choice([1, 10, 16, 32])), k_, dtype=np.uint64).astype(np.uint32)
            idx[idx == 0xffffffff] = 3
            got = eng.pripoly_eval(cf, idx)
            for g_ in sorted({0, m - 1}):
                for i in sorted({0, k_ // 2, k_ - 1}):
                    assert bytes(got[g_, i]) == orc.pripoly_eval(cf[g_], int(idx[i]))
        elif op == "lagrange":
            t = int(rng.choice([1, 2, 3, 17, 64, 200]))
            m = max(1, min(n, 1200) // t)
            idx = np.stack([np.sort(rng.choice(1 << int(rng.choice([9, 16, 31])), t, replace=False)) for _ in range(m)]).astype(np.uint32)
            lam = eng.lagrange_coeffs(idx)
            for g_ in sorted({0, m - 1}):
                xs = [int(v) + 1 for v in idx[g_]]
                for i in sorted({0, t // 2, t - 1}):
                    num = den = 1
                    for j in range(t):
                        if j != i:
                            num = num * xs[j] % L
                            den = den * (xs[j] - xs[i]) % L
                    assert int.from_bytes(bytes(lam[g_, i]), "little") == num * pow(den, L - 2, L) % L
        else:
            t = int(rng.choice([1, 2, 5, 16]))
            m = max(1, min(n, 1200) // t)
            pp = POOL_P[:m * t].reshape(m, t, 40)
            got = eng.sum_points(pp)
            for g_ in range(0, m, max(1, m // 8)):
                acc = pp[g_, 0]
                for j in range(1, t):
                    acc = orc.add(acc, pp[g_, j])
                assert bytes(got[g_]) == orc.encode(acc)
    except AssertionError:
        print("MISMATCH", ctx, flush=True)
        raise
print(f"fuzz_small_batches: {cases} cases in {budget:.0f} s, seed {seed}, all equal to the oracle; per operation {dict(sorted(counts.items()))}", flush=True)
