"""Runs the DEVICE arithmetic source (kyber-rs_amd/csrc/*.h) compiled by g++ against the oracle.

Purpose: catch limb-bound and formula errors in the code the HIP kernels inline, on a machine without
a GPU.  Every 32x32 multiply and every limb add/sub in that build is shadowed by an overflow check
(KYB_HOST_TEST).  This is a test of the device source, not a CPU implementation of the product: the
library built here lives under tests/ and nothing outside tests/ loads it.  CPU only."""
import ctypes
import gzip
import hashlib
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bigint_model as M  # noqa: E402

KATS = json.load(open(os.path.join(HERE, "golden", "kats.json")))


@pytest.fixture(scope="module")
def hd():
    src = os.path.join(HERE, "hostcheck", "device_src_host.cpp")
    out = os.path.join(HERE, "hostcheck", "_build", "libdevsrc_host.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    deps = [src] + [os.path.join(ROOT, "kyber-rs_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "kyber-rs_amd", "csrc")) if f.endswith((".h", ".inc"))]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-o", out, src])
    lib = ctypes.CDLL(out)
    lib.hd_overflows.restype = ctypes.c_long
    lib.hd_decode.restype = ctypes.c_int
    return lib


def B(n):
    return ctypes.create_string_buffer(n)


def p32(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_field_ops_match_integers(hd):
    rnd = random.Random(1)
    mask = 2**255 - 1
    for _ in range(3000):
        a = bytes(rnd.getrandbits(8) for _ in range(32)); b = bytes(rnd.getrandbits(8) for _ in range(32)); o = B(32)
        ai, bi = int.from_bytes(a, "little") & mask, int.from_bytes(b, "little") & mask
        hd.hd_fe_mul(o, a, b, 0)
        assert int.from_bytes(o.raw, "little") == ai * bi % M.P
        hd.hd_fe_mul(o, a, b, 1)
        assert int.from_bytes(o.raw, "little") == ai * ai % M.P
    for v in [0, 1, 2, M.P - 1, M.P, M.P + 18, 2**255 - 1, 19, 2**254]:
        o = B(32)
        hd.hd_fe_invert(o, v.to_bytes(32, "little"))
        assert int.from_bytes(o.raw, "little") == pow(v % M.P, M.P - 2, M.P)
    assert hd.hd_overflows() == 0


def test_divstep_inversion_equals_the_exponentiation(hd):
    """fe_invert_gcd (600 constant-time divsteps, fe_invert_gcd.h) == z^(p-2) mod p: edge values (0 -> 0, 1, p - 1, values >= p, powers of
    two), 4000 random values, and loose limbs (3 x tight) against the exponentiation on the same element"""
    rnd = random.Random(7)
    P = M.P
    vals = [0, 1, 2, 3, 19, 38, P - 1, P - 2, P - 19, P, P + 1, P + 18, 2**255 - 1, (P - 1) // 2, (P + 1) // 2, 2**254, 2**254 - 1, 2**30, 2**30 - 1]
    vals += [1 << k for k in range(0, 255, 5)] + [P - (1 << k) for k in range(1, 254, 9)] + [rnd.getrandbits(255) for _ in range(4000)]
    for v in vals:
        o = B(32)
        hd.hd_fe_invert_gcd(o, v.to_bytes(32, "little"))
        assert int.from_bytes(o.raw, "little") == pow(v % P, P - 2, P), hex(v)
    for v in vals[:200]:
        o, o2 = B(32), B(32)
        hd.hd_fe_invert_gcd_loose(o, o2, v.to_bytes(32, "little"))
        assert o.raw == o2.raw and int.from_bytes(o.raw, "little") == pow(3 * v % P, P - 2, P), hex(v)
    assert hd.hd_overflows() == 0


def test_limb_bound_contract(hd):
    """fe_mul(f <= 6T, g <= 3.3T) and fe_sq(f <= 3.3T) never overflow; just beyond, they do."""
    base = hd.hd_overflows()
    for kf, kg in [(600, 330), (630, 336), (101, 101), (500, 101), (300, 300)]:
        hd.hd_fe_mul_bound_probe(kf, kg)
    for k in (101, 202, 330, 336):
        hd.hd_fe_sq_bound_probe(k)
    assert hd.hd_overflows() == base
    hd.hd_fe_mul_bound_probe(640, 337)
    hd.hd_fe_sq_bound_probe(340)
    assert hd.hd_overflows() > base
    # short-fold variants (top carry must fit 32 bits): bound(f) * bound(g) <= 6.3 resp. f <= 2.5T
    base = hd.hd_overflows()
    for kf, kg in [(300, 200), (200, 300), (600, 101), (190, 330), (505, 101), (250, 250), (630, 100), (210, 300)]:
        hd.hd_fe_mul_b6_bound_probe(kf, kg)
    for k in (101, 202, 250):
        hd.hd_fe_sq_b2_bound_probe(k)
    assert hd.hd_overflows() == base
    hd.hd_fe_mul_b6_bound_probe(330, 200)
    assert hd.hd_overflows() > base
    base = hd.hd_overflows()
    hd.hd_fe_sq_b2_bound_probe(262)
    assert hd.hd_overflows() > base


def test_recoding_matches_reference_recode(hd):
    """sc_recode (one 256-bit add) == the reference's carry sweep (ge.rs:443-459), incl. e[63] in 9..16 -> 0"""
    rnd = random.Random(2)
    cases = [bytes(32), bytes([255] * 32), bytes([0x88] * 32), bytes([0x77] * 32), bytes([0] * 31 + [0x80])]
    cases += [bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(2000)]
    for s in cases:
        e = (ctypes.c_int8 * 64)()
        hd.hd_recode(e, s)
        want = M.recode(s)
        if not (0 <= want[63] <= 8):
            want[63] = 0
        assert list(e) == want


def test_scalar_mult_matches_oracle(hd, oracle):
    base = hd.hd_overflows()
    rnd = random.Random(3)
    for q in KATS["quirk_mul_base"]:
        o = B(32); hd.hd_mul_base(o, bytes.fromhex(q["scalar"]))
        assert o.raw.hex() == q["out"]
    for q in KATS["quirk_mul"]:
        if not q["ok"]:
            continue
        ext, _ = oracle.decode(bytes.fromhex(q["point"]))
        o = B(32); hd.hd_mul(o, None, bytes.fromhex(q["scalar"]), p32(ext))
        assert o.raw.hex() == q["out"], q
    for _ in range(150):
        s = bytes(rnd.getrandbits(8) for _ in range(32))
        pt = oracle.mul_base_ext(bytes(rnd.getrandbits(8) for _ in range(32)))
        o = B(32); ext = np.zeros(40, dtype=np.int32)
        hd.hd_mul(o, p32(ext), s, p32(pt))
        assert o.raw == oracle.mul(s, pt)
        assert oracle.encode(ext) == o.raw and list(ext[20:30]) == [1] + [0] * 9
        # reference limb bounds (fe.rs:275-281) and fe_from_bytes normal form: Y limbs equal the oracle's decode
        assert all(abs(int(v)) <= (1 << 25 if i % 2 == 0 else 1 << 24) for i, v in enumerate(ext))
        assert list(ext[10:20]) == list(oracle.decode(o.raw)[0][10:20])
        # and the exported point is usable by the reference arithmetic: s2 * (ext) through the oracle
        assert oracle.mul(s, ext) == oracle.mul(s, oracle.decode(o.raw)[0])
        hd.hd_mul_base(o, s)
        assert o.raw == oracle.mul_base(s)
    assert hd.hd_overflows() == base


def test_decode_encode_add(hd, oracle):
    base = hd.hd_overflows()
    rnd = random.Random(4)
    encs = [bytes.fromhex(h) for h in KATS["weak_keys"] + KATS["invalid_encodings"] + [KATS["decode_kat"]]]
    encs += [(M.P + k).to_bytes(32, "little") for k in range(19)]
    encs += [bytes([1] + [0] * 30 + [0x80])]
    encs += [bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(300)]
    good = []
    for e in encs:
        ext = np.zeros(40, dtype=np.int32)
        ok = hd.hd_decode(p32(ext), e)
        oext, ook = oracle.decode(e)
        assert ok == ook, e.hex()
        if ok:
            assert oracle.encode(ext) == oracle.encode(oext)
            o = B(32); hd.hd_encode(o, p32(oext))
            assert o.raw == oracle.encode(oext)
            good.append(oext)
    for i in range(len(good) - 1):
        for sub in (0, 1):
            out = np.zeros(40, dtype=np.int32)
            hd.hd_add(p32(out), p32(good[i]), p32(good[i + 1]), sub)
            assert oracle.encode(out) == oracle.encode(oracle.add(good[i], good[i + 1], sub=bool(sub)))
    assert hd.hd_overflows() == base


def test_power_of_a_share_index_mod_8L(hd):
    """sc_pow_mod8L_signed (the multipliers of the segmented polynomial evaluation): x^e mod 8L as the signed representative of
    smallest magnitude, against Python integers; even and odd x, exponents 0 .. 2^24"""
    import ctypes
    import random
    L = 2**252 + 27742317777372353535851937790883648493
    rnd = random.Random(8)
    cases = [(1, 0), (1, 5), (2, 1), (2, 2), (2, 3), (2, 300), (4, 2), (6, 3), (3, 1), (7, 682), (1024, 683), (513, 341), (0xffffffff, 1), (0xffffffff, 0xffffff)]
    cases += [(rnd.randrange(1, 1 << 32), rnd.randrange(0, 1 << 24)) for _ in range(60)] + [(rnd.randrange(1, 2000), rnd.randrange(0, 2000)) for _ in range(60)]
    for x, e in cases:
        mag = ctypes.create_string_buffer(32)
        neg = ctypes.c_uint32(7)
        hd.hd_sc_pow_mod8L(ctypes.c_uint32(x), ctypes.c_uint32(e), mag, ctypes.byref(neg))
        v = int.from_bytes(mag.raw, "little")
        assert v < 4 * L + 1 and neg.value in (0, 1), (x, e)
        assert ((-v if neg.value else v) - pow(x, e, 8 * L)) % (8 * L) == 0, (x, e)


def test_scalar_mod_l_and_sha512(hd):
    rnd = random.Random(5)
    edge = [bytes(32), bytes([255] * 32), M.L.to_bytes(32, "little"), (M.L - 1).to_bytes(32, "little")]
    cases = [(a, b, c) for a in edge for b in edge for c in edge]
    cases += [tuple(bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(3)) for _ in range(2000)]
    for a, b, c in cases:
        o = B(32); hd.hd_sc_muladd(o, a, b, c)
        assert int.from_bytes(o.raw, "little") == (int.from_bytes(a, "little") * int.from_bytes(b, "little") + int.from_bytes(c, "little")) % M.L
    for x in [bytes(64), bytes([255] * 64)] + [bytes(rnd.getrandbits(8) for _ in range(64)) for _ in range(2000)]:
        o = B(32); hd.hd_sc_reduce512(o, x)
        assert int.from_bytes(o.raw, "little") == int.from_bytes(x, "little") % M.L
    for n in [0, 1, 55, 111, 112, 113, 127, 128, 129, 239, 240, 241, 300, 1000]:
        m = bytes(rnd.getrandbits(8) for _ in range(n)); o = B(64)
        hd.hd_sha512(o, m, n)
        assert o.raw == hashlib.sha512(m).digest()
        for cut in sorted({c for c in (0, 1, 3, 7, 8, 9, 63, 64, 65, 120, 127, 128, 131, n // 2, n - 1, n) if 0 <= c <= n}):      # sha512_bytes from any block position
            o = B(64); hd.hd_sha512_split(o, m, n, cut)
            assert o.raw == hashlib.sha512(m).digest(), (n, cut)


def test_sign_golden_lines(hd, oracle):
    """device signing source on the first 96 golden EdDSA lines re-expressed as (x, k, msg) triples"""
    lines = gzip.open(os.path.join(HERE, "golden", "sign.input.gz"), "rt").read().split("\n")[:96]
    for ln in lines:
        p = ln.split(":")
        seed, msg, sig = bytes.fromhex(p[0])[:32], bytes.fromhex(p[2]), bytes.fromhex(p[3])[:64]
        x, prefix, _ = oracle.eddsa_expand(seed)
        k = oracle.sc_reduce64(hashlib.sha512(prefix + msg).digest())
        o = B(64); hd.hd_schnorr_sign(o, x, k, msg, len(msg))
        assert o.raw == sig


def verify_cases(oracle):
    """(pub, msg, sig) triples: valid golden signatures, every reject reason, and random corruptions"""
    lines = gzip.open(os.path.join(HERE, "golden", "sign.input.gz"), "rt").read().split("\n")[:24]
    cases = []
    neg = KATS["verify_negative"]
    nonc, small = bytes.fromhex(neg["non_canonical_point"]), bytes.fromhex(neg["small_order_point"])
    cases.append((bytes.fromhex(neg["golang_pk"]), bytes.fromhex(neg["golang_msg"]), bytes.fromhex(neg["golang_sig"])))
    rnd = random.Random(9)
    for ln in lines:
        p = ln.split(":")
        pub, msg, sig = bytes.fromhex(p[1]), bytes.fromhex(p[2]), bytes.fromhex(p[3])[:64]
        cases.append((pub, msg, sig))
        s_plus_l = ((int.from_bytes(sig[32:], "little") + M.L) % 2**256).to_bytes(32, "little")
        cases.append((pub, msg, sig[:32] + s_plus_l))                       # malleability: s + L
        cases.append((pub, msg, nonc + sig[32:]))                           # non-canonical R
        cases.append((nonc, msg, sig))                                      # non-canonical pk
        cases.append((pub, msg, small + sig[32:]))                          # small-order R
        cases.append((small, msg, sig))                                     # small-order pk
        cases.append((pub, msg + b"x", sig))                                # wrong message
        bad = bytearray(sig); bad[rnd.randrange(64)] ^= 1 << rnd.randrange(8)
        cases.append((pub, msg, bytes(bad)))                                # one flipped bit
        cases.append((bytes.fromhex(KATS["invalid_encodings"][0]), msg, sig))   # pk does not decode
        cases.append((pub, msg, bytes.fromhex(KATS["invalid_encodings"][1]) + sig[32:]))   # R does not decode
        nc1 = (M.P + 1).to_bytes(32, "little")                              # y = p + 1: decodes (to the neutral element) but is not canonical
        cases.append((pub, msg, nc1 + sig[32:]))
        cases.append((nc1, msg, sig))
        cases.append((nonc, msg, nonc + s_plus_l))                          # several failures at once: order matters
        cases.append((small, msg, small + s_plus_l))
    # the edge of the reference's is_canonical (point.rs:315-337 computes 0xED - (1 - b0), not libsodium's 0xED - 1 - b0):
    # with bytes 1..30 = ff, b31 = 7f / ff the answer flips between b0 = 0x13 and 0x14, not between 0xec and 0xed
    pub, msg, sig = cases[1]
    for b0 in (0x00, 0x01, 0x13, 0x14, 0x15, 0x80, 0xeb, 0xec, 0xed, 0xee, 0xff):
        for top in (0x7f, 0xff):
            edge = bytes([b0]) + b"\xff" * 30 + bytes([top])
            cases.append((edge, msg, sig))
            cases.append((pub, msg, edge + sig[32:]))
    return cases


def test_verify_matches_oracle_and_model(hd, oracle):
    base = hd.hd_overflows()
    seen = {0: set(), 1: set()}
    for i, (pub, msg, sig) in enumerate(verify_cases(oracle)):
        for flavor in (0, 1):
            want = oracle.verify(flavor, pub, msg, sig)
            if i % 7 == 0:
                assert M.verify(flavor, pub, msg, sig) == want
            assert hd.hd_verify(flavor, pub, msg, len(msg), sig) == want, (i, flavor, want)
            seen[flavor].add(want)
    assert seen[0] >= {0, 2, 3, 4, 5, 6, 7, 8, 9} and seen[1] >= {0, 2, 3, 4, 5, 6, 7, 8, 9}
    assert hd.hd_overflows() == base


def test_verify_prep_with_the_key_as_a_point(hd, oracle):
    """verify_prep_a_point_with (kyb_verify_points_batch) == verify_prep_a on marshal_binary(point): same flags, same challenge, same group
    element — affine and projective representations, small-order keys, and limbs that are no point of the curve (their bytes decide)"""
    import numpy as np
    base = hd.hd_overflows()
    P = M.P
    rnd = random.Random(3)

    def limbs(v):                                        # integer -> ten reference limbs (radix 2^25.5, non-negative)
        out, bits = [], [26, 25] * 5
        for b in bits:
            out.append(v & ((1 << b) - 1)); v >>= b
        return out

    keys = [oracle.mul_base_ext((rnd.getrandbits(250)).to_bytes(32, "little")) for _ in range(12)]
    keys += [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    cases = []
    for e in keys:
        cases.append(np.array(e, dtype=np.int32))
        X0, Y0, Z0 = [sum(int(e[10 * c + i]) << (0, 26, 51, 77, 102, 128, 153, 179, 204, 230)[i] for i in range(10)) % P for c in (0, 1, 2)]
        x, y = X0 * pow(Z0, P - 2, P) % P, Y0 * pow(Z0, P - 2, P) % P      # (the oracle's own limbs are a projective representation already)
        lam = rnd.randrange(2, P)
        X, Y, Z = x * lam % P, y * lam % P, lam
        T = X * Y * pow(Z, P - 2, P) % P
        cases.append(np.array(limbs(X) + limbs(Y) + limbs(Z) + limbs(T), dtype=np.int32))           # the same point, Z != 1
        cases.append(np.array(limbs(X) + limbs(Y) + limbs(Z) + limbs((T + 1) % P), dtype=np.int32))  # T inconsistent: not a point
    cases.append(np.array(limbs(5) + limbs(7) + limbs(0) + limbs(0), dtype=np.int32))                # Z = 0
    cases.append(np.array([rnd.randrange(-(1 << 24), 1 << 24) for _ in range(40)], dtype=np.int32))  # random limbs
    on = 0
    for ext in cases:
        pub = oracle.encode(ext)
        msg = bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(0, 70)))
        sig = bytes(rnd.getrandbits(8) for _ in range(64))
        same, onc = ctypes.c_int(0), ctypes.c_int(0)
        fl = hd.hd_verify_prep_point(ext.ctypes.data_as(ctypes.c_void_p), pub, msg, len(msg), sig, ctypes.byref(same), ctypes.byref(onc))
        assert fl < 0x100 and same.value == 1, (hex(fl), same.value, onc.value)
        on += onc.value
    assert on == 2 * len(keys)                           # exactly the oracle's own and the re-scaled representations pass the curve test
    assert hd.hd_overflows() == base


def test_pubpoly_eval_and_equal(hd, oracle):
    """short-ladder Horner == the reference's eval (64-window mult by x = i + 1, then add), poly.rs:457-469"""
    base = hd.hd_overflows()
    rnd = random.Random(12)
    for t in (1, 2, 5):
        commits = np.stack([oracle.mul_base_ext(bytes(rnd.getrandbits(8) for _ in range(32))) for _ in range(t)])
        for idx in (0, 1, 2, 6, 255, 256, 65534, 2**32 - 2):
            nbits = (idx + 1).bit_length()
            for nb in {nbits, min(32, nbits + 3)}:
                o = B(32)
                hd.hd_pubpoly_eval(o, p32(commits), t, ctypes.c_uint32(idx), nb)
                assert o.raw == oracle.pubpoly_eval(commits, idx), (t, idx, nb)
    a = oracle.mul_base_ext((5).to_bytes(32, "little"))
    b = oracle.add(oracle.mul_base_ext((2).to_bytes(32, "little")), oracle.mul_base_ext((3).to_bytes(32, "little")))   # same point, other Z
    c = oracle.mul_base_ext((6).to_bytes(32, "little"))
    assert hd.hd_equal(p32(a), p32(b)) == 1 and hd.hd_equal(p32(a), p32(c)) == 0 and hd.hd_equal(p32(a), p32(oracle.neg(a))) == 0
    assert hd.hd_overflows() == base


def test_pubpoly_eval_cut_into_segments(hd, oracle):
    """k_poly_eval_part's flow restated on the host with the device's own field / group / scalar code under the overflow shadow: partial Horner
    chains, multiplier x^(s len) mod 8L as a signed representative, ladder, pairwise sums == the reference's eval (poly.rs:457-469) — also
    when commitments have small-order components, where a multiplier reduced mod L would give another point"""
    base = hd.hd_overflows()
    rnd = random.Random(21)
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    t = 11
    commits = np.stack([oracle.mul_base_ext(bytes(rnd.getrandbits(8) for _ in range(32))) for _ in range(t)])
    commits[2] = oracle.add(commits[2], weak[2])           # mixed order
    commits[5] = weak[3]                                     # small order
    commits[7] = oracle.null()
    commits[10] = oracle.add(commits[10], weak[4])
    for idx in (0, 1, 6, 513, 65534, 2**32 - 2):
        want = oracle.pubpoly_eval(commits, idx)
        nbits = (idx + 1).bit_length()
        for ln in (1, 2, 3, 4, 6, 10, 11):
            o = B(32)
            hd.hd_pubpoly_eval_segments(o, p32(commits), t, ctypes.c_uint32(idx), nbits, ln)
            assert o.raw == want, (idx, ln)
    assert hd.hd_overflows() == base


def test_common_leading_zero_bits_of_a_small_call(hd):
    """the host-side scan behind the short-multiplier path of kyb_mul_batch (scalar_scan.h)"""
    rnd = random.Random(3)
    for _ in range(300):
        n = rnd.randrange(1, 9)
        bits = [rnd.choice([0, 1, 2, 7, 8, 9, 31, 32, 33, 63, 64]) for _ in range(n)]
        vals = [(rnd.getrandbits(b) | (1 << (b - 1))) if b else 0 for b in bits]
        buf = b"".join(v.to_bytes(32, "little") for v in vals)
        got = hd.hd_common_leading_zero_bits(buf, n)
        assert got == 256 - max(bits)
    # a long scalar anywhere ends the short path: the result only has to be below 192
    buf = (5).to_bytes(32, "little") + (1 << 200).to_bytes(32, "little") + (7).to_bytes(32, "little")
    assert hd.hd_common_leading_zero_bits(buf, 3) < 192
    assert hd.hd_common_leading_zero_bits(((1 << 256) - 1).to_bytes(32, "little"), 1) == 0
    # exact mode (what kyb_mul_batch uses: the canonical test must see a long scalar behind a shorter long one)
    buf = (1 << 250).to_bytes(32, "little") + (1 << 255).to_bytes(32, "little")
    assert hd.hd_common_leading_zero_bits_exact(buf, 2) == 0
    assert hd.hd_common_leading_zero_bits_exact((1 << 250).to_bytes(32, "little") * 2, 2) == 5


def test_ladder_path_matches_oracle(hd, oracle):
    """table-free variable-base path (ge_ladder.h) == the reference's windowed multiplication on every
    quirk vector (small-order / mixed-order / non-canonical points; scalars 0, L, 8L, >= 2^255) and on
    random inputs; sc_effective == the integer the reference's recoding multiplies by"""
    base = hd.hd_overflows()
    rnd = random.Random(21)
    cases = [bytes(32), bytes([255] * 32), bytes([0x88] * 32), bytes([0] * 31 + [0x80]), bytes([0xff] * 31 + [0x8f])]
    cases += [bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(500)]
    for s in cases:
        mag = B(32); neg = ctypes.c_int(0)
        hd.hd_effective(mag, ctypes.byref(neg), s)
        want = M.effective_scalar(s)
        assert (-1 if neg.value else 1) * int.from_bytes(mag.raw, "little") == want
    for q in KATS["quirk_mul"]:
        if not q["ok"]:
            continue
        ext, _ = oracle.decode(bytes.fromhex(q["point"]))
        o = B(32); hd.hd_mul_ladder(o, bytes.fromhex(q["scalar"]), p32(ext))
        assert o.raw.hex() == q["out"], q
        o = B(32); hd.hd_mul_ladder_proj(o, bytes.fromhex(q["scalar"]), p32(ext), 0)      # projective-base variant (small-batch kernel)
        assert o.raw.hex() == q["out"], q
        o = B(32); hd.hd_mul_ladder_quad_model(o, bytes.fromhex(q["scalar"]), p32(ext), 0)
        assert o.raw.hex() == q["out"], q
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    for i in range(120):
        s = bytes(rnd.getrandbits(8) for _ in range(32))
        pt = oracle.mul_base_ext(bytes(rnd.getrandbits(8) for _ in range(32)))
        if i % 3 == 0:
            pt = oracle.add(pt, weak[rnd.choice([0, 2, 3, 4])])       # mixed-order point
        if i % 10 == 0:
            s = ((rnd.choice([1, 2, 4, 8]) * M.L + rnd.choice([-1, 0, 1])) % 2**256).to_bytes(32, "little")
        o = B(32); hd.hd_mul_ladder(o, s, p32(pt))
        assert o.raw == oracle.mul(s, pt), (i, s.hex())
        if i % 2 == 0:
            pz = oracle.add(oracle.add(pt, weak[0]), weak[0], sub=True)                     # the same point with Z != 1
            o = B(32); hd.hd_mul_ladder_proj(o, s, p32(pz), 0)
            assert o.raw == oracle.mul(s, pt), (i, s.hex())
            o = B(32); hd.hd_mul_ladder_quad_model(o, s, p32(pz), 0)                         # the four-lane ladder's data flow (ge_ladder_quad.h), lanes as array elements
            assert o.raw == oracle.mul(s, pt), (i, s.hex())
    # verification multiplies by h < L < 2^253: the ladder may start three bits lower
    for v in [0, 1, 2, M.L - 1, M.L - 2, 2**252, 2**252 - 1, 2**252 + 12345] + [rnd.randrange(M.L) for _ in range(40)]:
        s = v.to_bytes(32, "little")
        pt = oracle.add(oracle.mul_base_ext(bytes(rnd.getrandbits(8) for _ in range(32))), weak[rnd.choice([0, 2, 3, 4])])
        o = B(32); hd.hd_mul_ladder_skip(o, s, p32(pt), 3)
        assert o.raw == oracle.mul(s, pt), v
        o = B(32); hd.hd_mul_ladder_proj(o, s, p32(pt), 3)
        assert o.raw == oracle.mul(s, pt), v
        o = B(32); hd.hd_mul_ladder_quad_model(o, s, p32(pt), 3)
        assert o.raw == oracle.mul(s, pt), v
    assert hd.hd_overflows() == base


def test_one_inversion_per_wavefront_model(hd, oracle):
    """k_finish_wave's Montgomery trick across the 64 lanes of a wavefront (butterfly product, one inversion, six kept sub-products multiplied back on)
    as a host model through the overflow-checked build: 64 projective points — sums, the neutral element, small-order points, Z = 0 garbage in three
    lanes — encode as the oracle encodes them one by one; the garbage gets the reference's (0, 0) and disturbs nobody"""
    import numpy as np
    base = hd.hd_overflows()
    rnd = random.Random(66)
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    pts = []
    for l in range(64):
        p = oracle.mul_base_ext(bytes(rnd.getrandbits(8) for _ in range(32)))
        q = oracle.mul_base_ext(bytes(rnd.getrandbits(8) for _ in range(32)))
        pts.append(oracle.add(p, q))                                   # Z != 1
    pts[3], pts[17], pts[40] = oracle.null(), weak[1], oracle.add(pts[40], weak[2])
    want = [oracle.encode(p) for p in pts]
    arr = np.stack(pts).astype(np.int32)
    for bad in (0, 31, 63):
        arr[bad, 20:30] = 0                                            # Z = 0
        want[bad] = bytes(32)
    out = B(64 * 32)
    hd.hd_finish_wave_model(out, arr.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    for l in range(64):
        assert out.raw[32 * l: 32 * l + 32] == want[l], l
    assert hd.hd_overflows() == base


def test_fixed_base_radix32_matches_oracle(hd, oracle):
    """52-window signed radix-32 fixed-base multiplication (sc_effective + sc_recode32) == the
    reference's 64-window routine, quirk scalars included"""
    base = hd.hd_overflows()
    rnd = random.Random(31)
    for q in KATS["quirk_mul_base"]:
        o = B(32); hd.hd_mul_base32(o, bytes.fromhex(q["scalar"]))
        assert o.raw.hex() == q["out"], q["scalar"]
    cases = [bytes([0xff] * 32), bytes([0xf8] * 32), bytes([0x10] * 32), bytes([0x84] * 32), bytes([0x42, 0x08, 0x21, 0x84, 0x10] * 6 + [0x42, 0x08])]
    cases += [bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(200)]
    for s in cases:
        o = B(32); hd.hd_mul_base32(o, s)
        assert o.raw == oracle.mul_base(s), s.hex()
    assert hd.hd_overflows() == base


def test_fixed_base_radix64_matches_oracle(hd, oracle):
    """52 -> 43 windows: the radix-64 odd-digit recoding / table / routine gives the reference's bytes on the quirk
    scalars (>= 2^255, top digits 8..16), on even and odd values (even ones go through k + L), on digit patterns that
    hit the window extremes (c = 0, 31, 32, 63), and at random"""
    base = hd.hd_overflows()
    rnd = random.Random(64)
    for q in KATS["quirk_mul_base"]:
        o = B(32); hd.hd_mul_base64(o, bytes.fromhex(q["scalar"]))
        assert o.raw.hex() == q["out"], q["scalar"]
    L = M.L
    ints = [0, 1, 31, 32, 33, 63, 64, 2**252 - 1, 2**252, 2**252 + 1, 9 * 2**252 - 1, 8 * 2**252 + 5, 2**255 - 19, 2**256 - 1, L, L - 1, 8 * L,
            sum(32 << (6 * i) for i in range(42)), sum(31 << (6 * i) for i in range(43)) % 2**256, sum(33 << (6 * i) for i in range(42)),
            2, 3, 4, 2 * L, 2 * L + 1, 2**253 - 2, 2**253 - 1]
    for c in (0, 31, 32, 63):          # k >> 1 has every 6-bit group equal to c; k odd
        ints.append((2 * sum(c << (6 * i) for i in range(42)) + 1) % 2**256)
        ints.append((2 * sum(c << (6 * i) for i in range(42))) % 2**256)
    cases = [v.to_bytes(32, "little") for v in ints] + [bytes([0xff] * 32), bytes([0xf8] * 32), bytes([0x10] * 32), bytes([0x84] * 32)]
    cases += [bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(200)]
    for s in cases:
        o = B(32); hd.hd_mul_base64(o, s)
        assert o.raw == oracle.mul_base(s), s.hex()
        # ... and in four quarters of the windows added up, as the mid-size kernel forms it (k_mul_base64_quarters)
        o4 = B(32); hd.hd_mul_base64_quarters(o4, s)
        assert o4.raw == o.raw, s.hex()
    assert hd.hd_overflows() == base


def test_eddsa_sign_golden_lines_device_source(hd):
    """EdDSA::sign (key expansion + deterministic nonce + Schnorr equations) straight from (seed, msg)"""
    for ln in gzip.open(os.path.join(HERE, "golden", "sign.input.gz"), "rt").read().split("\n")[:64]:
        p = ln.split(":")
        seed, msg, sig = bytes.fromhex(p[0])[:32], bytes.fromhex(p[2]), bytes.fromhex(p[3])[:64]
        o = B(64); hd.hd_eddsa_sign(o, seed, msg, len(msg))
        assert o.raw == sig


def test_lincomb_flow_matches_oracle(hd, oracle):
    """kyb_lincomb_batch's flow (ladder products + pairwise halving sums) == recover_commit's one-by-one
    accumulation (poly.rs:579-600), incl. equal / opposite / small-order / neutral operands; and Lagrange
    interpolation of public shares at 0 gives back the secret commitment"""
    base = hd.hd_overflows()
    rnd = random.Random(33)
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    for t in (1, 2, 3, 5, 8, 13):
        sc = [bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(t)]
        pts = [oracle.mul_base_ext(bytes(rnd.getrandbits(8) for _ in range(32))) for _ in range(t)]
        if t >= 3:
            pts[1] = pts[0]; sc[1] = sc[0]                         # P + P through the unified addition
            pts[2] = oracle.add(pts[2], weak[3])                    # mixed-order operand
        if t >= 5:
            pts[3] = oracle.neg(pts[4]); sc[3] = sc[4]              # s P + s (-P) = neutral element
        if t >= 8:
            pts[5] = oracle.null(); pts[6] = weak[2]; sc[7] = bytes(32)
        scb = b"".join(sc); ptb = np.stack(pts)
        o = B(32); hd.hd_lincomb(o, scb, p32(ptb), t)
        assert o.raw == oracle.lincomb(np.frombuffer(scb, dtype=np.uint8), ptb), t
    # recover_commit: shares y_i = eval(i) of a threshold-4 public polynomial at indices {1, 3, 4, 6}
    t = 4
    coeffs = [rnd.randrange(M.L) for _ in range(t)]
    xs = [i + 1 for i in (1, 3, 4, 6)]
    ys = np.stack([oracle.mul_base_ext((sum(c * pow(x, j, M.L) for j, c in enumerate(coeffs)) % M.L).to_bytes(32, "little")) for x in xs])
    lam = []
    for xi in xs:
        num = den = 1
        for xj in xs:
            if xj != xi:
                num = num * xj % M.L
                den = den * (xj - xi) % M.L
        lam.append(num * pow(den, M.L - 2, M.L) % M.L)
    o = B(32); hd.hd_lincomb(o, b"".join(v.to_bytes(32, "little") for v in lam), p32(ys), t)
    assert o.raw == oracle.mul_base(coeffs[0].to_bytes(32, "little"))
    assert hd.hd_overflows() == base
