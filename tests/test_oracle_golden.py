"""Pins the CPU oracle (oracle/ed25519_oracle.c) against every known-answer vector the reference's
own tests hold for the hot path and against the independent big-int model.
CPU only."""
import gzip
import hashlib
import json
import os
import random
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import bigint_model as M  # noqa: E402

GOLD = os.path.join(HERE, "golden")
KATS = json.load(open(os.path.join(GOLD, "kats.json")))


def golden_lines():
    for ln in gzip.open(os.path.join(GOLD, "sign.input.gz"), "rt").read().split("\n"):
        if ln:
            p = ln.split(":")
            yield bytes.fromhex(p[0])[:32], bytes.fromhex(p[1]), bytes.fromhex(p[2]), bytes.fromhex(p[3])[:64]


def test_golden_sign_input_all_1024_lines(oracle):
    """tests/sign/eddsa.rs:36-94: public key and signature of every line (2 fixed-base mults, 2 encodes,
    SHA-512 -> mod L, s = r + a*h mod L per line)."""
    n = 0
    for seed, pub, msg, sig in golden_lines():
        secret, prefix, p = oracle.eddsa_expand(seed)
        assert p == pub
        assert oracle.eddsa_sign(seed, msg) == sig
        n += 1
    assert n == 1024


def test_rfc8032_vectors(oracle):
    """src/sign/eddsa/eddsa_test.rs:19-107"""
    for v in KATS["rfc8032"]:
        seed, msg = bytes.fromhex(v["private"]), bytes.fromhex(v["message"])
        assert oracle.eddsa_expand(seed)[2].hex() == v["public"]
        assert oracle.eddsa_sign(seed, msg).hex() == v["signature"]
        assert M.eddsa_sign(seed, msg).hex() == v["signature"]


def test_golden_verify_equation_pins_variable_base(oracle):
    """eddsa_sig.rs:194-211: s*B == R + h*A on golden signatures — the only place the reference's
    vectors reach the variable-base routine (h*A).  All 1024 lines."""
    for i, (seed, pub, msg, sig) in enumerate(golden_lines()):
        a_ext, ok = oracle.decode(pub)
        r_ext, ok2 = oracle.decode(sig[:32])
        assert ok and ok2
        h = oracle.sc_reduce64(hashlib.sha512(sig[:32] + pub + msg).digest())
        ha = oracle.mul_ext(h, a_ext)
        rha = oracle.add(r_ext, ha)
        assert oracle.encode(rha) == oracle.mul_base(sig[32:])


def test_scalar_kats(oracle):
    """scalar_test.rs:27-75 (set_int64 / set_bytes / Display)"""
    k = KATS["scalar_kats"]
    L = M.L
    one = (1).to_bytes(32, "little")
    x100 = (0x100).to_bytes(32, "little")
    zero = bytes(32)
    assert oracle.sc_muladd(one, x100, one).hex() == k["int64_0x100_plus_1"]          # 0x100 + 1
    assert oracle.sc_reduce32(((L - 1) % 2**256).to_bytes(32, "little")).hex() == k["int64_minus_1"]
    assert oracle.sc_muladd(zero, zero, one).hex() == k["int64_1"]
    assert oracle.sc_reduce64(bytes([0, 1, 2, 3]) + bytes(60)).hex() == k["set_bytes_00010203"]


def test_scalar_arithmetic_is_canonical_mod_l(oracle):
    """sc_mul_add / set_bytes restated with ref10 limbs must equal plain integer arithmetic mod L for
    ANY 256/512-bit inputs (scalar.rs:279-744; integer.rs:386-397)."""
    rnd = random.Random(7)
    edge = [bytes(32), bytes([255] * 32), (M.L).to_bytes(32, "little"), (M.L - 1).to_bytes(32, "little"), (2**255).to_bytes(32, "little")]
    cases = [(a, b, c) for a in edge for b in edge for c in edge]
    cases += [tuple(bytes(rnd.getrandbits(8) for _ in range(32)) for _ in range(3)) for _ in range(3000)]
    for a, b, c in cases:
        want = (int.from_bytes(a, "little") * int.from_bytes(b, "little") + int.from_bytes(c, "little")) % M.L
        assert int.from_bytes(oracle.sc_muladd(a, b, c), "little") == want
    for x in [bytes(64), bytes([255] * 64)] + [bytes(rnd.getrandbits(8) for _ in range(64)) for _ in range(3000)]:
        assert int.from_bytes(oracle.sc_reduce64(x), "little") == int.from_bytes(x, "little") % M.L


def test_decode_kats(oracle):
    """ge.rs:65-73 and point_test.rs:19-25 (the five WEAK_KEYS decode)"""
    assert oracle.decode(bytes.fromhex(KATS["decode_kat"]))[1] == 1
    for h in KATS["weak_keys"]:
        ext, ok = oracle.decode(bytes.fromhex(h))
        assert ok == 1
        # small order: 8 * P == neutral element
        assert oracle.mul((8).to_bytes(32, "little"), ext) == M.encode(M.IDENT)
    for h in KATS["invalid_encodings"]:
        assert oracle.decode(bytes.fromhex(h))[1] == 0


def test_quirk_vectors_match_fixture(oracle):
    """scalars >= 2^255, L, 8L, non-canonical encodings, x = 0 with sign bit, small-order points"""
    for q in KATS["quirk_mul"]:
        ext, ok = oracle.decode(bytes.fromhex(q["point"]))
        assert ok == q["ok"]
        if ok:
            assert oracle.mul(bytes.fromhex(q["scalar"]), ext).hex() == q["out"]
    for q in KATS["quirk_mul_base"]:
        assert oracle.mul_base(bytes.fromhex(q["scalar"])).hex() == q["out"]


def test_oracle_equals_bigint_model_on_random_inputs(oracle):
    rnd = random.Random(11)
    for _ in range(24):
        s = bytes(rnd.getrandbits(8) for _ in range(32))
        ps = bytes(rnd.getrandbits(8) for _ in range(32))
        assert oracle.mul_base(s) == M.encode(M.point_mul(s))
        pt = oracle.mul_base_ext(ps)                     # random point, arbitrary Z
        affine = M.point_mul(ps)
        assert oracle.encode(pt) == M.encode(affine)
        assert oracle.mul(s, pt) == M.encode(M.point_mul(s, affine))
        x, k, m = s, ps, bytes(rnd.getrandbits(8) for _ in range(rnd.randrange(0, 200)))
        assert oracle.schnorr_sign(x, k, m) == M.schnorr_sign(x, k, m)


def test_group_identities(oracle):
    """util/test/group_test.rs:243-435 restated on encodings: B+B == 2B, (-1)B + B == 0, DH
    commutativity, 0*P == 0, 1*P == P, mul(s, Some(B)) == mul(s, None)."""
    rnd = random.Random(5)
    Bext = oracle.base()
    two = (2).to_bytes(32, "little")
    assert oracle.encode(oracle.add(Bext, Bext)) == oracle.mul_base(two)
    m1 = (M.L - 1).to_bytes(32, "little")
    assert oracle.encode(oracle.add(oracle.mul_base_ext(m1), Bext)) == M.encode(M.IDENT)
    assert oracle.encode(oracle.null()) == M.encode(M.IDENT)
    for _ in range(8):
        s1 = M.sc_bytes(rnd.getrandbits(300))
        s2 = M.sc_bytes(rnd.getrandbits(300))
        p1, p2 = oracle.mul_base_ext(s1), oracle.mul_base_ext(s2)
        assert oracle.mul(s2, p1) == oracle.mul(s1, p2)
        assert oracle.mul(s1, Bext) == oracle.mul_base(s1)
        assert oracle.mul(bytes(32), p1) == M.encode(M.IDENT)
        assert oracle.mul((1).to_bytes(32, "little"), p1) == oracle.encode(p1)
        assert oracle.encode(oracle.add(p1, p2, sub=True)) == oracle.encode(oracle.add(p1, oracle.neg(p2)))


def test_sha512(oracle):
    rnd = random.Random(3)
    for n in [0, 1, 55, 111, 112, 113, 127, 128, 129, 239, 240, 241, 1000]:
        m = bytes(rnd.getrandbits(8) for _ in range(n))
        assert oracle.sha512(m) == hashlib.sha512(m).digest()


def test_verify_golden_and_negative_vectors(oracle):
    """eddsa_test.rs:51-107 (verify after sign), :110-272 (malleability, non-canonical R / pk, small-order R / pk)"""
    neg = KATS["verify_negative"]
    msg2, sig2, pk2 = (bytes.fromhex(neg[k]) for k in ("golang_msg", "golang_sig", "golang_pk"))
    nonc, small = bytes.fromhex(neg["non_canonical_point"]), bytes.fromhex(neg["small_order_point"])
    assert oracle.verify(0, pk2, msg2, sig2) == 2                      # "signature is not canonical"
    assert oracle.weak_keys() == [bytes.fromhex(h) for h in KATS["weak_keys"]]   # regenerated from the group law
    for i, (seed, pub, msg, sig) in enumerate(golden_lines()):
        if i % 8:
            continue
        for flavor in (0, 1):
            assert oracle.verify(flavor, pub, msg, sig) == 0
        s_plus_l = ((int.from_bytes(sig[32:], "little") + M.L) % 2**256).to_bytes(32, "little")
        assert oracle.verify(0, pub, msg, sig[:32] + s_plus_l) == 2    # eddsa_test.rs:127-137
        assert oracle.verify(0, pub, msg, nonc + sig[32:]) == 3        # "R is not canonical"
        assert oracle.verify(0, nonc, msg, sig) == 6                   # "public key is not canonical"
        assert oracle.verify(0, pub, msg, small + sig[32:]) == 5       # "R has small order"
        assert oracle.verify(0, small, msg, sig) == 8                  # "public key has small order"
        assert oracle.verify(0, pub, msg + b"!", sig) == 9
        assert oracle.verify(0, pub, msg, sig[:63]) == 1
        if i % 64 == 0:
            for flavor in (0, 1):
                for p_, m_, s_ in ((pub, msg, sig), (nonc, msg, nonc + s_plus_l), (small, msg, small + sig[32:]), (pub, msg + b"!", sig)):
                    assert M.verify(flavor, p_, m_, s_) == oracle.verify(flavor, p_, m_, s_)


def _ref_is_canonical(b: bytes) -> bool:
    """point.rs:315-337 transliterated operation by operation (u8 / u16 wrapping, Cargo.toml:9-10 overflow-checks = false)"""
    if len(b) != 32:
        return False
    c = (b[31] & 0x7f) ^ 0x7f
    for i in range(30, 0, -1):
        c |= b[i] ^ 0xff
    c = (((c - 1) & 0xffff) >> 8) & 0xff
    d = (((0xED - ((1 - b[0]) & 0xffff)) & 0xffff) >> 8) & 0xff
    return 1 - (c & d & 1) == 1


def test_point_is_canonical_follows_the_reference_expression(oracle):
    """The reference's is_canonical deviates from the libsodium routine it cites (0xED - (1 - b0) instead of 0xED - 1 - b0):
    oracle and big-int model must answer what the reference answers, for every low byte, through verify's status codes."""
    assert [_ref_is_canonical(bytes([b0]) + b"\xff" * 30 + b"\x7f") for b0 in (0x13, 0x14, 0xec, 0xed)] == [True, False, False, False]
    lines = list(golden_lines())
    _, pub, msg, sig = lines[0]
    for b0 in range(256):
        for top in (0x7f, 0xff):
            for mid in (0xff, 0xfe):
                enc = bytes([b0]) + bytes([mid]) + b"\xff" * 29 + bytes([top])
                can = _ref_is_canonical(enc)
                assert M.point_is_canonical(enc) == can, enc.hex()
                # flavor 0 checks the canonical form of the public key before anything else about it (eddsa_sig.rs:176-190)
                st = oracle.verify(0, enc, msg, sig)
                assert (st == 6) == (not can), (enc.hex(), st)
                st_r = oracle.verify(0, pub, msg, enc + sig[32:])
                assert (st_r == 3) == (not can), (enc.hex(), st_r)
                if b0 % 16 == 4:
                    for flavor in (0, 1):
                        assert M.verify(flavor, enc, msg, sig) == oracle.verify(flavor, enc, msg, sig)
                        assert M.verify(flavor, pub, msg, enc + sig[32:]) == oracle.verify(flavor, pub, msg, enc + sig[32:])


def _point_check_inputs():
    """received encodings for the stand-alone checks: the WEAK_KEYS with and without the sign bit, their non-canonical aliases, y = p-220 .. p+18
    (the reference's is_canonical turns at p-217), bytes that decode to no point, ordinary points"""
    P = M.P
    encs = []
    for h in KATS["weak_keys"]:
        w = bytes.fromhex(h)
        encs += [w, w[:31] + bytes([w[31] | 0x80])]
        y = int.from_bytes(w, "little") & ((1 << 255) - 1)
        if y + P < (1 << 255):
            encs += [(y + P).to_bytes(32, "little"), (y + P + (1 << 255)).to_bytes(32, "little")]     # y = 0 -> p, y = 1 -> p + 1: small order, not canonical
    for y in range(P - 220, P + 19):
        encs += [y.to_bytes(32, "little"), (y | (1 << 255)).to_bytes(32, "little")]
    encs += [bytes.fromhex(h) for h in KATS["invalid_encodings"]]
    rng = random.Random(5)
    for _ in range(64):
        encs.append(M.encode(M.point_mul(rng.randrange(1, M.L).to_bytes(32, "little"))))
        encs.append(bytes(rng.randrange(256) for _ in range(32)))
    return encs


def test_point_checks_oracle_matches_the_model():
    """orc_point_checks (what tests/test_gpu_parity.py holds kyb_point_checks_batch against) == the big-int model's point_is_canonical /
    has_small_order / decode on every input class; and the property the engine's byte-only kernel relies on: the small-order bit depends on
    y mod p alone, and every weak y decodes"""
    import oracle_lib
    orc = oracle_lib.Oracle()
    weak_y = set()
    for enc in _point_check_inputs():
        f = orc.point_checks(enc)
        pt = M.decode(enc)
        assert bool(f & 4) == (pt is not None), enc.hex()
        assert bool(f & 1) == M.point_is_canonical(enc) == _ref_is_canonical(enc), enc.hex()
        assert bool(f & 2) == (pt is not None and M.has_small_order(pt)), enc.hex()
        if f & 2:
            weak_y.add(pt[1])
        if pt is not None:
            e, ok = orc.decode(enc)
            assert ok and orc.point_checks_ext(e) == (f & 2) | int(M.point_is_canonical(orc.encode(e)))
    assert len(weak_y) == 5
    for y in weak_y:                                           # a byte string whose y mod p is weak always decodes
        for alias in (y, y + M.P):
            if alias < (1 << 255):
                assert M.decode(alias.to_bytes(32, "little")) is not None


def test_embed_and_pick_known_answers(oracle):
    """A13, point.rs:90-92 / 106-167 over replayed key streams: the accepted candidate (bytes), how many blocks the loop drew and
    Point::data of the result — C oracle and big-int model against the committed vectors; the rejected candidates fail for the
    recorded reason (no square root / 8 P = O for pick / L P != O for embed)."""
    assert len(KATS["embed"]) >= 30
    seen = set()
    for v in KATS["embed"]:
        data = None if v["data"] is None else bytes.fromhex(v["data"])
        stream = bytes.fromhex(v["stream"])
        ext, enc, n = oracle.embed(data, stream)
        assert (enc.hex(), n) == (v["out"], v["consumed"]), v["note"]
        pm, nm = M.embed(data, stream)
        assert (M.encode(pm).hex(), nm) == (v["out"], v["consumed"]), v["note"]
        if data is None:
            assert v["point_data"] is None
            assert M.mul_int(M.L, pm) == M.IDENT and pm != M.IDENT              # pick lands in the prime-order subgroup
        else:
            assert oracle.point_data(ext).hex() == v["point_data"] == data[:29].hex()
            assert enc[0] == min(29, len(data)) and enc[1:1 + enc[0]] == data[:29]
        # a stream that ends before the accepted block is exhausted without a result
        assert oracle.embed(data, stream[:32 * (n - 1)])[2] == -1
        dl = 0 if data is None else min(29, len(data))
        for i, reason in enumerate(v["rejected"]):
            b = bytearray(stream[32 * i:32 * i + 32])
            if data is not None:
                b[0] = dl
                b[1:1 + dl] = data[:dl]
            e, ok = oracle.decode(bytes(b))
            assert bool(ok) == (reason == "order"), (v["note"], i)
            if ok:
                q = oracle.mul((8).to_bytes(32, "little") if data is None else M.L.to_bytes(32, "little"), e)
                assert (q == M.encode(M.IDENT)) == (data is None), (v["note"], i)
            seen.add((data is None, reason))
    assert seen == {(True, "decode"), (True, "order"), (False, "decode"), (False, "order")}
    # Point::data on a point whose length byte exceeds embed_len: PointError::EmbedDataLength (point.rs:172-175)
    e, ok = oracle.decode(M.encode(M.B))
    assert ok and oracle.point_data(e) is None                                     # B's encoding starts with 0x58


def test_pripoly_eval_matches_python_integers(oracle):
    """orc_pripoly_eval (PriPoly::eval, poly.rs:133-141) == sum c_j x^j mod L in Python integers: canonical and unreduced coefficients, t = 1, the
    largest index the ABI takes"""
    import numpy as np
    import synth
    L = synth.L
    for t, seed in ((1, 1), (2, 2), (9, 3), (64, 4)):
        coeffs = np.concatenate([synth.scalars(t - t // 2, 900 + seed), synth.raw256(t // 2, 900 + seed)])
        ints = [int.from_bytes(bytes(c), "little") for c in coeffs]
        for index in (0, 1, 2, 511, 65535, 0xfffffffe):
            want = sum(c * pow(index + 1, j, L) for j, c in enumerate(ints)) % L
            assert int.from_bytes(oracle.pripoly_eval(coeffs, index), "little") == want, (t, index)
