"""What a transcript of tests/cpp/test_dss_round.cpp must satisfy, checked with the oracle and Python integers only: the participants' keys and both
distributed keys' commitments are the fixed-base multiples of the printed scalars, hash_sig is SHA-512(R || A || msg) mod L (dss_sig.rs:312-326), the
partial signature is hash * alpha_i + beta_i (:215-237) and carries a valid Schnorr signature, the public shares are the two public polynomials at
every index (poly.rs:457-469), every honest partial signature is accepted and the two bad ones refused, and the distributed signature is
R || (r + hash * a), which the oracle accepts as a plain EdDSA signature under the distributed key (dss_sig.rs:293-310, 330-332)."""
import hashlib

import numpy as np

import synth

L = synth.L


def by_tag(lines):
    by = {}
    for ln in lines:
        tag, val = ln.split()
        by.setdefault(tag, []).append(val)
    return by


def _poly_at(coeffs, i):
    x, v = i + 1, 0
    for c in reversed(coeffs):
        v = (v * x + c) % L
    return v


def check_transcript(lines, n, t, oracle):
    by = by_tag(lines)
    b = bytes.fromhex
    le = lambda h: int.from_bytes(b(h), "little")
    me = n // 2
    assert len(by["PRIV"]) == n and len(by["LCOEFF"]) == t and len(by["RCOEFF"]) == t
    assert by["INDEX"] == [str(me)]
    assert by["LCOMMIT"] == [oracle.mul_base(b(c)).hex() for c in by["LCOEFF"]]
    assert by["RCOMMIT"] == [oracle.mul_base(b(c)).hex() for c in by["RCOEFF"]]
    msg = b(by["MSG"][0])
    lc, rc = [le(c) for c in by["LCOEFF"]], [le(c) for c in by["RCOEFF"]]
    h = int.from_bytes(hashlib.sha512(b(by["RCOMMIT"][0]) + b(by["LCOMMIT"][0]) + msg).digest(), "little") % L
    assert le(by["HASHSIG"][0]) == h
    assert le(by["PARTIAL"][0]) == (h * _poly_at(lc, me) + _poly_at(rc, me)) % L
    pub_me = oracle.mul_base(b(by["PRIV"][me]))
    assert oracle.verify(1, pub_me, b(by["PSMSG"][0]), b(by["PSSIG"][0])) == 0            # schnorr::verify_with_checks of the own partial signature
    others = [i for i in range(n) if i != me]
    lc_ext = np.stack([oracle.mul_base_ext(b(c)) for c in by["LCOEFF"]])
    rc_ext = np.stack([oracle.mul_base_ext(b(c)) for c in by["RCOEFF"]])
    assert by["PSOK"] == ["1"] * (n - 1)
    assert by["RANDSHARE"] == [oracle.pubpoly_eval(rc_ext, i).hex() for i in others]
    assert by["LONGSHARE"] == [oracle.pubpoly_eval(lc_ext, i).hex() for i in others]
    # the share check itself, once more from the scalars: partial_i * B == rand_share_i + hash * long_share_i
    for i in others[:3]:
        partial = (h * _poly_at(lc, i) + _poly_at(rc, i)) % L
        right = oracle.add(oracle.decode(oracle.pubpoly_eval(rc_ext, i))[0], oracle.mul_ext(h.to_bytes(32, "little"), oracle.decode(oracle.pubpoly_eval(lc_ext, i))[0]))
        assert oracle.encode(right) == oracle.mul_base(partial.to_bytes(32, "little"))
    assert by["BADSHARE_SIGOK"] == ["1"] and by["BADSHARE_OK"] == ["0"] and by["BADSIG_STATUS"] == ["9"]
    gamma = (rc[0] + h * lc[0]) % L
    sig = b(by["SIGNATURE"][0])
    assert sig == b(by["RCOMMIT"][0]) + gamma.to_bytes(32, "little")
    assert oracle.verify(0, b(by["LCOMMIT"][0]), msg, sig) == 0 and by["FINALOK"] == ["1"]     # eddsa::verify_with_checks under the distributed key
