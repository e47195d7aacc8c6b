#!/usr/bin/env python3
"""Regenerates tests/golden/* from the reference tree (run in the build container only; the GPU box
has no /root/reference and uses the committed files).

What is copied is DATA the reference's own tests hold for this path, never source:
  * sign.input.gz     — src/sign/eddsa/testdata/sign.input.gz, byte for byte (the 1024-line golden
                        file checked by tests/sign/eddsa.rs:36-94; public-domain ed25519 test vectors)
  * kats.json         — hex strings lifted from test code:
        rfc8032        src/sign/eddsa/eddsa_test.rs:19-46   (RFC 8032 §7.1 vectors)
        scalar_kats    src/group/edwards25519/scalar_test.rs:27-75
        decode_kat     src/group/edwards25519/ge.rs:65-73
        weak_keys      src/group/edwards25519/constants.rs:3744-3775
    plus quirk vectors whose expected outputs come from the C oracle and are cross-checked against
    oracle/bigint_model.py before being written (no reference test covers them).
        embed          Point::embed / Point::pick (point.rs:90-92, 106-167) over replayed key streams: accepted candidate, blocks
                       consumed and the reason every earlier candidate was rejected (C oracle == big-int model, as above)
"""
import hashlib
import json
import os
import re
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import bigint_model as M
    import oracle_lib

    orc = oracle_lib.Oracle()
    shutil.copyfile(os.path.join(REF, "src/sign/eddsa/testdata/sign.input.gz"), os.path.join(HERE, "sign.input.gz"))

    out = {}
    # RFC 8032 vectors
    txt = open(os.path.join(REF, "src/sign/eddsa/eddsa_test.rs")).read()
    vecs = re.findall(r'private:\s*"([0-9a-f]*)",\s*public:\s*"([0-9a-f]*)",\s*message:\s*"([0-9a-f]*)",\s*signature:\s*"([0-9a-f]*)"', txt)
    assert len(vecs) == 5
    out["rfc8032"] = [dict(private=a, public=b, message=c, signature=d) for a, b, c, d in vecs]
    # scalar KATs
    out["scalar_kats"] = {
        "int64_0x100_plus_1": "0101000000000000000000000000000000000000000000000000000000000000",
        "int64_minus_1": "ecd3f55c1a631258d69cf7a2def9de1400000000000000000000000000000010",
        "int64_1": "0100000000000000000000000000000000000000000000000000000000000000",
        "set_bytes_00010203": "0001020300000000000000000000000000000000000000000000000000000000",
    }
    st = open(os.path.join(REF, "src/group/edwards25519/scalar_test.rs")).read()
    for v in out["scalar_kats"].values():
        assert v in st, v
    # decode KAT
    gt = open(os.path.join(REF, "src/group/edwards25519/ge.rs")).read()
    m = re.search(r"let arr: \[u8; 32\] = \[([^\]]*)\]", gt)
    out["decode_kat"] = bytes(int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()).hex()
    # weak keys
    ct = open(os.path.join(REF, "src/group/edwards25519/constants.rs")).read()
    wk = ct[ct.index("pub const WEAK_KEYS"):]
    rows = re.findall(r"\[\s*((?:0x[0-9a-f]{2},\s*){32})\]", wk)
    assert len(rows) == 5
    out["weak_keys"] = [bytes(int(x, 16) for x in re.findall(r"0x[0-9a-f]{2}", r)).hex() for r in rows]

    # quirk vectors: scalar x point -> encoding, oracle output cross-checked with the big-int model
    L, P = M.L, M.P
    sc = [0, 1, 2, 8, L - 1, L, L + 1, 8 * L, 2**252, 2**255 - 1, 2**255, 2**255 + 1, 2**256 - 1]
    sc_bytes = [v.to_bytes(32, "little") for v in sc]
    for top in (0x7F, 0x80, 0x8F, 0x90, 0xFF):
        sc_bytes.append(bytes([0x11] * 31 + [top]))
        sc_bytes.append(bytes([0xEE] * 31 + [top]))
    pts = [M.encode(M.IDENT), M.encode(M.B)] + [bytes.fromhex(h) for h in out["weak_keys"]]
    pts += [(P + k).to_bytes(32, "little") for k in (0, 1, 3, 18)]               # non-canonical y
    pts += [bytes([1] + [0] * 30 + [0x80])]                                     # x = 0 with the sign bit
    pts += [M.encode(M.point_mul((12345).to_bytes(32, "little")))]
    q = []
    for pe in pts:
        ext, ok = orc.decode(pe)
        pm = M.decode(pe)
        assert (pm is not None) == bool(ok), pe.hex()
        if not ok:
            q.append(dict(point=pe.hex(), ok=0))
            continue
        assert orc.encode(ext) == M.encode(pm)
        for s in sc_bytes:
            o = orc.mul(s, ext)
            assert o == M.encode(M.point_mul(s, pm)), (s.hex(), pe.hex())
            q.append(dict(point=pe.hex(), ok=1, scalar=s.hex(), out=o.hex()))
    out["quirk_mul"] = q
    qb = []
    for s in sc_bytes:
        o = orc.mul_base(s)
        assert o == M.encode(M.point_mul(s))
        qb.append(dict(scalar=s.hex(), out=o.hex()))
    out["quirk_mul_base"] = qb
    # invalid encodings (no square root): first few y values that do not decode
    bad = []
    y = 2
    while len(bad) < 4:
        b = y.to_bytes(32, "little")
        if M.decode(b) is None:
            assert orc.decode(b)[1] == 0
            bad.append(b.hex())
        y += 1
    out["invalid_encodings"] = bad
    # negative verification vectors (eddsa_test.rs:142-163 golang malleability vector; :169-173 and :199-203
    # non-canonical R / pk; :223-227 and :251-255 small-order R / pk)
    blocks = re.findall(r"\[u8; (?:32|64)\] = \[(.*?)\];", txt, flags=re.S)
    arrs = [bytes(int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{2}", b)) for b in blocks]
    sig2 = [a for a in arrs if len(a) == 64][0]
    pk2 = [a for a in arrs if len(a) == 32 and a[0] == 0x7d][0]
    noncanon = [a for a in arrs if len(a) == 32 and a[0] == 0xef][0]
    small = [a for a in arrs if len(a) == 32 and a[0] == 0xc7][0]
    assert small.hex() == out["weak_keys"][3]
    out["verify_negative"] = dict(golang_msg="54657374", golang_sig=sig2.hex(), golang_pk=pk2.hex(),
                                  non_canonical_point=noncanon.hex(), small_order_point=small.hex())
    assert orc.verify(0, pk2, bytes.fromhex("54657374"), sig2) == 2 == M.verify(0, pk2, bytes.fromhex("54657374"), sig2)
    out["embed"] = embed_vectors(orc, M, out)
    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(q), "quirk_mul,", len(qb), "quirk_mul_base vectors")


def key_stream(tag: bytes, blocks: int) -> bytes:
    """a deterministic stand-in for `rand.xor_key_stream(&mut b, &[0; 32])`: block i = SHA-512("kyber-hip/v1/embed" || tag || le64(i))[:32]"""
    return b"".join(hashlib.sha512(b"kyber-hip/v1/embed" + tag + i.to_bytes(8, "little")).digest()[:32] for i in range(blocks))


def embed_vectors(orc, M, out):
    """Known answers for A13.  Every vector: data (hex, or null for pick), the key stream up to one block past the accepted candidate,
    the accepted point's encoding, the blocks consumed, the bytes Point::data() returns, and `rejected` = why each earlier candidate
    failed ("decode": no square root; "order": decodes, but 8 P = O for pick / L P != O for embed)."""
    def why(data, stream, n):
        r = []
        dl = 0 if data is None else min(29, len(data))
        for i in range(n - 1):
            b = bytearray(stream[32 * i:32 * i + 32])
            if data is not None:
                b[0] = dl
                b[1:1 + dl] = data[:dl]
            r.append("decode" if M.decode(bytes(b)) is None else "order")
        return r

    vecs = []

    def add(data, stream, note):
        e, enc, n = orc.embed(data, stream)
        pm, nm = M.embed(data, stream)
        assert n == nm and n > 0 and enc == M.encode(pm), note
        d_out = None if data is None else orc.point_data(e)
        if data is not None:
            assert d_out == data[:29], note
        # the accepted candidate must not depend on what follows it; keep one block more to show that it is not drawn
        keep = stream[:32 * min(n + 1, len(stream) // 32)]
        assert orc.embed(data, keep)[1] == enc
        vecs.append(dict(note=note, data=None if data is None else data.hex(), stream=keep.hex(), out=enc.hex(), consumed=n,
                         point_data=None if d_out is None else d_out.hex(), rejected=why(data, stream, n)))

    # natural streams: pick accepts about every second candidate, embed about one in sixteen (decodes AND lies in the prime-order
    # subgroup); tags are searched so that the committed streams stay short but both kinds of rejection occur
    datas = [None, b"", b"H", b"Hi!", bytes(range(1, 30)), b"The quick brown fox jumps over the lazy dog", bytes([0xff] * 29), bytes(40)]
    for data in datas:
        got = 0
        k = 0
        while got < 3:
            tag = (b"-" if data is None else data[:4]) + b"/" + str(k).encode()
            k += 1
            st = key_stream(tag, 80)
            n = M.embed(data, st)[1]
            lim = 6 if data is None else 24
            if 0 < n <= lim and (got != 2 or n >= 2):
                add(data, st, "natural stream " + tag.decode("latin1"))
                got += 1
    # crafted pick streams: encodings that do not decode, then every small-order encoding (8 P = O: "unlucky; try again"), then B
    bad = [bytes.fromhex(h) for h in out["invalid_encodings"]]
    weak = [bytes.fromhex(h) for h in out["weak_keys"]]
    ident = bytes([1] + [0] * 31)
    add(None, b"".join(bad) + M.encode(M.B) + ident, "pick: four undecodable blocks, then B")
    add(None, b"".join(weak) + ident + M.encode(M.B) + ident, "pick: the five WEAK_KEYS and the neutral element are skipped, then B")
    add(None, weak[2] + bad[0] + bytes([weak[3][0]]) + weak[3][1:31] + bytes([weak[3][31] | 0x80]) + M.encode(M.point_mul((7).to_bytes(32, "little"))) + ident,
        "pick: small order, undecodable, small order with the sign bit, then 7B")
    # a mixed-order candidate (B + a point of order 8): pick returns 8 (B + T) = 8 B; embed's order test must reject such a point
    t8 = M.decode(weak[2])
    mixed = M.encode(M.add(M.B, t8))
    add(None, mixed + ident, "pick: mixed-order candidate, the result is 8 times it")
    # crafted embed streams: the data bytes overwrite the front of the block, so take an accepted natural candidate and prepend failures
    for data in (b"", b"Hi!"):
        st = key_stream(b"craft/" + data, 400)
        pm, n = M.embed(data, st)
        good = st[32 * (n - 1):32 * n]
        # a block whose candidate decodes to a point outside the subgroup: search the stream for one
        dl = len(data)
        order_fail = None
        for i in range(n - 1):
            b = bytearray(st[32 * i:32 * i + 32]); b[0] = dl; b[1:1 + dl] = data
            if M.decode(bytes(b)) is not None:
                order_fail = st[32 * i:32 * i + 32]
                break
        assert order_fail is not None
        add(data, order_fail + order_fail + good + ident, "embed %r: two candidates outside the prime-order subgroup, then an accepted one" % data)
    return vecs


if __name__ == "__main__":
    main()
