"""Small-batch (one item per wavefront, lane-cooperative) kernels: bit-identical to the oracle and to the batch kernels
on the quirk vectors (scalars >= 2^255, L, 8L, 0, 1; identity, small-order, mixed-order points), random inputs, ragged
sizes; selected by batch size through `coop.max_items` / `coop.base_max_items`."""
import json
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "kats.json")))


@pytest.fixture()
def coop_engine(engine):
    keys = ("coop.max_items", "coop.base_max_items", "coop.ladder_max_items", "coop.ladder_enc_max_items")
    old = [engine.get_option(k) for k in keys]
    for k in keys:
        engine.set_option(k, 1 << 20)
    yield engine
    for k, v in zip(keys, old):
        engine.set_option(k, v)


@pytest.fixture()
def coop_xengine(xengine):
    """the same on the cross-check build, for the tests that also switch kernel variants"""
    keys = ("coop.max_items", "coop.base_max_items", "coop.ladder_max_items", "coop.ladder_enc_max_items")
    old = [xengine.get_option(k) for k in keys]
    for k in keys:
        xengine.set_option(k, 1 << 20)
    yield xengine
    for k, v in zip(keys, old):
        xengine.set_option(k, v)


def test_coop_fixed_base_matches_oracle(coop_engine, oracle):
    eng = coop_engine
    q = KATS["quirk_mul_base"]
    sc = np.frombuffer(b"".join(bytes.fromhex(v["scalar"]) for v in q), dtype=np.uint8).reshape(-1, 32)
    got = eng.mul_base(sc)
    assert [bytes(r).hex() for r in got] == [v["out"] for v in q]
    for n in (1, 2, 63, 64, 65, 300):
        s = synth.raw256(n, 400 + n)
        enc, ext = eng.mul_base(s, want_ext=True)
        assert np.array_equal(enc, oracle.mul_base_batch(s, nthreads=8))
        assert [oracle.encode(e) for e in ext[:8]] == [bytes(r) for r in enc[:8]]
        assert all(list(e[20:30]) == [1] + [0] * 9 for e in ext[:8])
    assert eng.mul_base(sc[:0]).shape == (0, 32)


def test_coop_variable_base_matches_oracle(coop_engine, oracle):
    eng = coop_engine
    q = [v for v in KATS["quirk_mul"] if v["ok"]]
    sc = np.frombuffer(b"".join(bytes.fromhex(v["scalar"]) for v in q), dtype=np.uint8).reshape(-1, 32)
    pts = np.stack([oracle.decode(bytes.fromhex(v["point"]))[0] for v in q])
    got = eng.mul(sc, pts_ext=pts)
    assert [bytes(r).hex() for r in got] == [v["out"] for v in q]
    # the same from the wire encodings (decode kernel in front), invalid encodings included
    penc = np.frombuffer(b"".join(bytes.fromhex(v["point"]) for v in KATS["quirk_mul"]), dtype=np.uint8).reshape(-1, 32)
    sc_all = np.frombuffer(b"".join(bytes.fromhex(v["scalar"]) for v in KATS["quirk_mul"]), dtype=np.uint8).reshape(-1, 32)
    enc, ok = eng.mul(sc_all, pts_enc=penc, want_ok=True)
    for v, e, o in zip(KATS["quirk_mul"], enc, ok):
        assert bool(o) == bool(v["ok"])
        if v["ok"]:
            assert bytes(e).hex() == v["out"]
    for n in (1, 3, 64, 65, 257):
        s = synth.raw256(n, 500 + n)
        p = oracle.mul_base_ext_batch(synth.scalars(n, 600 + n, b"point"))
        enc, ext = eng.mul(s, pts_ext=p, want_ext=True)
        assert np.array_equal(enc, oracle.mul_batch(s, p, nthreads=8)), n
        assert [oracle.encode(e) for e in ext[:4]] == [bytes(r) for r in enc[:4]]
    # projective inputs with Z != 1 (sums of points) and reduced scalars
    a = oracle.mul_base_ext_batch(synth.scalars(40, 700, b"point"))
    b = np.stack([oracle.add(x, y) for x, y in zip(a, np.roll(a, 1, axis=0))])
    s = synth.scalars(40, 701)
    assert np.array_equal(eng.mul(s, pts_ext=b), oracle.mul_batch(s, b, nthreads=8))


def test_variable_base_in_four_pieces_around_the_cuts(engine, oracle):
    """k_mul_coop with an item's scalar cut at bits 144 / 210 / 241 and every piece on a workgroup of its own (4 n <= coop.verify_max_items): scalars
    whose pieces are zero, one, all ones, and that straddle a cut — on ordinary points, the neutral element, small-order points and projective sums —
    against the oracle and against the one-wavefront form; the arrival counters and records in device scratch must be left clean (call after call, sizes
    going up and down)."""
    import kyber_rs_amd
    cuts = (144, 210, 241)
    vals = [0, 1, 2, 8, synth.L, synth.L - 1, synth.L + 1, 8 * synth.L, (1 << 255) - 19, (1 << 256) - 1, 1 << 255, (1 << 252) + 1]
    for c in cuts:
        vals += [1 << c, (1 << c) - 1, (1 << c) + 1, 1 << (c - 1), ((1 << 256) - 1) ^ ((1 << c) - 1), (1 << c) | 1, 3 << (c - 1)]
    vals += [(1 << 144) | (1 << 241), ((1 << 210) - 1) ^ ((1 << 144) - 1), ((1 << 241) - 1) ^ ((1 << 210) - 1), 0xf << 252, (1 << 253) - 1]
    sc = np.frombuffer(b"".join((v % (1 << 256)).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32)
    n = len(sc)
    assert 4 * n <= engine.get_option("coop.verify_max_items")           # the whole set travels in pieces
    ordinary = oracle.mul_base_ext_batch(synth.scalars(n, 811, b"point"))
    sums = np.stack([oracle.add(x, y) for x, y in zip(ordinary, np.roll(ordinary, 3, axis=0))])      # Z != 1
    weak = [oracle.decode(k)[0] for k in oracle.weak_keys() if oracle.decode(k)[1]]
    special = np.stack([oracle.null()] + weak)
    keep = engine.get_option("coop.verify_max_items")
    for pts in (ordinary, sums, np.stack([special[i % len(special)] for i in range(n)])):
        want = oracle.mul_batch(sc, pts, nthreads=8)
        got = engine.mul(sc, pts_ext=pts)
        assert np.array_equal(got, want)
        engine.set_option("coop.verify_max_items", 0)                     # one wavefront per item
        try:
            assert np.array_equal(engine.mul(sc, pts_ext=pts), want)
        finally:
            engine.set_option("coop.verify_max_items", keep)
    # sizes up and down through the hand-over (256 | 257 on 256 compute units) and back to one item: nothing of an earlier call may be left in the scratch
    for m in (1, 128, 3, 256, 2, 257, 129, 64, 1):
        s, p = np.resize(sc, (m, 32)), np.resize(ordinary, (m, 40))
        assert np.array_equal(engine.mul(s, pts_ext=p), oracle.mul_batch(s, p, nthreads=8)), m
    # projective limbs out: the same point whichever workgroup comes last (the four are added in a fixed order)
    eng = kyber_rs_amd.Engine(0, private=True)
    try:
        eng.set_option("ext.projective", 1)
        first = eng.mul(sc, pts_ext=ordinary, ext_only=True)
        for _ in range(5):
            assert np.array_equal(eng.mul(sc, pts_ext=ordinary, ext_only=True), first)
        assert np.array_equal(eng.encode(first), oracle.mul_batch(sc, ordinary, nthreads=8))
    finally:
        eng.close()


def test_coop_decode_sign_verify_match_oracle(coop_xengine, oracle):
    coop_engine = coop_xengine
    """the small-batch forms of unmarshal_binary (cooperative square-root chain), Schnorr signing and both verification
    flavours (cooperative decodes of A and R, both multiplications projective into the final comparison): weak keys, invalid
    and non-canonical encodings, every reject reason, the 1024 golden EdDSA lines"""
    import gzip
    import hashlib
    from test_device_source_on_host import verify_cases
    eng = coop_engine
    rng = np.random.default_rng(15)
    encs = [bytes.fromhex(h) for h in KATS["weak_keys"] + KATS["invalid_encodings"] + [KATS["decode_kat"]]]
    P = 2**255 - 19
    encs += [(P + k).to_bytes(32, "little") for k in range(19)] + [bytes([1] + [0] * 30 + [0x80])]
    encs += [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(300)]
    ext, ok = eng.decode(np.frombuffer(b"".join(encs), dtype=np.uint8))
    for i, e in enumerate(encs):
        oext, ook = oracle.decode(e)
        assert ok[i] == ook, e.hex()
        if ook:
            assert oracle.encode(ext[i]) == oracle.encode(oext)
    cases = verify_cases(oracle)
    pubs = np.frombuffer(b"".join(c[0] for c in cases), dtype=np.uint8)
    sigs = np.frombuffer(b"".join(c[2] for c in cases), dtype=np.uint8)
    msgs = [c[1] for c in cases]
    vmax = eng.get_option("coop.verify_max_items")
    assert vmax >= 64
    for flavor in (0, 1):
        want = np.array([oracle.verify(flavor, *c) for c in cases], dtype=np.uint8)
        assert set(want.tolist()) >= {0, 2, 3, 4, 5, 6, 7, 8, 9}
        # the single-launch kernel (three wavefronts per signature: hash + ladder on y alone | both decodes | s B)
        eng.profile_begin(16)
        assert np.array_equal(eng.verify(pubs, msgs, sigs, flavor), want)
        assert [k for k, _ in eng.profile_read(16)] == ["k_verify_coop"]
        eng.profile_begin(0)
        for m in (1, 2, 5):
            assert np.array_equal(eng.verify(pubs[:32 * m], msgs[:m], sigs[:64 * m], flavor), want[:m])
        # the five-kernel sequence of the one-item-per-wavefront kernels (what larger small batches take)
        eng.set_option("coop.verify_max_items", 0)
        try:
            for overlap in (1, 0):
                eng.set_option("verify.overlap", overlap)
                eng.profile_begin(16)
                assert np.array_equal(eng.verify(pubs, msgs, sigs, flavor), want)
                names = [k for k, _ in eng.profile_read(16)]
                assert "k_mul_coop" in names and "k_mul_base_coop" in names and "k_mul_ladder" not in names
                assert np.array_equal(eng.verify(pubs[:32], msgs[:1], sigs[:64], flavor), want[:1])
        finally:
            eng.profile_begin(0)
            eng.set_option("verify.overlap", 1)
            eng.set_option("coop.verify_max_items", vmax)
    xs, ks, ms, ss, ps = [], [], [], [], []
    for ln in gzip.open(os.path.join(HERE, "golden", "sign.input.gz"), "rt").read().split("\n"):
        if not ln:
            continue
        p = ln.split(":")
        seed, msg, sig = bytes.fromhex(p[0])[:32], bytes.fromhex(p[2]), bytes.fromhex(p[3])[:64]
        d = bytearray(hashlib.sha512(seed).digest())
        d[0] &= 0xF8; d[31] &= 0x7F; d[31] |= 0x40
        r = int.from_bytes(hashlib.sha512(bytes(d[32:]) + msg).digest(), "little") % synth.L
        xs.append(bytes(d[:32])); ks.append(r.to_bytes(32, "little")); ms.append(msg); ss.append(sig); ps.append(bytes.fromhex(p[1]))
    u8 = lambda lst: np.frombuffer(b"".join(lst), dtype=np.uint8)
    assert [bytes(r) for r in eng.schnorr_sign(u8(xs), u8(ks), ms)] == ss
    assert not eng.verify(u8(ps), ms, u8(ss), 0).any()                   # 1024 > coop.verify_max_items: the kernel sequence
    assert not eng.verify(u8(ps[:400]), ms[:400], u8(ss[:400]), 0).any()   # the single-launch kernel, messages of 0..399 bytes
    bad = bytearray(b"".join(ss)); bad[40] ^= 1
    assert eng.verify(u8(ps), ms, np.frombuffer(bytes(bad), dtype=np.uint8), 0)[0] == 9
    assert eng.verify(u8(ps[:3]), ms[:3], np.frombuffer(bytes(bad[:192]), dtype=np.uint8), 0).tolist() == [9, 0, 0]


def test_coop_pubpoly_eval_matches_oracle(coop_xengine, oracle):
    coop_engine = coop_xengine
    """PubPoly::eval one evaluation per wavefront (cooperative doubling / addition): indices of every bit length up to 2^32 - 2,
    commitments with small-order components and the neutral element among them, the many-polynomials form"""
    eng = coop_engine
    t = 9
    commits = oracle.mul_base_ext_batch(synth.scalars(t, 900, b"coef"))
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"][:5]]
    commits[2] = oracle.add(commits[2], weak[2])                   # a point of mixed order
    commits[5] = weak[0]                                           # a small-order point
    commits[7] = oracle.decode(bytes([1] + [0] * 31))[0]           # the neutral element
    idx = np.array([0, 1, 2, 3, 6, 7, 254, 255, 256, 1023, 65534, 65535, (1 << 31) - 1, (1 << 32) - 2], dtype=np.uint32)
    eng.profile_begin(4)
    got = eng.pubpoly_eval(commits, idx)
    assert [k for k, _ in eng.profile_read(4)] == ["k_poly_eval_coop"]
    eng.profile_begin(0)
    for g, i in zip(got, idx):
        assert bytes(g) == oracle.pubpoly_eval(commits, int(i)), int(i)
    enc, ext = eng.pubpoly_eval(commits, idx[:5], want_ext=True)
    assert [oracle.encode(e) for e in ext] == [bytes(g) for g in got[:5]]
    polys = np.stack([commits, np.roll(commits, 3, axis=0), commits[::-1].copy()])
    ii = np.array([[4, 77], [0, 1000], [12345, 2]], dtype=np.uint32)
    multi = eng.pubpoly_eval_multi(polys, ii)
    for p in range(3):
        for q in range(2):
            assert bytes(multi[p, q]) == oracle.pubpoly_eval(polys[p], int(ii[p, q]))
    one = eng.pubpoly_eval(commits[:1], idx[:3])                   # degree 0
    assert all(bytes(o) == oracle.encode(commits[0]) for o in one)
    # several wavefronts per evaluation: the chain cut into segments, recombined with x^(s len) mod 8L — exact on the commitments with
    # small-order components too (mod L would not be)
    want = [oracle.pubpoly_eval(commits, int(i)) for i in idx]
    try:
        for segs in (2, 3, 4, 8, 9):
            eng.set_option("poly.segments", segs)
            assert [bytes(g) for g in eng.pubpoly_eval(commits, idx)] == want, segs
            multi = eng.pubpoly_eval_multi(polys, ii)
            assert all(bytes(multi[p, q]) == oracle.pubpoly_eval(polys[p], int(ii[p, q])) for p in range(3) for q in range(2))
    finally:
        eng.set_option("poly.segments", 0)
    # a longer polynomial, segments chosen by the engine (t = 150 -> 6 wavefronts per evaluation), and the maximum of 32
    t2 = 150
    long_c = oracle.mul_base_ext_batch(synth.scalars(t2, 901, b"coef"))
    long_c[17] = oracle.add(long_c[17], weak[3]); long_c[149] = oracle.add(long_c[149], weak[4]); long_c[75] = weak[1]
    for i in (1, 6, 1000, 65535):
        assert bytes(eng.pubpoly_eval(long_c, np.array([i], dtype=np.uint32))[0]) == oracle.pubpoly_eval(long_c, i), i
    eng.set_option("poly.segments", 32)
    try:
        i3 = np.array([5, 77, 4000], dtype=np.uint32)
        assert [bytes(g) for g in eng.pubpoly_eval(long_c, i3)] == [oracle.pubpoly_eval(long_c, int(i)) for i in i3]
    finally:
        eng.set_option("poly.segments", 0)


def test_coop_single_launch_signing(coop_engine, oracle):
    """schnorr::sign / EdDSA::sign for up to 512 signatures in one launch (two wavefronts per signature), with and without the
    signer's stored public key: the reference's golden EdDSA lines (ragged messages), random triples, unreduced secrets"""
    import gzip
    eng = coop_engine
    seeds, pubs, msgs, sigs = [], [], [], []
    for ln in gzip.open(os.path.join(HERE, "golden", "sign.input.gz"), "rt").read().split("\n")[:400]:
        if ln:
            p = ln.split(":")
            seeds.append(bytes.fromhex(p[0])[:32]); pubs.append(bytes.fromhex(p[1])); msgs.append(bytes.fromhex(p[2])); sigs.append(bytes.fromhex(p[3])[:64])
    for v in KATS["rfc8032"]:
        seeds.append(bytes.fromhex(v["private"])); pubs.append(bytes.fromhex(v["public"])); msgs.append(bytes.fromhex(v["message"])); sigs.append(bytes.fromhex(v["signature"]))
    u8 = lambda lst: np.frombuffer(b"".join(lst), dtype=np.uint8)
    eng.profile_begin(8)
    sig, pub = eng.eddsa_sign(u8(seeds), msgs, want_pub=True)
    assert [k for k, _ in eng.profile_read(8)] == ["k_eddsa_prep", "k_sign_coop"]
    eng.profile_begin(0)
    assert [bytes(r) for r in pub] == pubs and [bytes(r) for r in sig] == sigs
    assert [bytes(r) for r in eng.eddsa_sign(u8(seeds), msgs, pubs=pub)] == sigs
    assert [bytes(r) for r in eng.eddsa_sign(u8(seeds[:1]), msgs[:1])] == sigs[:1]
    n = 300
    x, k = synth.raw256(n, 41), synth.scalars(n, 42, b"k")                 # secrets as the clamped EdDSA ones: not reduced mod L
    x[:, 31] &= 0x7f
    ms = synth.messages(n, 43)
    want = oracle.schnorr_sign_batch(x, k, ms, nthreads=8)
    assert np.array_equal(eng.schnorr_sign(x, k, ms), want)
    assert np.array_equal(eng.schnorr_sign(x, k, ms, pubs=eng.mul_base(x)), want)
    for m in (1, 2, 63, 65):
        assert np.array_equal(eng.schnorr_sign(x[:m], k[:m], ms[:m]), want[:m])


def test_parity_suites_again_on_the_small_batch_kernels(coop_engine, oracle):
    """the parity tests of linear combinations, decode / encode / add / sub and polynomial evaluation (tests/test_gpu_parity.py,
    which the session runs with the one-item-per-wavefront kernels switched off) once more with them switched on:
    products of a linear combination one per wavefront and handed over projective, cooperative finish and encode"""
    import test_gpu_parity as P
    eng = coop_engine
    sc = synth.scalars(12, 61).reshape(3, 4, 32)
    pts = oracle.mul_base_ext_batch(synth.scalars(12, 62)).reshape(3, 4, 40)
    eng.profile_begin(8)
    got = eng.lincomb(sc, pts_ext=pts)
    names = [k for k, _ in eng.profile_read(8)]
    eng.profile_begin(0)
    assert names[0] == "k_mul_coop" and names[-1] == "k_finish_coop" and "k_mul_ladder" not in names
    assert [bytes(g) for g in got] == [oracle.lincomb(sc[g], pts[g]) for g in range(3)]
    P.test_lincomb_matches_oracle(eng, oracle)
    P.test_decode_encode_add_sub(eng, oracle)
    P.test_pubpoly_eval_and_equal(eng, oracle)
    P.test_quirk_vectors(eng, oracle)
    P.test_mul_from_encodings_and_invalid_points(eng, oracle)


def test_structured_fuzz_on_the_small_batch_kernels(coop_engine, oracle):
    """the 2^16 structured (scalar, point) pairs of tests/test_gpu_parity.py (bit runs, recoding extremes, multiples of L, torsion and
    small-order points, the neutral element) through k_mul_base_coop and k_mul_coop: image, ladder, y-recovery and exceptional cases
    in quads, every output against the oracle"""
    import test_gpu_parity as P
    P.test_structured_fuzz_against_oracle(coop_engine, oracle)


def test_projective_extended_results_on_request(oracle):
    """option ext.projective: a small-batch multiplication asked for extended limbs ONLY hands the point over as (X : Y : Z : T) with
    Z != 1 (no inversion) — the encodings of those points, and everything computed from them, are unchanged"""
    import kyber_rs_amd
    eng = kyber_rs_amd.Engine(0, private=True)
    try:
        eng.set_option("ext.projective", 1)
        small = 5 * eng.get_option("device.cus")                           # (engine.hip COOP_BASE_TO_QUARTERS_PER_CU: above it the fixed base is a batch kernel, affine limbs)
        for n in (1, 5, 300, 700, small, 2000):                            # four wavefronts per item / one / the mid-size form
            s, k = synth.raw256(n, 51), synth.scalars(n, 52)
            want_b = oracle.mul_base_batch(s, nthreads=8)
            eb = eng.mul_base(s, ext_only=True)
            assert [oracle.encode(e) for e in eb[:40]] == [bytes(w) for w in want_b[:40]]
            assert any(list(e[20:30]) != [1] + [0] * 9 for e in eb[:8]) == (n <= small)   # really projective where the option applies
            assert np.array_equal(eng.encode(eb), want_b)
            em = eng.mul(k, pts_ext=eb, ext_only=True)                     # projective points in, projective points out
            want_m = oracle.mul_batch(k, oracle.mul_base_ext_batch(s), nthreads=8)
            assert np.array_equal(eng.encode(em), want_m)
            assert np.array_equal(eng.mul(k, pts_ext=eb), want_m)          # with an encoding asked for: the affine path
            enc, ext = eng.mul(k, pts_ext=eb, want_ext=True)
            assert all(list(e[20:30]) == [1] + [0] * 9 for e in ext[:8])
            assert eng.equal(em, ext).all()
            sums = eng.add(em, eb)
            assert [oracle.encode(x) for x in sums[:20]] == [oracle.encode(oracle.add(oracle.decode(bytes(a))[0], oracle.decode(bytes(b))[0])) for a, b in zip(want_m[:20], want_b[:20])]
        q = [v for v in KATS["quirk_mul"] if v["ok"]]
        sc = np.frombuffer(b"".join(bytes.fromhex(v["scalar"]) for v in q), dtype=np.uint8).reshape(-1, 32)
        pts = np.stack([oracle.decode(bytes.fromhex(v["point"]))[0] for v in q])
        assert [bytes(r).hex() for r in eng.encode(eng.mul(sc, pts_ext=pts, ext_only=True))] == [v["out"] for v in q]
        # polynomial evaluation (one wavefront per evaluation, and segmented) and sums hand over projective as well
        commits = eng.mul_base(synth.scalars(150, 53), ext_only=True)
        idx = np.array([0, 5, 1000], dtype=np.uint32)
        ref_commits = oracle.mul_base_ext_batch(synth.scalars(150, 53))
        for t in (9, 150):
            ev = eng.pubpoly_eval(commits[:t], idx, ext_only=True)
            assert [bytes(r) for r in eng.encode(ev)] == [oracle.pubpoly_eval(ref_commits[:t], int(i)) for i in idx], t
        sm = eng.sum_points(commits[:12].reshape(3, 4, 40), ext_only=True)
        assert np.array_equal(eng.encode(sm), eng.sum_points(commits[:12].reshape(3, 4, 40)))
    finally:
        eng.close()


@pytest.mark.parametrize("cus", [0, 64])
def test_every_routing_boundary_with_default_options(oracle, cus):
    """a fresh context with the DEFAULT thresholds — wavefronts per compute unit times the compute units the context works with: the device's, and
    64 declared with option device.cus (a CU-masked stream, a partition) — batch sizes on both sides of every routing boundary the options imply
    (multi-wavefront kernels, mul from encodings, cooperative kernels, two-lane ladder, batch kernels above), all four operations against the oracle"""
    import kyber_rs_amd
    eng = kyber_rs_amd.Engine(0, private=True)
    try:
        hw = eng.get_option("device.cus")
        assert hw == eng.device_info()["compute_units"]
        if cus:
            eng.set_option("device.cus", cus)
        n_cu = cus or hw
        opt = {k: eng.get_option(k) for k in ("coop.max_items", "coop.base_max_items", "coop.ladder_max_items", "coop.ladder_enc_max_items", "coop.decode_max_items",
                                              "coop.verify_max_items", "ladder.pair_max_items", "ladder.quad_max_items")}
        assert opt == {"coop.max_items": 24 * n_cu, "coop.base_max_items": 18 * n_cu, "coop.ladder_max_items": 14 * n_cu, "coop.ladder_enc_max_items": 6 * n_cu,
                       "coop.decode_max_items": 4 * n_cu, "coop.verify_max_items": 2 * n_cu, "ladder.pair_max_items": 128 * n_cu, "ladder.quad_max_items": 64 * n_cu}, opt      # no absolute item count among the defaults
        # every size the routing of engine.hip compares a batch with, from these options (x/2, x/4, 2x, 7x/8: the derived comparisons there)
        marks = set()
        for v in opt.values():
            marks |= {v, v // 2, v // 4, 2 * v, 7 * v // 8}
        marks |= {4 * n_cu, 5 * n_cu, 15 * n_cu // 4, 9 * n_cu, 128 * n_cu, 8 * n_cu}      # (5: fixed base hands over to its four-wavefronts-per-64-items form, signing at 3/4 of it; 128: that form's last size)
        sizes = sorted({n for m in marks for n in (m - 1, m, m + 1) if 1 <= n <= 24 * n_cu + 1 and n <= 6200})
        nmax = max(sizes)
        s = synth.raw256(nmax, 31); s[::7] = synth.scalars(len(s[::7]), 32)
        k = synth.scalars(nmax, 33, b"k")
        pts = oracle.mul_base_ext_batch(synth.scalars(nmax, 34, b"point"))
        enc_max = min(nmax, 8 * n_cu + 64)
        enc = np.stack([np.frombuffer(oracle.encode(p), dtype=np.uint8) for p in pts[:enc_max]])
        msgs = synth.messages(nmax, 35)
        want_base = oracle.mul_base_batch(s, nthreads=8)
        want_mul = oracle.mul_batch(s, pts, nthreads=8)
        x = s.copy(); x[:, 31] &= 0x7f
        want_sig = oracle.schnorr_sign_batch(x, k, msgs, nthreads=8)
        pubs = oracle.mul_base_batch(x, nthreads=8)
        bad = want_sig.copy(); bad[::5, 40] ^= 1
        want_st = np.array([0 if i % 5 else 9 for i in range(nmax)], dtype=np.uint8)
        for n in sizes:
            assert np.array_equal(eng.mul_base(s[:n]), want_base[:n]), n
            assert np.array_equal(eng.mul(s[:n], pts_ext=pts[:n]), want_mul[:n]), n
            if n <= enc_max:
                assert np.array_equal(eng.mul(s[:n], pts_enc=enc[:n]), want_mul[:n]), n
            assert np.array_equal(eng.schnorr_sign(x[:n], k[:n], msgs[:n]), want_sig[:n]), n
            assert np.array_equal(eng.verify(pubs[:n], msgs[:n], bad[:n], 1), want_st[:n]), n
        # marshal_binary of PROJECTIVE points around its hand-over (coop.decode_max_items): one point per wavefront with the inversion over its lanes |
        # one point per lane and one inversion per wavefront (k_finish_wave; a last wavefront of 1, 63 and 64 live lanes)
        dm = opt["coop.decode_max_items"]
        proj = np.stack([oracle.add(a_, b_) for a_, b_ in zip(pts[: dm + 130], np.roll(pts[: dm + 130], 1, axis=0))])
        want_enc = np.stack([np.frombuffer(oracle.encode(e), dtype=np.uint8) for e in proj])
        for n in (dm - 1, dm, dm + 1, dm + 63, dm + 64, dm + 65, dm + 128):
            assert np.array_equal(eng.encode(proj[:n]), want_enc[:n]), n
        eng.profile_begin(4)
        eng.encode(proj[:dm]); eng.encode(proj[: dm + 1])
        assert [nm for nm, _ in eng.profile_read(4)] == ["k_finish_coop", "k_encode_batched"]
        eng.profile_begin(0)
        # the kernel families really change with the declared CU count: 1,000 variable-base items are a one-item-per-wavefront launch on 256 CUs
        # (9 per CU = 2,304) and a four-lane ladder launch on 64 (576)
        eng.profile_begin(16)
        eng.mul(s[:1000], pts_ext=pts[:1000])
        names = [nm for nm, _ in eng.profile_read(16)]
        assert ("k_mul_coop" in names) == (1000 <= 9 * n_cu) and ("k_mul_ladder_pair" in names) == (1000 > 9 * n_cu), (n_cu, names)      # (9: to the four-lane ladder, profiled under the two-lane ladder's name)
        if cus:
            eng.set_option("device.cus", 0)
            assert eng.get_option("coop.max_items") == 24 * hw
    finally:
        eng.close()


def test_coop_and_batch_kernels_agree_at_the_threshold(engine, oracle):
    """the routing by size: n <= coop.max_items -> cooperative kernel, above -> batch kernels; same bytes either way"""
    s = synth.raw256(130, 800)
    p = oracle.mul_base_ext_batch(synth.scalars(130, 801, b"point"))
    old = (engine.get_option("coop.max_items"), engine.get_option("coop.base_max_items"))
    engine.set_option("coop.max_items", 0)
    engine.set_option("coop.base_max_items", 0)
    ref_mul, ref_base = engine.mul(s, pts_ext=p), engine.mul_base(s)
    try:
        engine.set_option("coop.max_items", 64)
        engine.set_option("coop.base_max_items", 64)
        engine.profile_begin(64)
        assert np.array_equal(engine.mul(s[:64], pts_ext=p[:64]), ref_mul[:64])
        assert np.array_equal(engine.mul(s[:65], pts_ext=p[:65]), ref_mul[:65])
        assert np.array_equal(engine.mul_base(s[:64]), ref_base[:64])
        assert np.array_equal(engine.mul_base(s[:65]), ref_base[:65])
        names = [k for k, _ in engine.profile_read(64)]
        assert names.count("k_mul_coop") == 1 and names.count("k_mul_base_coop") == 1 and "k_mul_ladder_pair" in names
    finally:
        engine.profile_begin(0)
        engine.set_option("coop.max_items", old[0])          # (this used to leave the session's engine with the small-batch kernels switched off)
        engine.set_option("coop.base_max_items", old[1])


def test_the_wavefront_inversion_matches_the_exponentiation_and_python(xengine):
    """csrc/kernels_coop.hip fe_invert_gcd_wave (safegcd over the lanes; what every one-item kernel ends in since round 6) through the test hook of the
    cross-check build (op 9), against the cooperative exponentiation it replaced (op 2, cinv) and against Python's pow(z, p - 2, p): edge values
    (0 -> 0, 1, p - 1, values around 2^255, small and huge) and random ones; tools/gcd_lanes_model.py is the limb-exact model of the same code."""
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import coop_model as M
    import kyber_rs_amd
    lib = kyber_rs_amd.load_library(crosscheck=True)
    xengine.device_info()
    c = M.lane_consts()
    P = 2**255 - 19
    rng = random.Random(606)
    vals = [0, 1, 2, 3, 19, P - 1, P - 2, (P - 1) // 2, 2**254, 2**255 - 20, 2**30, 2**30 - 1, 2**240, 2**252 + 27742317777372353535851937790883648493]
    vals += [rng.randrange(P) for _ in range(150)] + [rng.randrange(2**64) for _ in range(20)] + [P - rng.randrange(2**64) for _ in range(20)]
    for z in vals:
        A = np.ascontiguousarray(M.quad_from_ints(c, [z, z, z, z]), dtype=np.uint32)
        B = np.zeros(64, np.uint32)
        got = {}
        for op in (9, 2):
            out = np.zeros(64, np.uint32)
            assert lib.kyb_diag_coop(op, A.ctypes.data, B.ctypes.data, out.ctypes.data) == 0
            got[op] = [v % P for v in M.ints_from_quad(out.astype(np.uint64))]
        want = pow(z, P - 2, P)
        assert got[9] == [want] * 4, (hex(z), got[9][0], want)
        assert got[2] == [want] * 4, hex(z)


def test_coop_primitives_match_the_lane_model(xengine):
    """every cooperative primitive (cmul4, cnorm, csub, cinv, table entry, mixed addition, ladder step, layout round
    trip) run by one wavefront through the test hook of the cross-check build (the same device code as the product's; the product
    library does not export the hook) == the lane-level numpy model tools/coop_model.py"""
    engine = xengine
    import ctypes
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import coop_model as M
    import kyber_rs_amd
    lib = kyber_rs_amd.load_library(crosscheck=True)
    engine.device_info()                       # makes the fixture's context current on this thread

    def run(op, A, B=None):
        A = np.ascontiguousarray(A, dtype=np.uint32)
        B = np.zeros(64, np.uint32) if B is None else np.ascontiguousarray(B, dtype=np.uint32)
        out = np.zeros(64, np.uint32)
        assert lib.kyb_diag_coop(op, A.ctypes.data, B.ctypes.data, out.ctypes.data) == 0
        return out.astype(np.uint64)

    c = M.lane_consts()
    rnd = random.Random(5)
    act = (np.arange(64) & 15) < 10            # lanes 10..15 of a row are don't-care (coop25519.h)
    same = lambda got, want: np.array_equal(np.asarray(got)[act], np.asarray(want)[act])
    for _ in range(4):
        a = [rnd.randrange(M.P) for _ in range(4)]
        b = [rnd.randrange(M.P) for _ in range(4)]
        F, G = M.quad_from_ints(c, a), M.quad_from_ints(c, b)
        assert same(run(7, F), F)
        assert same(run(1, M.cadd(F, G)), M.cnorm(c, M.cadd(F, G)))
        assert same(run(5, F, G), M.csub(c, F, G))
        G4 = M.cadd(M.cadd(G, G), M.cadd(G, G))
        assert same(run(0, F, G4), M.cmul4(c, F, G4))
        assert same(run(8, F), M.cmul4(c, F, F))                  # the symmetric squaring gives the very limbs of the product
        assert M.ints_from_quad(run(0, F, G)) == [x * y % M.P for x, y in zip(a, b)]
        assert M.ints_from_quad(run(2, F)) == [pow(x, M.P - 2, M.P) for x in a]
        h = M.quad_from_ints(c, [rnd.randrange(M.P) for _ in range(4)])
        E = M.quad_from_ints(c, [rnd.randrange(M.P), rnd.randrange(M.P), rnd.randrange(M.P), 0])
        assert same(run(3, h, E), M.madd(c, h, E))
        for swap0 in (0, 1):
            for bit in (0, 1):
                S = M.quad_from_ints(c, [rnd.randrange(M.P) for _ in range(4)])
                UWQ = M.quad_from_ints(c, [rnd.randrange(M.P), 0, rnd.randrange(M.P), 0])       # U1 in row 0, W1 in row 2
                B = UWQ.copy(); B[16] = swap0; B[17] = bit
                assert same(run(6, S, B), M.ladder_step(c, S, UWQ, swap0, bit)[0])
    assert M.ints_from_quad(run(2, M.quad_from_ints(c, [0, 1, M.P - 1, 2]))) == [0, 1, M.P - 1, pow(2, M.P - 2, M.P)]
    img64 = engine.base_table().view(np.uint32)[(65536 + 106496) // 4:]
    for pos, idx, neg in [(0, 0, 0), (0, 5, 0), (0, 31, 1), (7, 12, 1), (41, 8, 0), (42, 3, 0), (42, 15, 0)]:
        B = np.zeros(64, np.uint32); B[0], B[1], B[2] = pos, idx, neg
        n_e, base = (32, pos * 960) if pos < 42 else (16, 42 * 960)

        def word(j, k):
            g_, r_ = k // 10, k % 10
            big = 0 if g_ == 0 else (8 * n_e if g_ == 1 else 20 * n_e)
            small = 16 * n_e if g_ == 0 else (18 * n_e if g_ == 1 else 28 * n_e)
            return base + (big + ((r_ >> 2) * n_e + j) * 4 + (r_ & 3) if r_ < 8 else small + j * 2 + (r_ - 8))
        want = np.zeros(64, np.uint64)
        for g_ in range(3):
            ge = (g_ ^ neg) if g_ < 2 else g_
            for k in range(10):
                v = int(img64[word(idx, 10 * ge + k)])
                want[16 * g_ + k] = (M.P2[k] - v) if (g_ == 2 and neg) else v
        assert same(run(4, np.zeros(64, np.uint32), B), want), (pos, idx, neg)


def test_short_public_scalars_take_a_short_ladder(engine, oracle):
    """kyb_mul_public_batch: a host-pointer mul of a few items whose multipliers the caller DECLARES public and which are ALL below 2^64
    (share indices, the cofactor: what PubPoly::eval and Point::pick multiply by, poly.rs:464, point.rs:148) starts its ladder below the
    leading zeros; same points as the full ladder and as the reference's 64-window routine, also on small-order and mixed-order operands.
    kyb_mul_batch itself never shortens (mul.short_scalars defaults to 0); with the option set it behaves like the public call."""
    assert engine.get_option("mul.short_scalars") == 0
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    pts = list(oracle.mul_base_ext_batch(synth.scalars(6, 501, b"short")))
    pts[1] = oracle.add(pts[1], weak[2])
    pts[2] = weak[3]
    pts[3] = oracle.null()
    small = [0, 1, 2, 3, 8, 513, 65535, 65536, (1 << 32) - 1, 1 << 32, (1 << 63) + 5, (1 << 64) - 1]
    for v in small:
        for p in pts:
            s = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)[None, :]
            got, ext = engine.mul(s, pts_ext=p[None, :], want_ext=True, public=True)
            assert bytes(got[0]) == oracle.mul(bytes(s[0]), p), (v, "enc")
            assert oracle.encode(ext[0]) == bytes(got[0])
    # several items in one call: short only when every scalar is short
    sc = np.stack([np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8) for v in small[:6]])
    P = np.stack(pts)
    want = oracle.mul_batch(sc, P)
    assert np.array_equal(engine.mul(sc, pts_ext=P, public=True), want)
    mixed = sc.copy(); mixed[4] = synth.scalars(1, 9)[0]
    assert np.array_equal(engine.mul(mixed, pts_ext=P, public=True), oracle.mul_batch(mixed, P))
    big = np.frombuffer((1 << 64).to_bytes(32, "little"), dtype=np.uint8)[None, :]
    assert bytes(engine.mul(big, pts_ext=P[:1], public=True)[0]) == oracle.mul(bytes(big[0]), P[0])
    # the plain call (full ladder), then the plain call with the option that treats every call as public
    assert np.array_equal(engine.mul(sc, pts_ext=P), want)
    engine.set_option("mul.short_scalars", 1)
    try:
        assert np.array_equal(engine.mul(sc, pts_ext=P), want)
    finally:
        engine.set_option("mul.short_scalars", 0)
    # the declaration is what shortens the ladder: a 10-bit multiplier is several times faster through the public call
    import time
    one = np.frombuffer((513).to_bytes(32, "little"), dtype=np.uint8)[None, :]
    def med(fn):
        fn(); ts = []
        for _ in range(30):
            a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
        return sorted(ts)[len(ts) // 2]
    t_pub, t_plain = med(lambda: engine.mul(one, pts_ext=P[:1], ext_only=True, public=True)), med(lambda: engine.mul(one, pts_ext=P[:1], ext_only=True))
    assert t_pub < 0.7 * t_plain, (t_pub, t_plain)
    # projective hand-over keeps working on the short path
    engine.set_option("ext.projective", 1)
    try:
        ext = engine.mul(sc, pts_ext=P, ext_only=True, public=True)
        assert np.array_equal(engine.encode(ext), want)
    finally:
        engine.set_option("ext.projective", 0)


def test_encode_of_few_points_with_and_without_a_literal_z_of_one(engine, oracle):
    """kyb_encode_batch on a handful of points (one point per wavefront, k_finish_coop): points whose Z is literally (1, 0, ..., 0) skip the
    inversion, everything else — projective Z, Z = 0, and a Z that IS one but not written as the literal (1 + p in limbs) — takes it; every
    encoding against the oracle, limbs out as well"""
    n = 24
    aff = oracle.mul_base_ext_batch(synth.scalars(n, 611))                     # the oracle's multiplication leaves Z != 1
    enc = np.stack([np.frombuffer(oracle.encode(p), dtype=np.uint8) for p in aff])
    dec = np.stack([oracle.decode(bytes(e))[0] for e in enc])                   # decoded: Z literally one
    assert (dec[:, 20] == 1).all() and not dec[:, 21:30].any() and (aff[:, 20:30] != dec[:, 20:30]).any()
    odd = dec.copy()
    p_limbs = np.array([0x3ffffed, 0x1ffffff, 0x3ffffff, 0x1ffffff, 0x3ffffff, 0x1ffffff, 0x3ffffff, 0x1ffffff, 0x3ffffff, 0x1ffffff], dtype=np.int64)
    odd[:, 20:30] = (odd[:, 20:30].astype(np.int64) + p_limbs).astype(np.int32)      # Z = 1 + p: one, but not the literal
    zero_z = dec[:3].copy(); zero_z[:, 20:30] = 0                                # Z = 0: the reference's 0^(p-2) = 0 -> encoding of (0, 0)
    mixed = np.concatenate([dec[:8], aff[8:16], odd[16:]])
    for pts in (dec, aff, odd, mixed, zero_z, dec[:1], aff[:1]):
        want = np.stack([np.frombuffer(oracle.encode(p), dtype=np.uint8) for p in pts])
        assert np.array_equal(engine.encode(pts), want)
    # has_small_order / point checks marshal through the same kernel
    assert np.array_equal(engine.point_checks(pts_ext=dec), engine.point_checks(enc=enc))
