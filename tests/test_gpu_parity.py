"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
inputs — bit-exact 32-byte encodings / 64-byte signatures (integer work: no tolerance).

Sizes: oracle-checked cases finish in seconds; BASELINE.json's full sizes (2^20 / 2^18) are covered by
size-independent properties (fixed-base == variable-base on B, DH commutativity, decode(encode) round
trip, sign == golden) plus an oracle-checked random sample."""
import contextlib
import ctypes
import gzip
import hashlib
import json
import os
import sys
import time

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import synth  # noqa: E402

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(HERE, "golden", "kats.json")))
IDENT = bytes([1] + [0] * 31)


def rows(a):
    return [bytes(r) for r in np.asarray(a, dtype=np.uint8)]


def rand_points_ext(oracle, n, seed):
    """random subgroup points with arbitrary Z, in the reference's limb layout"""
    return oracle.mul_base_ext_batch(synth.scalars(n, seed, b"point"))


@pytest.mark.parametrize("select", [0, 1])
def test_mul_base_matches_oracle(xengine, oracle, select):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    """the radix-16 kernel (64 x 8 table), both selection variants, fused and split finish; then the default kernel"""
    s = np.concatenate([synth.scalars(1500, 1), synth.raw256(549, 1)])
    want = oracle.mul_base_batch(s, nthreads=8)
    engine.set_option("mul_base.select", select)
    engine.set_option("mul_base.radix", 16)
    try:
        assert np.array_equal(engine.mul_base(s), want)
        engine.set_option("finish.min_items", 1 << 20)          # fused per-item inversion
        assert np.array_equal(engine.mul_base(s), want)
    finally:
        engine.set_option("finish.min_items", 1)
        engine.set_option("mul_base.radix", 64)
        engine.set_option("mul_base.select", 1)
    assert np.array_equal(engine.mul_base(s), want)
    for n in (1, 2, 63):                                         # tiny batches through the default (radix-64, split) path
        assert np.array_equal(engine.mul_base(s[:n]), want[:n])


@pytest.mark.parametrize("select", [0, 1])
def test_mul_matches_oracle(xengine, oracle, select):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    """the windowed-table kernel (mul.algo=0), both merge variants"""
    engine.set_option("mul.algo", 0)
    engine.set_option("mul.select", select)
    n = 1029
    s = np.concatenate([synth.scalars(700, 2), synth.raw256(n - 700, 2)])
    pts = rand_points_ext(oracle, n, 2)
    got, ext = engine.mul(s, pts_ext=pts, want_ext=True)
    want = oracle.mul_batch(s, pts, nthreads=8)
    assert np.array_equal(got, want)
    # out_ext is the affine point in reference limbs (Z = 1) and re-encodes to out_enc
    assert np.array_equal(ext[:, 20:30], np.tile(np.array([1] + [0] * 9, dtype=np.int32), (n, 1)))
    for i in range(0, n, 97):
        assert oracle.encode(ext[i]) == bytes(got[i])
    engine.set_option("mul.select", 1)
    engine.set_option("mul.algo", 1)


def test_quirk_vectors(engine, oracle):
    """scalars >= 2^255 (dropped top digit), L, 8L, small-order and non-canonical points"""
    qb = KATS["quirk_mul_base"]
    got = engine.mul_base(np.frombuffer(b"".join(bytes.fromhex(q["scalar"]) for q in qb), dtype=np.uint8))
    assert [bytes(r).hex() for r in got] == [q["out"] for q in qb]
    q = [v for v in KATS["quirk_mul"] if v["ok"]]
    sc = np.frombuffer(b"".join(bytes.fromhex(v["scalar"]) for v in q), dtype=np.uint8)
    pe = np.frombuffer(b"".join(bytes.fromhex(v["point"]) for v in q), dtype=np.uint8)
    got, ok = engine.mul(sc, pts_enc=pe, want_ok=True)
    assert ok.all()
    assert [bytes(r).hex() for r in got] == [v["out"] for v in q]
    # the same through extended-limb inputs
    ext = np.stack([oracle.decode(bytes.fromhex(v["point"]))[0] for v in q])
    assert [bytes(r).hex() for r in engine.mul(sc, pts_ext=ext)] == [v["out"] for v in q]


def test_mul_from_encodings_and_invalid_points(engine, oracle):
    good = [oracle.encode(p) for p in rand_points_ext(oracle, 40, 3)]
    bad = [bytes.fromhex(h) for h in KATS["invalid_encodings"]]
    encs = good[:20] + bad + good[20:]
    s = synth.scalars(len(encs), 3)
    got, ok = engine.mul(s, pts_enc=np.frombuffer(b"".join(encs), dtype=np.uint8), want_ok=True)
    for i, e in enumerate(encs):
        ext, okk = oracle.decode(e)
        assert ok[i] == okk
        if okk:
            assert bytes(got[i]) == oracle.mul(bytes(s[i]), ext)
        else:
            assert bytes(got[i]) == IDENT


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 255, 257, 513])
def test_ragged_sizes(engine, oracle, n):
    s = synth.scalars(n, 4)
    assert np.array_equal(engine.mul_base(s), oracle.mul_base_batch(s)) if n else engine.mul_base(s).shape == (0, 32)
    if n:
        pts = rand_points_ext(oracle, n, 4)
        assert np.array_equal(engine.mul(s, pts_ext=pts), oracle.mul_batch(s, pts, nthreads=8))


def test_decode_encode_add_sub(engine, oracle):
    rng = np.random.default_rng(5)
    encs = [bytes.fromhex(h) for h in KATS["weak_keys"] + KATS["invalid_encodings"] + [KATS["decode_kat"]]]
    P = 2**255 - 19
    encs += [(P + k).to_bytes(32, "little") for k in range(19)] + [bytes([1] + [0] * 30 + [0x80])]
    encs += [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(600)]
    ext, ok = engine.decode(np.frombuffer(b"".join(encs), dtype=np.uint8))
    good = []
    for i, e in enumerate(encs):
        oext, ook = oracle.decode(e)
        assert ok[i] == ook, e.hex()
        if ook:
            assert oracle.encode(ext[i]) == oracle.encode(oext)
            good.append(oext)
    good = np.stack(good)
    enc = engine.encode(good)
    assert rows(enc) == [oracle.encode(g) for g in good]
    a, b = good[:-1], good[1:]
    for sub in (False, True):
        out = engine.add(a, b, subtract=sub)
        assert rows(engine.encode(out)) == [oracle.encode(oracle.add(x, y, sub=sub)) for x, y in zip(a, b)]


def test_point_checks_stand_alone(engine, oracle):
    """kyb_point_checks_batch (PointCanCheckCanonicalAndSmallOrder, point.rs:286-337) == the oracle: from received bytes (WEAK_KEYS with either sign bit
    and their non-canonical aliases, y = p-220 .. p+18 where the reference's is_canonical turns, undecodable bytes, random points and random bytes),
    from limbs (projective representations, small-order and mixed-order points), device-pointer flavour, batch sizes on both sides of the
    one-point-per-wavefront marshal"""
    import torch
    from test_oracle_golden import _point_check_inputs
    encs = _point_check_inputs()
    rng = np.random.default_rng(15)
    encs += [rng.integers(0, 256, 32, dtype=np.uint8).tobytes() for _ in range(3000)]
    want = np.array([oracle.point_checks(e) & 3 for e in encs], dtype=np.uint8)
    arr = np.frombuffer(b"".join(encs), dtype=np.uint8).reshape(-1, 32)
    got = engine.point_checks(enc=arr)
    assert np.array_equal(got, want)
    assert {0, 1, 2, 3} <= set(want.tolist())                       # every combination occurs: e.g. y = p (small order, not canonical)
    assert np.array_equal(engine.point_checks(enc=arr[:1]), want[:1])
    # limbs: decoded points, times a random Z (projective), plus sums with torsion points
    exts = []
    for e in encs[:700]:
        x, ok = oracle.decode(e)
        if ok:
            exts.append(x)
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    exts += [oracle.add(exts[10 + i], w) for i, w in enumerate(weak)]      # mixed order: not small
    exts += [oracle.add(w, weak[(i + 2) % 5]) for i, w in enumerate(weak)]   # sums of torsion points: small
    exts = np.stack(exts)
    want_x = np.array([oracle.point_checks_ext(x) for x in exts], dtype=np.uint8)
    assert np.array_equal(engine.point_checks(pts_ext=exts), want_x)
    assert np.array_equal(engine.point_checks(pts_ext=exts[:3]), want_x[:3])
    assert (want_x & 2).any() and not (want_x & 2).all()
    d_enc = torch.from_numpy(arr.copy()).cuda()
    d_ext = torch.from_numpy(exts.copy()).cuda()
    f1 = torch.zeros(arr.shape[0], dtype=torch.uint8, device="cuda")
    f2 = torch.zeros(exts.shape[0], dtype=torch.uint8, device="cuda")
    engine.point_checks_dev(f1, enc=d_enc)
    engine.point_checks_dev(f2, pts_ext=d_ext)
    engine.sync()
    assert np.array_equal(f1.cpu().numpy(), want) and np.array_equal(f2.cpu().numpy(), want_x)
    flags = np.zeros(1, dtype=np.uint8)
    assert engine.lib.kyb_point_checks_batch(None, None, 1, flags.ctypes.data) < 0                               # neither operand
    assert engine.lib.kyb_point_checks_batch(arr.ctypes.data, exts.ctypes.data, 1, flags.ctypes.data) < 0      # both


def test_sign_random_and_golden(xengine, oracle):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    n = 300
    x, k = synth.scalars(n, 6, b"x"), synth.scalars(n, 6, b"k")
    msgs = synth.messages(n, 6)
    assert np.array_equal(engine.schnorr_sign(x, k, msgs), oracle.schnorr_sign_batch(x, k, msgs, nthreads=8))
    # the 1024 golden EdDSA lines as (x, k, msg) triples: bit-exact signature bytes, ragged messages 0..1023 B
    xs, ks, ms, sigs = [], [], [], []
    for ln in gzip.open(os.path.join(HERE, "golden", "sign.input.gz"), "rt").read().split("\n"):
        if not ln:
            continue
        p = ln.split(":")
        seed, msg, sig = bytes.fromhex(p[0])[:32], bytes.fromhex(p[2]), bytes.fromhex(p[3])[:64]
        d = bytearray(hashlib.sha512(seed).digest())
        d[0] &= 0xF8; d[31] &= 0x7F; d[31] |= 0x40
        r = int.from_bytes(hashlib.sha512(bytes(d[32:]) + msg).digest(), "little") % synth.L
        xs.append(bytes(d[:32])); ks.append(r.to_bytes(32, "little")); ms.append(msg); sigs.append(sig)
    got = engine.schnorr_sign(np.frombuffer(b"".join(xs), dtype=np.uint8), np.frombuffer(b"".join(ks), dtype=np.uint8), ms)
    assert rows(got) == sigs
    # the fused single-kernel signer (k_sign: both multiplications, inversions, hash in one kernel) stays selectable
    engine.set_option("finish.batched", 0)
    try:
        got = engine.schnorr_sign(np.frombuffer(b"".join(xs), dtype=np.uint8), np.frombuffer(b"".join(ks), dtype=np.uint8), ms)
        assert rows(got) == sigs
        assert np.array_equal(engine.schnorr_sign(x[:3], k[:3], msgs[:3]), oracle.schnorr_sign_batch(x[:3], k[:3], msgs[:3]))
    finally:
        engine.set_option("finish.batched", 1)
    for m in (1, 2, 31, 33):                              # tiny batches through the default pipeline (one launch for k*B and x*B)
        assert np.array_equal(engine.schnorr_sign(x[:m], k[:m], msgs[:m]), oracle.schnorr_sign_batch(x[:m], k[:m], msgs[:m]))


def test_full_size_properties_2_20(engine, oracle):
    """BASELINE configs 2 and 3 at N = 2^20 through the device-pointer API."""
    import torch
    n = 1 << 20
    rng = np.random.default_rng(20)
    s_np = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    s_np[:, 31] &= 0x0F                                   # < 2^252: valid canonical scalars
    dev = torch.device("cuda:0")
    s = torch.from_numpy(s_np).to(dev)
    enc_base = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    ext_base = torch.empty((n, 40), dtype=torch.int32, device=dev)
    engine.mul_base_dev(s, out_enc=enc_base, out_ext=ext_base)
    # (1) variable base on P = B equals fixed base (ties the un-KAT'd routine to the KAT-pinned one)
    bext = torch.from_numpy(np.tile(oracle.base(), (n, 1))).to(dev)
    enc_var = torch.empty_like(enc_base)
    engine.mul_dev(s, pts_ext=bext, out_enc=enc_var)
    engine.sync()
    assert torch.equal(enc_base, enc_var)
    # (2) DH commutativity: t * (s * B) == s * (t * B)
    t_np = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    t_np[:, 31] &= 0x0F
    t = torch.from_numpy(t_np).to(dev)
    ext_t = torch.empty_like(ext_base)
    engine.mul_base_dev(t, out_ext=ext_t)
    lhs, rhs = torch.empty_like(enc_base), torch.empty_like(enc_base)
    engine.mul_dev(t, pts_ext=ext_base, out_enc=lhs)
    engine.mul_dev(s, pts_ext=ext_t, out_enc=rhs)
    engine.sync()
    assert torch.equal(lhs, rhs)
    # (3) SURVEY 8(d): FULL comparison at N = 2^20 against the multithreaded oracle, fixed and variable base
    threads = min(16, len(os.sched_getaffinity(0)))
    assert np.array_equal(enc_base.cpu().numpy(), oracle.mul_base_batch(s_np, nthreads=threads))
    assert np.array_equal(lhs.cpu().numpy(), oracle.mul_batch(t_np, ext_base.cpu().numpy(), nthreads=threads))
    # (4) checksum of checksums is stable across a repeat run (determinism, no cross-lane leakage)
    enc2 = torch.empty_like(enc_base)
    engine.mul_dev(t, pts_ext=ext_base, out_enc=enc2)
    engine.sync()
    assert hashlib.sha256(enc2.cpu().numpy().tobytes()).digest() == hashlib.sha256(lhs.cpu().numpy().tobytes()).digest()


def test_base_table_matches_oracle(engine, oracle):
    """the GPU-built table image: entry (pos, j) = (j+1) * 16^pos * B, affine (y+x, y-x, 2dxy)"""
    P = 2**255 - 19
    d = (-121665 * pow(121666, P - 2, P)) % P
    img = np.frombuffer(engine.base_table().tobytes(), dtype=np.uint32)
    bits = [26, 25] * 5

    def val(limbs):
        v, off = 0, 0
        for l, b in zip(limbs, bits):
            v += int(l) << off
            off += b
        return v

    def idx(pos, j, k):
        return ((pos * 8 + (k >> 2)) * 8 + j) * 4 + (k & 3)

    for pos in (0, 1, 7, 31, 62, 63):
        for j in range(8):
            enc = oracle.mul_base(((j + 1) << (4 * pos)).to_bytes(32, "little"))
            v = int.from_bytes(enc, "little")
            y, sign = v & (2**255 - 1), v >> 255
            ypx = val([img[idx(pos, j, k)] for k in range(10)])
            ymx = val([img[idx(pos, j, 10 + k)] for k in range(10)])
            xy2d = val([img[idx(pos, j, 20 + k)] for k in range(10)])
            x = (ypx - ymx) * pow(2, P - 2, P) % P
            assert (ypx + ymx) * pow(2, P - 2, P) % P == y and (x & 1) == sign
            assert xy2d == 2 * d * x * y % P


@pytest.mark.parametrize("block", [256, 512])
def test_split_finish_and_block_variants(xengine, oracle, block):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    """batched-inversion finish (Montgomery trick over 8 items per lane) and the 512-thread fixed-base
    kernel give the same bytes as the fused path / the oracle, including ragged tails"""
    engine.set_option("finish.min_items", 1)
    engine.set_option("mul_base.block", block)
    engine.set_option("mul_base.radix", 16)
    try:
        for n in (1, 7, 8, 9, 1000, 2051):
            s = np.concatenate([synth.scalars(n - n // 3, 31), synth.raw256(n // 3, 31)])
            engine.set_option("finish.batched", 1)
            got, ext = engine.mul_base(s, want_ext=True)
            assert np.array_equal(got, oracle.mul_base_batch(s, nthreads=8))
            pts = rand_points_ext(oracle, n, 32)
            for algo in (0, 1):
                engine.set_option("mul.algo", algo)
                assert np.array_equal(engine.mul(s, pts_ext=pts), oracle.mul_batch(s, pts, nthreads=8))
            engine.set_option("finish.batched", 0)
            got0, ext0 = engine.mul_base(s, want_ext=True)
            assert np.array_equal(got, got0) and np.array_equal(ext, ext0)
        n = 600
        x, k = synth.scalars(n, 33, b"x"), synth.raw256(n, 33, b"k")
        msgs = [bytes([i & 255]) * (i % 97) for i in range(n)]
        engine.set_option("finish.batched", 1)
        assert np.array_equal(engine.schnorr_sign(x, k, msgs), oracle.schnorr_sign_batch(x, k, msgs, nthreads=8))
    finally:
        engine.set_option("finish.batched", 1)
        engine.set_option("finish.min_items", 1)
        engine.set_option("mul_base.block", 256)
        engine.set_option("mul_base.radix", 64)


def test_split_finish_isolates_degenerate_z(xengine, oracle):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    """an invalid extended input whose result has Z = 0 must not disturb the items that share its
    batched inversion, and must give the same bytes as the per-item path"""
    n = 64
    s = synth.scalars(n, 41)
    pts = rand_points_ext(oracle, n, 41)
    pts[5] = 0          # X = Y = Z = T = 0: not a curve point; every formula output stays 0
    pts[22] = 0
    engine.set_option("finish.min_items", 1)
    engine.set_option("mul.algo", 0)
    try:
        engine.set_option("finish.batched", 1)
        a = engine.mul(s, pts_ext=pts)
        engine.set_option("finish.batched", 0)
        b = engine.mul(s, pts_ext=pts)
    finally:
        engine.set_option("finish.batched", 1)
        engine.set_option("finish.min_items", 1)
        engine.set_option("mul.algo", 1)
    assert np.array_equal(a, b)
    want = oracle.mul_batch(s, pts, nthreads=8)
    good = [i for i in range(n) if i not in (5, 22)]
    assert np.array_equal(a[good], want[good])
    assert np.array_equal(a[[5, 22]], want[[5, 22]])     # reference: 0^(p-2) = 0 -> all-zero encoding
    # the table-free path must isolate the bad items too (what it returns for them is unspecified)
    c = engine.mul(s, pts_ext=pts)
    assert np.array_equal(c[good], want[good])


def test_verify_matches_oracle(xengine, oracle):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    """kyb_verify_batch: every status code of eddsa::/schnorr::verify_with_checks, both check orders,
    on golden signatures, the reference's negative vectors and corrupted inputs"""
    from test_device_source_on_host import verify_cases
    cases = verify_cases(oracle)
    pubs = np.frombuffer(b"".join(c[0] for c in cases), dtype=np.uint8)
    sigs = np.frombuffer(b"".join(c[2] for c in cases), dtype=np.uint8)
    msgs = [c[1] for c in cases]
    for flavor in (0, 1):
        want = np.array([oracle.verify(flavor, *c) for c in cases], dtype=np.uint8)
        assert set(want.tolist()) >= {0, 2, 3, 4, 5, 6, 7, 8, 9}      # every reject reason is exercised
        for overlap, by_enc in ((1, 1), (0, 1), (1, 0), (0, 0)):       # s*B on the side stream / in line; equation on encodings / on points
            engine.set_option("verify.overlap", overlap)
            engine.set_option("verify.by_encoding", by_enc)
            try:
                for _ in range(3):                                     # back-to-back calls reuse the scratch of the one before
                    assert np.array_equal(engine.verify(pubs, msgs, sigs, flavor), want)
            finally:
                engine.set_option("verify.overlap", 1)
                engine.set_option("verify.by_encoding", 1)
    # the same cases through the batch kernels of DKG-sized calls (small-batch kernels off): with ladder.y_only the key's decode runs beside the
    # two-lane ladder, which starts on the y of the key bytes (k_verify_hash / k_mul_ladder_pair_y / k_ladder_recover, round 4); 0 = decode first
    saved = {k: engine.get_option(k) for k in ("coop.max_items", "coop.base_max_items", "coop.verify_max_items", "ladder.y_only")}
    try:
        for k in ("coop.max_items", "coop.base_max_items", "coop.verify_max_items"):
            engine.set_option(k, 0)
        for y_only in (2, 1, 0):
            engine.set_option("ladder.y_only", y_only)
            for flavor in (0, 1):
                want = np.array([oracle.verify(flavor, *c) for c in cases], dtype=np.uint8)
                for _ in range(2):
                    assert np.array_equal(engine.verify(pubs, msgs, sigs, flavor), want), (y_only, flavor)
    finally:
        for k, v in saved.items():
            engine.set_option(k, v)
    # all 1024 golden signatures verify; their messages are 0..1023 bytes long
    ps, ms, ss = [], [], []
    for ln in gzip.open(os.path.join(HERE, "golden", "sign.input.gz"), "rt").read().split("\n"):
        if ln:
            p = ln.split(":")
            ps.append(bytes.fromhex(p[1])); ms.append(bytes.fromhex(p[2])); ss.append(bytes.fromhex(p[3])[:64])
    st = engine.verify(np.frombuffer(b"".join(ps), dtype=np.uint8), ms, np.frombuffer(b"".join(ss), dtype=np.uint8), 0)
    assert not st.any()


def test_sign_then_verify_round_trip_2_16(engine, oracle):
    """size-independent property at a large size: every GPU signature verifies on the GPU; flipping one
    bit of each makes every one fail"""
    n = 1 << 16
    rng = np.random.default_rng(77)
    x = rng.integers(0, 256, (n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8); k[:, 31] &= 0x0F
    msgs = synth.messages(n, 77)
    sig = engine.schnorr_sign(x, k, msgs)
    pub = engine.mul_base(x)
    assert not engine.verify(pub, msgs, sig, 1).any()
    bad = sig.copy()
    bad[np.arange(n), rng.integers(32, 63, n)] ^= 1        # corrupt s (below the top byte: stays < L mostly)
    st = engine.verify(pub, msgs, bad, 1)
    assert (st != 0).all()
    idx = rng.choice(n, 64, replace=False)
    assert [oracle.verify(1, bytes(pub[i]), msgs[i], bytes(bad[i])) for i in idx] == st[idx].tolist()


def test_pubpoly_eval_and_equal(engine, oracle):
    """kyb_pubpoly_eval_batch == PubPoly::eval (poly.rs:457-469) for every index; kyb_equal_batch == Point::eq"""
    rng = np.random.default_rng(50)
    t = 7
    commits = oracle.mul_base_ext_batch(synth.scalars(t, 50))
    idx = np.concatenate([np.arange(0, 300), rng.integers(0, 2**32 - 2, 40), [2**32 - 2]]).astype(np.uint32)
    enc, ext = engine.pubpoly_eval(commits, idx, want_ext=True)
    sample = list(range(0, 300, 13)) + list(range(300, len(idx)))
    for i in sample:
        assert bytes(enc[i]) == oracle.pubpoly_eval(commits, int(idx[i])), int(idx[i])
        assert oracle.encode(ext[i]) == bytes(enc[i])
    # Point::eq on different projective representatives of equal / unequal points
    a = oracle.mul_base_ext_batch(synth.scalars(200, 51))
    b = np.stack([oracle.add(oracle.add(p, q), q, sub=True) for p, q in zip(a, np.roll(a, 1, axis=0))])   # (p + q) - q == p, other limbs
    eq = engine.equal(a, b)
    assert eq.all()
    assert not engine.equal(a, np.roll(a, 1, axis=0)).any()
    # records with Z = 0 (Point::default(): all-zero limbs) follow the reference's encode-and-compare: they encode as x = y = 0
    zero = np.zeros(40, dtype=np.int32)
    z_only = zero.copy(); z_only[0] = 5; z_only[10] = 7              # X, Y non-zero, Z = 0: still encodes as (0, 0)
    xy0 = zero.copy(); xy0[20] = 3                                   # X = Y = 0, Z = 3: affine (0, 0)
    pairs = [(zero, zero), (zero, z_only), (zero, xy0), (xy0, z_only), (zero, a[0]), (a[0], z_only), (xy0, a[0]), (a[0], a[0])]
    pa, pb = np.stack([x for x, _ in pairs]), np.stack([y for _, y in pairs])
    want_eq = [oracle.encode(x) == oracle.encode(y) for x, y in pairs]
    assert want_eq == [True, True, True, True, False, False, False, True]
    assert engine.equal(pa, pb).astype(bool).tolist() == want_eq
    # check (poly.rs:526-530): eval(i) == s_i * B for the matching private polynomial
    coeffs = [int.from_bytes(bytes(c), "little") for c in synth.scalars(t, 50)]
    shares = [sum(c * pow(int(i) + 1, j, synth.L) for j, c in enumerate(coeffs)) % synth.L for i in idx[:64]]
    want = engine.mul_base(np.frombuffer(b"".join(s.to_bytes(32, "little") for s in shares), dtype=np.uint8))
    assert np.array_equal(enc[:64], want)


@pytest.mark.parametrize("waves", [2, 4])
def test_ladder_path_matches_oracle(xengine, oracle, waves):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    """mul.algo=1 (Montgomery ladder + y-recovery, no per-lane table): same bytes as the oracle on quirk
    vectors, mixed-order points, scalars around multiples of L, encoded inputs incl. invalid ones, ragged sizes"""
    engine.set_option("mul.algo", 1)
    engine.set_option("mul.ladder_waves", waves)
    try:
        q = [v for v in KATS["quirk_mul"] if v["ok"]]
        sc = np.frombuffer(b"".join(bytes.fromhex(v["scalar"]) for v in q), dtype=np.uint8)
        pe = np.frombuffer(b"".join(bytes.fromhex(v["point"]) for v in q), dtype=np.uint8)
        got, ok = engine.mul(sc, pts_enc=pe, want_ok=True)
        assert ok.all() and [bytes(r).hex() for r in got] == [v["out"] for v in q]
        ext = np.stack([oracle.decode(bytes.fromhex(v["point"]))[0] for v in q])
        assert [bytes(r).hex() for r in engine.mul(sc, pts_ext=ext)] == [v["out"] for v in q]
        # random + mixed-order points, raw 256-bit scalars, scalars k*L + {-1,0,1}
        n = 1500
        rng = np.random.default_rng(61)
        s = np.concatenate([synth.scalars(600, 61), synth.raw256(n - 600, 61)])
        Lq = synth.L
        for j, k in enumerate([1, 2, 4, 8]):
            for d in (-1, 0, 1):
                s[3 * j + d + 1] = np.frombuffer(((k * Lq + d) % 2**256).to_bytes(32, "little"), dtype=np.uint8)
        pts = rand_points_ext(oracle, n, 61)
        weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
        for i in range(0, n, 3):
            pts[i] = oracle.add(pts[i], weak[int(rng.integers(0, 5))])
        got, gext = engine.mul(s, pts_ext=pts, want_ext=True)
        assert np.array_equal(got, oracle.mul_batch(s, pts, nthreads=8))
        for i in range(0, n, 131):
            assert oracle.encode(gext[i]) == bytes(got[i])
        # encoded inputs with invalid encodings in between
        good = [oracle.encode(p) for p in pts[:30]]
        bad = [bytes.fromhex(h) for h in KATS["invalid_encodings"]]
        encs = good[:11] + bad + good[11:]
        s2 = synth.scalars(len(encs), 62)
        got, ok = engine.mul(s2, pts_enc=np.frombuffer(b"".join(encs), dtype=np.uint8), want_ok=True)
        for i, e in enumerate(encs):
            pe_, okk = oracle.decode(e)
            assert ok[i] == okk
            assert bytes(got[i]) == (oracle.mul(bytes(s2[i]), pe_) if okk else IDENT)
        for m in (1, 7, 8, 9, 63, 65, 257):
            sm = synth.scalars(m, 63)
            pm = rand_points_ext(oracle, m, 63)
            assert np.array_equal(engine.mul(sm, pts_ext=pm), oracle.mul_batch(sm, pm, nthreads=8))
    finally:
        engine.set_option("mul.ladder_waves", 3)


def test_two_lane_ladder_matches_one_lane_and_oracle(xengine, oracle):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    """k_mul_ladder_pair (two lanes per item, ge_ladder_pair.h; launches of at most ladder.pair_max_items items) == k_mul_ladder_quad (four lanes,
    ge_ladder_quad.h; variable base from points, at most ladder.quad_max_items items) == k_mul_ladder == the oracle:
    quirk vectors, mixed-order points, scalars around multiples of L, canonical-only batches (252 steps) and batches with one unreduced
    scalar (256), invalid encodings, ragged and odd sizes, shared operands (linear combinations) and the h*A of a verification"""
    saved = {k: engine.get_option(k) for k in ("coop.max_items", "coop.base_max_items", "coop.verify_max_items", "ladder.pair_max_items", "ladder.quad_max_items")}
    try:
        for k in ("coop.max_items", "coop.base_max_items", "coop.verify_max_items"):
            engine.set_option(k, 0)                                       # small batches reach the ladder kernels
        q = [v for v in KATS["quirk_mul"] if v["ok"]]
        sc = np.frombuffer(b"".join(bytes.fromhex(v["scalar"]) for v in q), dtype=np.uint8)
        pe = np.frombuffer(b"".join(bytes.fromhex(v["point"]) for v in q), dtype=np.uint8)
        n = 1501
        rng = np.random.default_rng(611)
        s = np.concatenate([synth.scalars(600, 611), synth.raw256(n - 600, 611)])
        for j, k in enumerate([1, 2, 4, 8]):
            for d in (-1, 0, 1):
                s[3 * j + d + 1] = np.frombuffer(((k * synth.L + d) % 2**256).to_bytes(32, "little"), dtype=np.uint8)
        pts = rand_points_ext(oracle, n, 611)
        weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
        for i in range(0, n, 3):
            pts[i] = oracle.add(pts[i], weak[int(rng.integers(0, 5))])
        pts[5], pts[6] = weak[0], oracle.null()
        want = oracle.mul_batch(s, pts, nthreads=8)
        canon = synth.scalars(777, 612)                                   # every scalar below 2^252: the launch skips four steps
        want_canon = oracle.mul_batch(canon, pts[:777], nthreads=8)
        bad = [bytes.fromhex(h) for h in KATS["invalid_encodings"]]
        good = [oracle.encode(p) for p in pts[:30]]
        encs = good[:11] + bad + good[11:]
        s2 = synth.scalars(len(encs), 613)
        # verification and linear combinations run the same launch
        x = synth.scalars(700, 614); kk = synth.scalars(700, 615, b"k"); msgs = synth.messages(700, 616)
        sigs = oracle.schnorr_sign_batch(x, kk, msgs, nthreads=8)
        pubs = oracle.mul_base_batch(x, nthreads=8)
        sigs[::5, 40] ^= 4
        want_st = oracle.verify_batch(1, pubs, msgs, sigs, nthreads=8)
        lsc = synth.scalars(40 * 9, 617).reshape(40, 9, 32)
        lp = pts[:9]
        results = {}
        for pair_max in (0, 1 << 20, 1 << 21):                           # one lane | two lanes | four lanes where that kernel applies (points given as limbs), two elsewhere
            engine.set_option("ladder.pair_max_items", min(pair_max, 1 << 20))
            engine.set_option("ladder.quad_max_items", (1 << 20) if pair_max == 1 << 21 else 0)
            got, ok = engine.mul(sc, pts_enc=pe, want_ok=True)
            assert ok.all() and [bytes(r).hex() for r in got] == [v["out"] for v in q], pair_max
            g1, gext = engine.mul(s, pts_ext=pts, want_ext=True)
            assert np.array_equal(g1, want), pair_max
            assert np.array_equal(engine.mul(canon, pts_ext=pts[:777]), want_canon), pair_max
            got, ok = engine.mul(s2, pts_enc=np.frombuffer(b"".join(encs), dtype=np.uint8), want_ok=True)
            for i, e in enumerate(encs):
                pe_, okk = oracle.decode(e)
                assert ok[i] == okk and bytes(got[i]) == (oracle.mul(bytes(s2[i]), pe_) if okk else IDENT), (pair_max, i)
            for m in (1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 129, 255, 257):
                assert np.array_equal(engine.mul(s[:m], pts_ext=pts[:m]), want[:m]), (pair_max, m)
            assert np.array_equal(engine.verify(pubs, msgs, sigs, 1), want_st), pair_max
            lc_shared = engine.lincomb(lsc, pts_ext=lp)
            lc_own = engine.lincomb(lsc, pts_ext=pts[:360].reshape(40, 9, 40))
            for g_ in (0, 17, 39):
                assert bytes(lc_shared[g_]) == oracle.lincomb(lsc[g_], lp) and bytes(lc_own[g_]) == oracle.lincomb(lsc[g_], pts[9 * g_:9 * g_ + 9]), (pair_max, g_)
            results[pair_max] = (g1, gext, lc_shared, lc_own)
        for a_, b_ in zip(results[0], results[1 << 20]):
            assert np.array_equal(a_, b_)                                 # limbs included: the same field operations on the same values
        for a_, b_ in zip(results[0], results[1 << 21]):
            assert np.array_equal(a_, b_)
        engine.set_option("ladder.quad_max_items", 1 << 20)
        engine.profile_begin(4)
        engine.mul(s[:100], pts_ext=pts[:100])
        assert "k_mul_ladder_pair" in [nm for nm, _ in engine.profile_read(4)]      # (the four-lane launch is profiled under the two-lane ladder's name)
        engine.profile_begin(0)
        engine.set_option("ladder.quad_max_items", 0)
        # from wire encodings the two-lane ladder runs on the y of the encoding while a side stream decodes x (ladder.y_only, round 4): the same
        # bytes with the option off (decode first), on the quirk points, on encodings that do not decode, on y = +-1 (x = 0) and on non-canonical y
        engine.set_option("ladder.pair_max_items", 1 << 20)
        P_ = 2**255 - 19
        special = [(1).to_bytes(32, "little"), (P_ - 1).to_bytes(32, "little"), (P_ + 1).to_bytes(32, "little"), bytes([1] + [0] * 30 + [0x80]), P_.to_bytes(32, "little")]
        encs3 = np.frombuffer(b"".join(encs + special + [oracle.encode(p) for p in pts[30:900]]), dtype=np.uint8).reshape(-1, 32)
        s3 = np.concatenate([synth.scalars(450, 618), synth.raw256(encs3.shape[0] - 450, 618)])
        want3 = oracle.mul_enc_batch(s3, encs3, nthreads=8)
        for y_only in (2, 1, 0):
            engine.set_option("ladder.y_only", y_only)
            got, ok = engine.mul(sc, pts_enc=pe, want_ok=True)
            assert ok.all() and [bytes(r).hex() for r in got] == [v["out"] for v in q], y_only
            got3, ok3 = engine.mul(s3, pts_enc=encs3, want_ok=True)
            assert np.array_equal(ok3, want3[1]) and np.array_equal(got3, want3[0]), y_only
        engine.set_option("ladder.y_only", 2)
    finally:
        for k, v in saved.items():
            engine.set_option(k, v)


def test_fixed_base_radix32_kernel(xengine, oracle):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    """the 43-window radix-64 kernel (1024-thread workgroups, the whole 160 KiB LDS as table) == the 52-window radix-32
    kernel (104 KiB table) == the radix-16 kernel == oracle, through mul_base, sign and verify; quirk scalars included"""
    engine.set_option("finish.min_items", 1)       # route even small batches through it
    try:
        qb = KATS["quirk_mul_base"]
        qs = np.frombuffer(b"".join(bytes.fromhex(q["scalar"]) for q in qb), dtype=np.uint8)
        for radix in (64, 32, 16):
            engine.set_option("mul_base.radix", radix)
            assert [bytes(r).hex() for r in engine.mul_base(qs)] == [q["out"] for q in qb]
            for n in (1, 1023, 1024, 1025, 3000):
                s = np.concatenate([synth.scalars(n - n // 2, 70 + n), synth.raw256(n // 2, 70 + n)])
                enc, ext = engine.mul_base(s, want_ext=True)
                assert np.array_equal(enc, oracle.mul_base_batch(s, nthreads=8))
                assert oracle.encode(ext[n // 2]) == bytes(enc[n // 2])
            x, k = synth.scalars(700, 71, b"x"), synth.raw256(700, 71, b"k")
            msgs = synth.messages(700, 71, length=17)
            sig = engine.schnorr_sign(x, k, msgs)
            assert np.array_equal(sig, oracle.schnorr_sign_batch(x, k, msgs, nthreads=8))
            # raw 256-bit nonces >= 2^255 lose their top digit in R = k*B (ge.rs:459) but not in s = k + x*h, so those
            # signatures are invalid in the reference too: compare the verdicts, do not expect all-valid
            pub = engine.mul_base(x)
            st = engine.verify(pub, msgs, sig, 1)
            assert np.array_equal(st, oracle.verify_batch(1, pub, msgs, sig, nthreads=8))
            assert set(st.tolist()) == {0, 9}
            kc = synth.scalars(700, 72, b"k")
            assert not engine.verify(pub, msgs, engine.schnorr_sign(x, kc, msgs), 1).any()
    finally:
        engine.set_option("mul_base.radix", 64)
        engine.set_option("finish.min_items", 1)
    # table image, radix-32 part: entry (pos, j) = (j+1) * 32^pos * B
    P = 2**255 - 19
    whole = np.frombuffer(engine.base_table().tobytes(), dtype=np.uint32)
    img = whole[65536 // 4:]
    bits = [26, 25] * 5

    def val(limbs):
        v, off = 0, 0
        for l, b in zip(limbs, bits):
            v += int(l) << off
            off += b
        return v

    def idx(pos, j, k):
        return ((pos * 8 + (k >> 2)) * 16 + j) * 4 + (k & 3)

    import bigint_model as M
    for pos, j in ((0, 0), (0, 15), (1, 7), (25, 3), (50, 15), (51, 0), (51, 1)):
        x, y = M.mul_int((j + 1) << (5 * pos), M.B)
        assert val([img[idx(pos, j, k)] for k in range(10)]) == (y + x) % P
        assert val([img[idx(pos, j, 10 + k)] for k in range(10)]) == (y - x) % P
        assert val([img[idx(pos, j, 20 + k)] for k in range(10)]) == 2 * M.D * x * y % P
    # radix-64 part: entry (pos, j) = (2j+1) * 64^pos * B (odd multiples: the recoding has no zero digit), 30 packed words per entry
    img64 = whole[172032 // 4:]
    assert img64.shape[0] == 163200 // 4

    def idx64(pos, j, k):
        E = 32 if pos < 42 else 16
        g, r = divmod(k, 10)                          # g: 0 ypx, 1 ymx, 2 xy2d, each in its own planes
        big, small = ((0, 16 * E), (8 * E, 18 * E), (20 * E, 28 * E))[g]
        inner = big + ((r >> 2) * E + j) * 4 + (r & 3) if r < 8 else small + j * 2 + (r - 8)
        return (pos * 960 if pos < 42 else 42 * 960) + inner

    for pos, j in ((0, 0), (0, 31), (1, 7), (20, 30), (41, 31), (42, 0), (42, 8), (42, 15)):
        x, y = M.mul_int((2 * j + 1) << (6 * pos), M.B)
        assert val([img64[idx64(pos, j, k)] for k in range(10)]) == (y + x) % P
        assert val([img64[idx64(pos, j, 10 + k)] for k in range(10)]) == (y - x) % P
        assert val([img64[idx64(pos, j, 20 + k)] for k in range(10)]) == 2 * M.D * x * y % P


def test_two_streams_and_two_threads(engine, oracle):
    """device-pointer calls on two different streams may overlap (each stream owns its staging buffers);
    host-pointer calls from two threads are serialised by the engine — both give oracle-exact results"""
    import threading
    import torch
    dev = torch.device("cuda:0")
    n = 20000
    s1, s2 = synth.scalars(n, 81), synth.scalars(n, 82)
    p1 = rand_points_ext(oracle, 300, 81)
    p2 = rand_points_ext(oracle, 300, 82)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)   # noqa: E731
    P1, P2 = t(np.tile(p1, (n // 300 + 1, 1))[:n]), t(np.tile(p2, (n // 300 + 1, 1))[:n])
    S1, S2 = t(s1), t(s2)
    O1 = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    O2 = torch.empty_like(O1)
    B1, B2 = torch.empty_like(O1), torch.empty_like(O1)
    st1, st2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for _ in range(3):                       # interleave launches on the two streams
        engine.mul_dev(S1, pts_ext=P1, out_enc=O1, stream=st1.cuda_stream)
        engine.mul_dev(S2, pts_ext=P2, out_enc=O2, stream=st2.cuda_stream)
        engine.mul_base_dev(S1, out_enc=B1, stream=st1.cuda_stream)
        engine.mul_base_dev(S2, out_enc=B2, stream=st2.cuda_stream)
    torch.cuda.synchronize()
    idx = np.arange(0, n, 37)
    P1n, P2n = P1.cpu().numpy(), P2.cpu().numpy()
    assert np.array_equal(O1.cpu().numpy()[idx], oracle.mul_batch(s1[idx], P1n[idx], nthreads=8))
    assert np.array_equal(O2.cpu().numpy()[idx], oracle.mul_batch(s2[idx], P2n[idx], nthreads=8))
    assert np.array_equal(B1.cpu().numpy()[idx], oracle.mul_base_batch(s1[idx], nthreads=8))
    assert np.array_equal(B2.cpu().numpy()[idx], oracle.mul_base_batch(s2[idx], nthreads=8))
    # two host threads through the host-pointer API
    res = {}

    def work(tag, s, p):
        res[tag] = (engine.mul(s[:3000], pts_ext=np.tile(p, (10, 1))), engine.mul_base(s[:5000]))

    th = [threading.Thread(target=work, args=(1, s1, p1)), threading.Thread(target=work, args=(2, s2, p2))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert np.array_equal(res[1][0], oracle.mul_batch(s1[:3000], np.tile(p1, (10, 1)), nthreads=8))
    assert np.array_equal(res[2][0], oracle.mul_batch(s2[:3000], np.tile(p2, (10, 1)), nthreads=8))
    assert np.array_equal(res[1][1], oracle.mul_base_batch(s1[:5000], nthreads=8))
    assert np.array_equal(res[2][1], oracle.mul_base_batch(s2[:5000], nthreads=8))


def test_eddsa_sign_golden_file_on_gpu(engine, oracle):
    """the reference's own golden test (tests/sign/eddsa.rs:36-94) end to end on the GPU: all 1024 (seed, msg)
    lines -> public key and signature bytes; plus the five RFC 8032 vectors"""
    seeds, pubs, msgs, sigs = [], [], [], []
    for ln in gzip.open(os.path.join(HERE, "golden", "sign.input.gz"), "rt").read().split("\n"):
        if ln:
            p = ln.split(":")
            seeds.append(bytes.fromhex(p[0])[:32]); pubs.append(bytes.fromhex(p[1])); msgs.append(bytes.fromhex(p[2])); sigs.append(bytes.fromhex(p[3])[:64])
    for v in KATS["rfc8032"]:
        seeds.append(bytes.fromhex(v["private"])); pubs.append(bytes.fromhex(v["public"])); msgs.append(bytes.fromhex(v["message"])); sigs.append(bytes.fromhex(v["signature"]))
    sig, pub = engine.eddsa_sign(np.frombuffer(b"".join(seeds), dtype=np.uint8), msgs, want_pub=True)
    assert rows(pub) == pubs
    assert rows(sig) == sigs
    assert not engine.verify(pub, msgs, sig, 0).any()
    # a large batch goes through the split pipeline: spot-check against the oracle
    n = 6000
    sd = synth.raw256(n, 95)
    ms = synth.messages(n, 95, length=24)
    sig2, pub2 = engine.eddsa_sign(sd, ms, want_pub=True)
    for i in range(0, n, 499):
        assert bytes(sig2[i]) == oracle.eddsa_sign(bytes(sd[i]), ms[i])
        assert bytes(pub2[i]) == oracle.eddsa_expand(bytes(sd[i]))[2]
    # EdDSA::sign on key objects that hold their public key (eddsa_sig.rs:132-137): same bytes, one mult less;
    # both batch regimes (fused / split), and the Schnorr flavour of the same
    assert rows(engine.eddsa_sign(np.frombuffer(b"".join(seeds), dtype=np.uint8), msgs, pubs=pub)) == sigs
    assert np.array_equal(engine.eddsa_sign(sd, ms, pubs=pub2), sig2)
    assert np.array_equal(engine.eddsa_sign(sd[:100], ms[:100], pubs=pub2[:100]), sig2[:100])
    x = synth.scalars(n, 96); k = synth.scalars(n, 97)
    xpub = engine.mul_base(x)
    want = engine.schnorr_sign(x, k, ms)
    assert np.array_equal(engine.schnorr_sign(x, k, ms, pubs=xpub), want)
    assert np.array_equal(engine.schnorr_sign(x[:64], k[:64], ms[:64], pubs=xpub[:64]), want[:64])
    # a key object with a wrong public key signs something that does not verify (as in the reference)
    wrong = engine.schnorr_sign(x[:8], k[:8], ms[:8], pubs=xpub[1:9])
    assert (engine.verify(xpub[:8], ms[:8], wrong, 1) == 9).all()


def test_lincomb_matches_oracle(engine, oracle):
    """kyb_lincomb_batch == recover_commit's accumulation (poly.rs:579-600) per group: own points per group,
    shared points (recover_pub_poly shape), encoded inputs with an invalid one, group lengths that are not powers
    of two, one long group, t = 1"""
    rng = np.random.default_rng(60)
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    for m, t in ((1, 1), (5, 1), (3, 2), (7, 3), (4, 13), (33, 8), (2, 100)):
        sc = synth.scalars(m * t, 60 + t).reshape(m, t, 32).copy()
        pts = oracle.mul_base_ext_batch(synth.scalars(m * t, 160 + t)).reshape(m, t, 40).copy()
        if t >= 3:
            pts[0, 1] = pts[0, 0]; sc[0, 1] = sc[0, 0]
            pts[0, 2] = oracle.add(pts[0, 2], weak[3])
            sc[m - 1, 0] = 0
            sc[m - 1, 1] = np.frombuffer(bytes([255] * 32), dtype=np.uint8)
        if t >= 8:
            pts[1, 3] = oracle.neg(pts[1, 4]); sc[1, 3] = sc[1, 4]
            pts[1, 5] = oracle.null(); pts[1, 6] = weak[2]
        enc, ext = engine.lincomb(sc, pts_ext=pts, want_ext=True)
        for g in range(m):
            want = oracle.lincomb(sc[g], pts[g])
            assert bytes(enc[g]) == want, (m, t, g)
            assert oracle.encode(ext[g]) == want
        # shared points: every group uses the points of group 0
        enc_s = engine.lincomb(sc, pts_ext=pts[0])
        for g in range(m):
            assert bytes(enc_s[g]) == oracle.lincomb(sc[g], pts[0]), (m, t, g, "shared")
    # encoded operands; one encoding does not decode -> counts as the neutral element, ok flag 0
    m, t = 6, 5
    sc = synth.scalars(m * t, 77).reshape(m, t, 32)
    pts = oracle.mul_base_ext_batch(synth.scalars(m * t, 78)).reshape(m, t, 40).copy()
    pe = np.stack([np.frombuffer(oracle.encode(p), dtype=np.uint8) for p in pts.reshape(-1, 40)]).reshape(m, t, 32).copy()
    bad = next(bytes([v]) + bytes(31) for v in range(2, 50) if not oracle.decode(bytes([v]) + bytes(31))[1])
    pe[2, 3] = np.frombuffer(bad, dtype=np.uint8)
    pts[2, 3] = oracle.null()
    enc, ok = engine.lincomb(sc, pts_enc=pe, want_ok=True)
    assert ok.sum() == m * t - 1 and ok.reshape(m, t)[2, 3] == 0
    for g in range(m):
        assert bytes(enc[g]) == oracle.lincomb(sc[g], pts[g])
    # Lagrange: recover the secret commitment of a threshold-6 polynomial from 6 of its public shares, 64 polynomials at once
    t, m = 6, 64
    xs = [i + 1 for i in (0, 2, 3, 5, 8, 9)]
    lam = []
    for xi in xs:
        num = den = 1
        for xj in xs:
            if xj != xi:
                num = num * xj % synth.L
                den = den * (xj - xi) % synth.L
        lam.append(num * pow(den, synth.L - 2, synth.L) % synth.L)
    lam_b = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in lam), dtype=np.uint8).reshape(t, 32)
    coeffs = [[int.from_bytes(bytes(c), "little") for c in synth.scalars(t, 900 + g)] for g in range(m)]
    share_sc = b"".join((sum(c * pow(x, j, synth.L) for j, c in enumerate(cg)) % synth.L).to_bytes(32, "little") for cg in coeffs for x in xs)
    _, share_pts = engine.mul_base(np.frombuffer(share_sc, dtype=np.uint8), want_ext=True)
    rec = engine.lincomb(np.broadcast_to(lam_b, (m, t, 32)), pts_ext=share_pts.reshape(m, t, 40))
    want = engine.mul_base(np.frombuffer(b"".join(cg[0].to_bytes(32, "little") for cg in coeffs), dtype=np.uint8))
    assert np.array_equal(rec, want)


def test_cfg5_shard_2_21_linearity(engine, oracle):
    """BASELINE config 5 hands every GPU 2^21 items.  One such shard through the device-pointer API, checked by a
    size-independent property on ALL items - linearity, s*P + t*P == (s + t)*P with the sum formed mod L on the
    host - plus an oracle-checked sample and the largest index's neighbours (grid tail)."""
    import torch
    n = (1 << 21) + 77                                   # not a multiple of any block size
    rng = np.random.default_rng(21)
    limbs = rng.integers(0, 1 << 63, (n, 4), dtype=np.uint64)
    limbs[:, 3] &= (1 << 59) - 1                         # < 2^251 so that s + t < L without reduction
    s_np = limbs.view(np.uint8).reshape(n, 32)
    limbs_t = rng.integers(0, 1 << 63, (n, 4), dtype=np.uint64)
    limbs_t[:, 3] &= (1 << 59) - 1
    t_np = limbs_t.view(np.uint8).reshape(n, 32)
    dev = torch.device("cuda:0")
    s = torch.from_numpy(s_np.copy()).to(dev)
    t = torch.from_numpy(t_np.copy()).to(dev)
    # s + t as a 256-bit integer (< 2^252 < L, no reduction needed), formed with carries over 32-bit words
    a32 = s_np.view(np.uint32).astype(np.uint64)
    b32 = t_np.view(np.uint32).astype(np.uint64)
    c = np.zeros(n, dtype=np.uint64)
    st32 = np.empty((n, 8), dtype=np.uint32)
    for j in range(8):
        v = a32[:, j] + b32[:, j] + c
        st32[:, j] = (v & 0xFFFFFFFF).astype(np.uint32)
        c = v >> np.uint64(32)
    assert not c.any()
    st_np = st32.view(np.uint8).reshape(n, 32)
    st = torch.from_numpy(st_np).to(dev)
    pts = torch.empty((n, 40), dtype=torch.int32, device=dev)
    engine.mul_base_dev(t, out_ext=pts)                  # P_i = t_i * B
    sp = torch.empty((n, 40), dtype=torch.int32, device=dev)
    tp = torch.empty((n, 40), dtype=torch.int32, device=dev)
    stp = torch.empty((n, 40), dtype=torch.int32, device=dev)
    engine.mul_dev(s, pts_ext=pts, out_ext=sp)
    engine.mul_dev(t, pts_ext=pts, out_ext=tp)
    engine.mul_dev(st, pts_ext=pts, out_ext=stp)
    summed = torch.empty_like(sp)
    eq = torch.empty((n,), dtype=torch.uint8, device=dev)
    engine.add_dev(sp, tp, summed)
    engine.equal_dev(summed, stp, eq)
    engine.sync()
    assert int(eq.sum().item()) == n
    idx = np.concatenate([rng.choice(n, 1024, replace=False), [0, n - 1, n - 2, (1 << 21) - 1, 1 << 21]])
    want = oracle.mul_batch(s_np[idx], pts[idx].cpu().numpy(), nthreads=8)
    got = engine.encode(sp[idx].cpu().numpy())
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [5, 700, 3000, 9000])
def test_verification_with_the_public_keys_given_as_points(engine, oracle, n):
    """kyb_verify_points_batch (schnorr::verify / eddsa::verify take &Point and marshal it, schnorr_sig.rs:114-127): status == kyb_verify_batch on
    marshal_binary(point) == the oracle on those bytes, both check orders — valid and corrupted signatures, small-order keys, projective
    representations (Z != 1), and limbs that are NOT a point of the curve (random limbs, Z = 0), where the bytes decide as in the reference"""
    x = synth.scalars(n, 1200 + n); x[:, 31] &= 0x7f
    k = synth.scalars(n, 1300 + n, b"k")
    msg_list = synth.messages(n, 1400 + n)
    sigs = oracle.schnorr_sign_batch(x, k, msg_list, nthreads=8)
    pts = oracle.mul_base_ext_batch(x)
    rng = np.random.default_rng(n)
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    for j, w in enumerate(weak):
        if 3 + 7 * j < n:
            pts[3 + 7 * j] = w                                              # small-order public keys
    if n >= 64:
        engine.set_option("ext.projective", 1)                             # the same keys as (X : Y : Z : T) with Z != 1
        try:
            proj = engine.mul_base(x[40:60], ext_only=True)
        finally:
            engine.set_option("ext.projective", 0)
        assert (proj[:, 20:30] != pts[40:60, 20:30]).any()
        pts[40:60] = proj
    if n >= 700:
        pts[100:110] = rng.integers(-(1 << 24), 1 << 24, (10, 40), dtype=np.int64).astype(np.int32)      # not points at all
        pts[111, 20:30] = 0                                                 # Z = 0
        pts[113, 30:40] = pts[114, 30:40]                                   # T inconsistent with X Y / Z
    bad = sigs.copy()
    bad[::4, 35] ^= 0x10
    bad[2::9, 0] ^= 1
    bad[5::13, 63] |= 0xf0                                                  # s not canonical
    enc = engine.encode(pts)
    for i in sorted({0, 3, min(10, n - 1), n - 1} | ({100, 105, 111, 113} if n >= 700 else set())):
        assert bytes(enc[i]) == oracle.encode(pts[i])
    for flavor in (0, 1):
        want = engine.verify(enc, msg_list, bad, flavor)
        assert np.array_equal(want, oracle.verify_batch(flavor, enc, msg_list, bad, nthreads=8))
        got = engine.verify_points(pts, msg_list, bad, flavor)
        assert np.array_equal(got, want), (flavor, np.nonzero(got != want)[0][:10])
    assert (engine.verify_points(pts, msg_list, sigs, 1) == 0).sum() >= n - 40      # the untouched keys verify their signatures


def test_mid_size_host_calls_agree_whichever_way_the_arrays_travel(engine, oracle):
    """A DKG-sized host-pointer call takes the page-locked zero-copy window (host.zero_copy_kib, 4 MiB by default) or, above it, copies on
    the engine stream: same bytes both ways, for the record batches (mul, mul_base) and the calls with mixed arrays (sign, verify, dealer
    shares, linear combinations) — and the option is restored"""
    import kyber_rs_amd as K
    n = 6000
    s = synth.scalars(n, 910); kk = synth.scalars(n, 911, b"k")
    pts = oracle.mul_base_ext_batch(synth.scalars(n, 912, b"p"))
    msg_list = synth.messages(n, 913)
    msgs = K.pack_messages(msg_list)
    assert len(msgs) == n and np.array_equal(engine.schnorr_sign(s[100:164], kk[100:164], msgs[100:164]), engine.schnorr_sign(s[100:164], kk[100:164], msg_list[100:164]))
    was = engine.get_option("host.zero_copy_kib")
    got = {}
    try:
        for kib in (0, 64, 4096, 65536):
            engine.set_option("host.zero_copy_kib", kib)
            pubs = engine.mul_base(s)
            sig = engine.schnorr_sign(s, kk, msgs)
            bad = sig.copy(); bad[::11, 40] ^= 2
            got[kib] = (pubs, engine.mul(kk, pts_ext=pts), engine.mul(kk, pts_enc=pubs), sig, engine.verify(pubs, msgs, bad, 1),
                        engine.pripoly_eval(s[:300], np.arange(n, dtype=np.uint32)), engine.lincomb(s[:5400].reshape(600, 9, 32), pts_ext=pts[:5400].reshape(600, 9, 40)))
    finally:
        engine.set_option("host.zero_copy_kib", was)
    for kib in (64, 4096, 65536):
        for a_, b_ in zip(got[0], got[kib]):
            assert np.array_equal(a_, b_), kib
    pubs, mul_ext, mul_enc, sig, st, shares, lc = got[4096]
    assert np.array_equal(pubs, oracle.mul_base_batch(s, nthreads=8)) and np.array_equal(mul_ext, oracle.mul_batch(kk, pts, nthreads=8))
    assert np.array_equal(sig, oracle.schnorr_sign_batch(s, kk, msg_list, nthreads=8))
    bad = sig.copy(); bad[::11, 40] ^= 2
    assert np.array_equal(st, oracle.verify_batch(1, pubs, msg_list, bad, nthreads=8))
    assert bytes(shares[n - 1]) == oracle.pripoly_eval(s[:300], n - 1) and bytes(lc[599]) == oracle.lincomb(s[5391:5400], pts[5391:5400])


def test_host_pointer_paths_agree(engine, oracle):
    """The chunked host-pointer pipeline gives the same bytes whichever way the batch travels: pageable
    caller memory (engine's bounce buffers + copy threads), page-locked caller memory (direct DMA), one
    copy thread, and the device-pointer API; ragged size so that the last chunk is short"""
    import torch
    n = (1 << 17) + 12345
    rng = np.random.default_rng(88)
    s = rng.integers(0, 256, (n, 32), dtype=np.uint8); s[:, 31] &= 0x0F
    enc_b, ext = engine.mul_base(s, want_ext=True)                       # pageable in/out, two outputs
    ref = torch.empty((n, 32), dtype=torch.uint8, device="cuda:0")
    engine.mul_base_dev(torch.from_numpy(s).to("cuda:0"), out_enc=ref)
    engine.sync()
    assert np.array_equal(enc_b, ref.cpu().numpy())
    t = rng.integers(0, 256, (n, 32), dtype=np.uint8); t[:, 31] &= 0x0F
    enc_m, ok = engine.mul(t, pts_enc=enc_b, want_ok=True)               # pageable, encoded points in, ok flags out
    assert ok.all()
    enc_m2 = engine.mul(t, pts_ext=ext)                                  # pageable, 160-byte points in
    assert np.array_equal(enc_m, enc_m2)
    ps = engine.pinned_array((n, 32), np.uint8); ps[:] = t
    pe = engine.pinned_array((n, 40), np.int32); pe[:] = ext
    po = engine.pinned_array((n, 32), np.uint8)
    engine.mul_into(ps, pe, po)                                          # page-locked: direct DMA path
    assert np.array_equal(po, enc_m)
    threads = engine.get_option("host.copy_threads")
    assert threads >= 1
    engine.set_option("host.copy_threads", 1)
    try:
        assert np.array_equal(engine.mul(t, pts_ext=ext), enc_m)
    finally:
        engine.set_option("host.copy_threads", 0)
    idx = np.concatenate([rng.choice(n, 256, replace=False), [0, n - 1, (1 << 17) - 1, 1 << 17]])
    assert np.array_equal(enc_m[idx], oracle.mul_batch(t[idx], ext[idx], nthreads=8))
    # other chunk plans of the pipeline (unit = 1/2 and 1/64 of the batch: two chunks / the 16-chunk cap), page-locked and pageable,
    # with both outputs travelling back
    unit = engine.get_option("host.pipe_chunks")
    try:
        for div in (2, 64):
            engine.set_option("host.pipe_chunks", div)
            po[:] = 0
            engine.mul_into(ps, pe, po)
            assert np.array_equal(po, enc_m), div
            e2, x2 = engine.mul(t, pts_ext=ext, want_ext=True)
            assert np.array_equal(e2, enc_m) and np.array_equal(engine.encode(x2[idx]), enc_m[idx]), div
            assert np.array_equal(engine.mul_base(s), enc_b), div
    finally:
        engine.set_option("host.pipe_chunks", unit)


def test_large_pageable_input_of_an_unchunked_call(engine, oracle):
    """kyb_sum_batch with 21.6 MB of pageable points: the input travels through the two page-locked bounce buffers in 8 MiB pieces
    (engine.hip h2d: three pieces, both buffers reused).  sum_j s_j B == (sum_j s_j mod L) B ties the result to the fixed-base kernel."""
    m, t = 3, 45000
    s = synth.scalars(m * t, 123)
    _, pts = engine.mul_base(s, want_ext=True)
    ints = [int.from_bytes(bytes(r), "little") for r in s]
    tot = np.frombuffer(b"".join((sum(ints[g * t:(g + 1) * t]) % synth.L).to_bytes(32, "little") for g in range(m)), dtype=np.uint8).reshape(m, 32)
    want = engine.mul_base(tot)
    assert want.tolist() == [list(oracle.mul_base_batch(tot[g:g + 1])[0]) for g in range(m)]
    for _ in range(2):                                                     # twice: the bounce buffers and their events are reused
        assert np.array_equal(engine.sum_points(pts.reshape(m, t, 40)), want)
    # and a pinned caller buffer takes the direct path
    pp = engine.pinned_array((m, t, 40), np.int32); pp[:] = pts.reshape(m, t, 40)
    assert np.array_equal(engine.sum_points(pp), want)


def test_bad_arguments_are_rejected(xengine, oracle):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    """error behaviour of the C ABI (INTEGRATION.md §3): a bad call returns a negative code with a message and
    leaves the engine usable; nothing is written on error"""
    import ctypes
    import torch
    import kyber_rs_amd
    lib = engine.lib
    s = synth.scalars(4, 5)
    out = np.full((4, 32), 0xAB, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.kyb_mul_base_batch(None, 4, p(out), None) == -2
    assert b"null" in lib.kyb_last_error()
    assert lib.kyb_mul_base_batch(p(s), 4, None, None) == -2                         # no output requested
    assert lib.kyb_mul_batch(p(s), None, None, 4, p(out), None, None) == -2          # neither encodings nor limbs
    ext = oracle.mul_base_ext_batch(s)
    enc = np.stack([np.frombuffer(oracle.encode(e), dtype=np.uint8) for e in ext])
    assert lib.kyb_mul_batch(p(s), p(enc), p(ext), 4, p(out), None, None) == -2      # both given
    assert (out == 0xAB).all()
    off_bad = np.array([0, 8, 4, 12, 16], dtype=np.uint32)
    msgs = np.zeros(17, dtype=np.uint8)
    sig = np.zeros((4, 64), dtype=np.uint8)
    assert lib.kyb_schnorr_sign_batch(p(s), p(s), p(msgs), p(off_bad), 4, p(sig)) == -2
    assert b"non-decreasing" in lib.kyb_last_error()
    assert lib.kyb_verify_batch(p(enc), p(msgs), p(off_bad), p(sig), 4, 7, p(out)) == -2   # unknown flavor / bad offsets
    assert lib.kyb_lincomb_batch(p(s), None, p(ext), 0, 2, 0, p(out), None, None) == -2    # t = 0
    assert lib.kyb_pubpoly_eval_batch(p(ext), 4, p(np.array([0xFFFFFFFF], dtype=np.uint32)), 1, p(out), None) == -2   # index + 1 overflows
    assert lib.kyb_set_option(b"no.such.option", 1) == -2
    assert lib.kyb_set_option(b"mul_base.radix", 48) == -2
    # this round's entry points: the same discipline
    bigidx = np.array([3, 0xFFFFFFFF], dtype=np.uint32)
    sh = np.zeros((2, 32), dtype=np.uint8)
    assert lib.kyb_pripoly_eval_batch(p(s), 1, 4, p(bigidx), 2, p(sh)) == -2 and b"index" in lib.kyb_last_error()
    assert lib.kyb_pripoly_eval_batch(p(s), 1, 0, p(bigidx), 1, p(sh)) == -2                                   # no coefficients
    assert lib.kyb_pripoly_eval_batch(None, 0, 4, None, 0, None) == 0 and lib.kyb_pripoly_eval_batch(p(s), 1, 4, None, 0, None) == 0
    lam = np.zeros((2, 32), dtype=np.uint8)
    assert lib.kyb_lagrange_coeffs_batch(p(bigidx), 1, 2, p(lam)) == -2 and lib.kyb_lagrange_coeffs_batch(None, 0, 2, None) == 0
    assert lib.kyb_dkg_verify_round_enc(p(enc), 0, 1, 0, p(out), None, None, None, None) == -2                 # t = 0
    assert lib.kyb_dkg_verify_round_enc(p(enc), 4, 1, 0xFFFFFFFF, p(out), None, None, None, None) == -2
    assert lib.kyb_dkg_verify_round_enc(p(enc), 4, 1, 0, None, None, None, None, None) == -2                   # nowhere to put the evaluations
    assert lib.kyb_lincomb_public_batch(p(s), None, p(ext), 0, 2, 0, p(out), None, None) == -2
    assert lib.kyb_lincomb_public_batch(p(s), p(enc), p(ext), 0, 2, 2, p(out), None, None) == -2               # both point forms
    assert lib.kyb_mul_public_batch(p(s), None, None, 4, p(out), None, None) == -2                             # no points
    assert lib.kyb_set_option(b"host.zero_copy_kib", -1) == -2 and lib.kyb_set_option(b"coop.share_by_load", 2) == -2
    assert lib.kyb_set_option(b"ladder.pair_max_items", -5) == -2
    # device-pointer API: misaligned buffers are refused before any launch
    d = torch.zeros(4 * 32 + 16, dtype=torch.uint8, device="cuda:0")
    o = torch.zeros((4, 32), dtype=torch.uint8, device="cuda:0")
    assert lib.kyb_mul_base_batch_dev(ctypes.c_void_p(d.data_ptr() + 4), 4, ctypes.c_void_p(o.data_ptr()), None, None) == -2
    assert b"aligned" in lib.kyb_last_error()
    # n = 0 is a no-op that succeeds even with null buffers
    assert lib.kyb_mul_base_batch(None, 0, None, None) == 0
    assert lib.kyb_verify_batch(None, None, None, None, 0, 0, None) == 0
    # and the engine still works
    assert np.array_equal(engine.mul_base(s), oracle.mul_base_batch(s))
    with pytest.raises(kyber_rs_amd.KyberHipError):
        engine.set_option("finish.min_items", 0)


def test_cfg5_whole_2_24_on_one_gpu(engine, oracle):
    """BASELINE config 5's whole batch (2^24 items) as ONE launch: buffers beyond 2^31 bytes (2^24 x 160 B of points,
    2^24 x 128 B of staging) exercise the 64-bit index arithmetic; variable base on P = B must equal fixed base for
    every item, and a sample must equal the oracle"""
    import torch
    n = 1 << 24
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev); g.manual_seed(24)
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device=dev, generator=g)
    s[:, 31] &= 0x1F                                        # up to 2^253: reduced and unreduced scalars
    enc_fixed = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    engine.mul_base_dev(s, out_enc=enc_fixed)
    bext = torch.from_numpy(oracle.base()).to(dev).repeat(n, 1)
    assert bext.numel() * 4 > 2**31
    enc_var = torch.empty_like(enc_fixed)
    engine.mul_dev(s, pts_ext=bext, out_enc=enc_var)          # (default stream: torch's current one — ordered behind the `.repeat` that writes bext)
    engine.sync()
    bad = (enc_fixed != enc_var).any(dim=1)
    if bool(bad.any()):                                       # say WHICH side is wrong before failing
        i = int(torch.nonzero(bad)[0].item())
        want = bytes(oracle.mul_base_batch(s[i:i + 1].cpu().numpy())[0])
        pytest.fail(f"{int(bad.sum())} of {n} items differ, first at {i} (scalar top byte {int(s[i, 31])}): fixed base "
                    f"{'==' if bytes(enc_fixed[i].cpu().numpy()) == want else '!='} oracle, variable base {'==' if bytes(enc_var[i].cpu().numpy()) == want else '!='} oracle")
    # 2^16 random sample compared element-wise, SHA-256 over all outputs stable across a repeat run
    idx = torch.cat([torch.randint(0, n, (1 << 16,), generator=torch.Generator().manual_seed(1)), torch.tensor([0, n - 1, (1 << 23) - 1, 1 << 23, (1 << 24) - 1025])])
    threads = min(16, len(os.sched_getaffinity(0)))
    assert np.array_equal(enc_fixed[idx.to(dev)].cpu().numpy(), oracle.mul_base_batch(s[idx.to(dev)].cpu().numpy(), nthreads=threads))
    digest = hashlib.sha256(enc_fixed.cpu().numpy().tobytes()).hexdigest()
    enc_fixed.zero_()
    engine.mul_base_dev(s, out_enc=enc_fixed)
    engine.sync()
    assert hashlib.sha256(enc_fixed.cpu().numpy().tobytes()).hexdigest() == digest
    del bext, enc_var, enc_fixed, s
    torch.cuda.empty_cache()


def test_dev_calls_are_ordered_with_torchs_current_stream(engine, oracle):
    """Round 5's red run: the binding's _dev methods used to launch on the engine's own NON-BLOCKING stream, which waits for nothing torch has
    queued — operands still being written by null-stream kernels were read early (tools/repro_stream_race.py, profiles/r06/stream_race.log).
    Now the default is torch's current stream.  Here the operands are written BEHIND a few milliseconds of unrelated work on that stream and the
    engine is called at once: on the null stream, and inside `with torch.cuda.stream(...)`; results are also consumed by torch with no
    synchronisation in between."""
    import torch
    import kyber_rs_amd
    dev = torch.device("cuda:0")
    n = 1 << 15
    s_np = synth.scalars(n, 661)
    p_np = np.tile(rand_points_ext(oracle, 64, 662), (n // 64, 1))
    want_var = oracle.mul_batch(s_np, p_np, nthreads=8)
    want_fix = oracle.mul_base_batch(s_np, nthreads=8)
    s_src, p_src = torch.from_numpy(s_np).to(dev), torch.from_numpy(p_np).to(dev)
    junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for stream in (None, side):
        with torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext():
            s = torch.zeros_like(s_src); p = torch.zeros_like(p_src)
            out_v = torch.zeros((n, 32), dtype=torch.uint8, device=dev); out_f = torch.zeros_like(out_v)
            for _ in range(8):
                junk.random_()                                # the current stream is busy for a while ...
            s.copy_(s_src); p.copy_(p_src)                    # ... the operands are written behind that ...
            engine.mul_dev(s, pts_ext=p, out_enc=out_v)       # ... and the engine is called without any wait
            engine.mul_base_dev(s, out_enc=out_f)
            got_v, got_f = out_v.clone(), out_f.clone()       # consumed by torch on the same stream, again without a wait
            out_v.zero_(); out_f.zero_()
        torch.cuda.synchronize()
        assert np.array_equal(got_v.cpu().numpy(), want_var), stream
        assert np.array_equal(got_f.cpu().numpy(), want_fix), stream
    # the explicit handles of the C ABI: KYB_STREAM_LEGACY is the null stream; 0 is the engine's own stream, which the CALLER orders
    s.copy_(s_src)
    engine.mul_base_dev(s, out_enc=out_f, stream=kyber_rs_amd.STREAM_LEGACY)
    assert np.array_equal(out_f.cpu().numpy(), want_fix)
    out_f.zero_()
    torch.cuda.synchronize()
    engine.mul_base_dev(s, out_enc=out_f, stream=kyber_rs_amd.STREAM_ENGINE)
    engine.sync(kyber_rs_amd.STREAM_ENGINE)
    assert np.array_equal(out_f.cpu().numpy(), want_fix)
    engine.stream_release(side.cuda_stream)


def test_structured_fuzz_against_oracle(engine, oracle):
    """2^16 structured (scalar, point) pairs, every output compared with the oracle: scalars with long runs of equal
    bits, single bits, digit patterns at the recoding extremes (radix 16 / 32 / 64 windows all 0x8.., 0x7.., 31, 32),
    values around multiples of L and around 2^252..2^256; points with a torsion component, small-order points, the
    neutral element; both multiplications and the fixed-base routine"""
    rng = np.random.default_rng(2024)
    L = synth.L
    n = 1 << 16
    ints = []
    for i in range(256):
        ints += [1 << i, (1 << i) - 1, (1 << 256) - (1 << i)]
    for k in range(1, 17):
        ints += [(k * L + d) % (1 << 256) for d in (-2, -1, 0, 1, 2)]
    for w, vals in ((4, (7, 8, 9, 15)), (5, (15, 16, 17, 31)), (6, (31, 32, 33, 63))):
        for v in vals:
            ints.append(sum(v << (w * i) for i in range(256 // w + 1)) % (1 << 256))
            ints.append(sum((v if i % 2 else 0) << (w * i) for i in range(256 // w + 1)) % (1 << 256))
    while len(ints) < n:
        kind = len(ints) % 4
        if kind == 0:      # runs of ones and zeros
            v, pos = 0, 0
            while pos < 256:
                run = int(rng.integers(1, 40))
                if rng.integers(0, 2):
                    v |= ((1 << run) - 1) << pos
                pos += run
            ints.append(v % (1 << 256))
        elif kind == 1:    # sparse
            v = 0
            for _ in range(int(rng.integers(1, 6))):
                v |= 1 << int(rng.integers(0, 256))
            ints.append(v)
        elif kind == 2:    # reduced random
            ints.append(int.from_bytes(rng.bytes(32), "little") % L)
        else:              # raw 256-bit
            ints.append(int.from_bytes(rng.bytes(32), "little"))
    ints = ints[:n]
    s = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in ints), dtype=np.uint8).reshape(n, 32)
    assert np.array_equal(engine.mul_base(s), oracle.mul_base_batch(s, nthreads=8))
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    base_pts = oracle.mul_base_ext_batch(synth.scalars(64, 2024))
    pool = [p for p in base_pts] + [oracle.add(base_pts[i], weak[2 + i % 3]) for i in range(16)] + weak + [oracle.null()]
    pts = np.stack([pool[int(j)] for j in rng.integers(0, len(pool), n)])
    got = engine.mul(s, pts_ext=pts)
    assert np.array_equal(got, oracle.mul_batch(s, pts, nthreads=8))


def test_cfg4_full_2_18_signatures(engine, oracle):
    """BASELINE config 4 at its full size: 2^18 (x, k, 32-byte message) triples, EVERY signature compared with the
    oracle; the keyed variant (signer holds its public key) gives the same bytes"""
    n = 1 << 18
    rng = np.random.default_rng(418)
    x = rng.integers(0, 256, (n, 32), dtype=np.uint8); x[:, 31] &= 0x0F
    k = rng.integers(0, 256, (n, 32), dtype=np.uint8); k[:, 31] &= 0x0F
    blob = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    msgs = [bytes(r) for r in blob]
    sig = engine.schnorr_sign(x, k, msgs)
    threads = min(16, len(os.sched_getaffinity(0)))
    assert np.array_equal(sig, oracle.schnorr_sign_batch(x, k, msgs, nthreads=threads))
    assert np.array_equal(engine.schnorr_sign(x, k, msgs, pubs=engine.mul_base(x)), sig)


def test_soak_three_threads_mixed_operations(engine, oracle):
    """three host threads hammer the engine for a few seconds with mixed operations and sizes (host-pointer API:
    staging, bounce buffers, copy threads, the verification side stream; device-pointer API on a stream of their own):
    every result must equal what the same call gives alone"""
    import threading
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(99)
    N = 70000
    s = synth.scalars(4096, 500)
    s = np.tile(s, (N // 4096 + 1, 1))[:N].copy()
    k = np.roll(s, 1, axis=0).copy()
    pts = rand_points_ext(oracle, 256, 500)
    pts = np.tile(pts, (N // 256 + 1, 1))[:N].copy()
    msgs = synth.messages(2048, 500)
    # reference results, computed alone
    ref_mul = engine.mul(s, pts_ext=pts)
    ref_base = engine.mul_base(s)
    ref_sig = engine.schnorr_sign(s[:2048], k[:2048], msgs)
    ref_pub = ref_base[:2048]
    bad_sig = ref_sig.copy(); bad_sig[::3, 40] ^= 1
    ref_st = engine.verify(ref_pub, msgs, bad_sig, 1)
    assert (ref_st[1::3] == 0).all() and (ref_st[::3] != 0).all()
    errors = []
    stop = time.time() + float(os.environ.get("KYB_SOAK_SECONDS", "4"))

    def worker(tid):
        r = np.random.default_rng(1000 + tid)
        stream = torch.cuda.Stream(device=dev)
        S = torch.from_numpy(s).to(dev); P = torch.from_numpy(pts).to(dev)
        O = torch.empty((N, 32), dtype=torch.uint8, device=dev)
        it = 0
        try:
            while time.time() < stop:
                it += 1
                n = int(r.choice([1, 7, 64, 300, 2048, 5000, 66000]))
                op = it % 5
                if op == 0:
                    assert np.array_equal(engine.mul(s[:n], pts_ext=pts[:n]), ref_mul[:n]), ("mul", n)
                elif op == 1:
                    assert np.array_equal(engine.mul_base(s[:n]), ref_base[:n]), ("mul_base", n)
                elif op == 2:
                    m = min(n, 2048)
                    assert np.array_equal(engine.schnorr_sign(s[:m], k[:m], msgs[:m]), ref_sig[:m]), ("sign", m)
                elif op == 3:
                    m = min(n, 2048)
                    assert np.array_equal(engine.verify(ref_pub[:m], msgs[:m], bad_sig[:m], 1), ref_st[:m]), ("verify", m)
                else:
                    engine.mul_dev(S[:n], pts_ext=P[:n], out_enc=O[:n], stream=stream.cuda_stream)
                    stream.synchronize()
                    assert np.array_equal(O[:n].cpu().numpy(), ref_mul[:n]), ("mul_dev", n)
        except Exception as e:  # noqa: BLE001
            errors.append((tid, it, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors


def test_verify_equation_on_raw_kernels_all_golden_lines(engine):
    """eddsa_sig.rs:194-211 on the reference's own 1024 golden signatures, built from the RAW batch entry points —
    kyb_decode_batch (A, R), kyb_mul_batch (h*A: the ladder), kyb_add_batch (R + h*A), kyb_encode_batch, kyb_mul_base_batch
    (s*B) — with h computed by hashlib and Python integers: ties the variable-base kernel to reference-held data
    without the oracle and without kyb_verify_batch's own plumbing."""
    L = 2**252 + 27742317777372353535851937790883648493
    pubs, rs, ss, hs = [], [], [], []
    for ln in gzip.open(os.path.join(HERE, "golden", "sign.input.gz"), "rt").read().split("\n"):
        if not ln:
            continue
        p = ln.split(":")
        pub, msg, sig = bytes.fromhex(p[1]), bytes.fromhex(p[2]), bytes.fromhex(p[3])[:64]
        pubs.append(pub); rs.append(sig[:32]); ss.append(sig[32:])
        hs.append((int.from_bytes(hashlib.sha512(sig[:32] + pub + msg).digest(), "little") % L).to_bytes(32, "little"))
    assert len(pubs) == 1024
    as_u8 = lambda lst: np.frombuffer(b"".join(lst), dtype=np.uint8).reshape(-1, 32)
    a_ext, ok_a = engine.decode(as_u8(pubs))
    r_ext, ok_r = engine.decode(as_u8(rs))
    assert ok_a.all() and ok_r.all()
    _, ha_ext = engine.mul(as_u8(hs), pts_ext=a_ext, want_ext=True)
    lhs = engine.encode(engine.add(r_ext, ha_ext))
    rhs = engine.mul_base(as_u8(ss))
    assert np.array_equal(lhs, rhs)
    # and from the wire encodings of A (the decode runs inside kyb_mul_batch)
    _, ha2 = engine.mul(as_u8(hs), pts_enc=as_u8(pubs), want_ext=True)
    assert np.array_equal(engine.encode(engine.add(r_ext, ha2)), rhs)


def test_ladder_skips_the_leading_zeros_only_when_every_scalar_is_canonical(engine, oracle):
    """The batch ladder starts four bits lower when no scalar of the launch reaches 2^252 (k_mont_prep ORs their top bits on the way, two
    alternating words per stream): one unreduced scalar anywhere in the batch must switch the whole launch back to 256 steps, and the
    launches before and after it must not see its flag."""
    n = 8192
    s = synth.scalars(n, 61)
    pts = oracle.mul_base_ext_batch(synth.scalars(n, 62, b"p"))
    want = oracle.mul_batch(s, pts, nthreads=8)
    assert np.array_equal(engine.mul(s, pts_ext=pts), want)      # (two lanes per item at this size: no canonical test, always 256 steps)
    pair_max = engine.get_option("ladder.pair_max_items")
    engine.set_option("ladder.pair_max_items", 0)                # the one-lane kernel behind k_mont_prep, as for batches beyond 32,768 items
    try:
        _skip_canonical_cases(engine, oracle, n, s, pts, want)
    finally:
        engine.set_option("ladder.pair_max_items", pair_max)


def _skip_canonical_cases(engine, oracle, n, s, pts, want):
    assert np.array_equal(engine.mul(s, pts_ext=pts), want)
    for pos, top in ((n - 3, 0x10), (17, 0x20), (5, 0x80), (4097, 0xff)):
        t = s.copy()
        t[pos, 31] |= top                                   # bit 252, bit 253, bit 255, everything
        w = want.copy()
        w[pos] = np.frombuffer(oracle.mul(bytes(t[pos]), pts[pos]), dtype=np.uint8)
        assert np.array_equal(engine.mul(t, pts_ext=pts), w), (pos, top)
        assert np.array_equal(engine.mul(s, pts_ext=pts), want)      # the next launch is canonical again
    engine.set_option("ladder.skip_canonical", 0)
    try:
        assert np.array_equal(engine.mul(s, pts_ext=pts), want)
    finally:
        engine.set_option("ladder.skip_canonical", 1)
    # linear combinations with private points go through the same launch sequence
    lam = synth.scalars(40 * 200, 63).reshape(200, 40, 32)
    pp = pts[: 40 * 200].reshape(200, 40, 40)
    got = engine.lincomb(lam, pts_ext=pp)
    for g in (0, 99, 199):
        assert bytes(got[g]) == oracle.lincomb(lam[g], pp[g])


@pytest.mark.parametrize("block64", [512, 768, 1024])
def test_fixed_base_workgroup_sizes_and_wave_chunks(xengine, oracle, block64):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    """k_mul_base64 deals its items out per wavefront (chunks of 64, low wave numbers take the leftovers): every workgroup size gives the
    oracle's bytes on batches that end inside a chunk, inside a workgroup round and on a round boundary, and when signing puts two scalar
    arrays into one launch (the chunk that straddles them)."""
    cus = engine.device_info()["compute_units"]
    old = (engine.get_option("mul_base.block64"), engine.get_option("mul_base.small_chunks"), engine.get_option("coop.base_max_items"))
    engine.set_option("mul_base.block64", block64)
    engine.set_option("mul_base.small_chunks", 0)        # always the full-size workgroups, however small the batch
    engine.set_option("coop.base_max_items", 0)
    try:
        for n in (1, 63, 64, 65, block64 - 1, block64 + 1, 2 * block64 + 77, cus * block64 + 64 * 5 + 3, 300001):
            s = np.concatenate([synth.scalars(n - n // 3, 900 + n), synth.raw256(n // 3, 900 + n)])
            assert np.array_equal(engine.mul_base(s), oracle.mul_base_batch(s, nthreads=16)), (block64, n)
        for n in (1, 31, 97, 1000, 5003):                  # signing: k*B and x*B in one launch of 2n items
            x, k = synth.scalars(n, 910 + n, b"x"), synth.scalars(n, 911 + n, b"k")
            msgs = synth.messages(n, 912, length=9)
            assert np.array_equal(engine.schnorr_sign(x, k, msgs), oracle.schnorr_sign_batch(x, k, msgs, nthreads=16)), (block64, n)
    finally:
        engine.set_option("mul_base.block64", old[0])
        engine.set_option("mul_base.small_chunks", old[1])
        engine.set_option("coop.base_max_items", old[2])


@pytest.mark.parametrize("m,t", [(16, 512), (64, 128), (300, 40), (683, 683)])
def test_lincomb_public_point_tables(engine, oracle, m, t):
    """kyb_lincomb_public_batch over shared points (window tables of the points, kernels_msm.hip) == kyb_lincomb_batch (constant-time ladders)
    == the oracle's mul + add, with the scalars the multiplication routines treat specially (0, 1, L - 1, L, 8L, 2^255 - 1, >= 2^255:
    top-digit quirk, all ones) and points that are neutral, of small order, of mixed order, repeated and negated; wire encodings in, one of
    them not a point"""
    L = synth.L
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    sc = synth.scalars(m * t, 5000 + t).reshape(m, t, 32).copy()
    pts = oracle.mul_base_ext_batch(synth.scalars(t, 5100 + t, b"point")).copy()
    special = [0, 1, 2, 31, 32, 33, 63, 64, L - 1, L, L + 1, 8 * L, (1 << 255) - 1, 1 << 255, (1 << 255) + (1 << 252) * 5 + 12345, (1 << 256) - 1, 1 << 252, (1 << 252) - 1]
    for k, v in enumerate(special):
        sc[k % m, (3 * k + 1) % t] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
    sc[m - 1] = synth.raw256(t, 5200 + t)                                    # a row of unreduced 256-bit scalars
    pts[0] = oracle.null()
    pts[1] = weak[2]
    pts[2] = oracle.add(pts[3], weak[3])
    pts[4] = pts[5]
    pts[6] = oracle.neg(pts[5])
    got, ext = engine.lincomb(sc, pts_ext=pts, want_ext=True, public=True)
    ref = engine.lincomb(sc, pts_ext=pts)
    assert np.array_equal(got, ref)
    for g in sorted({0, 1, m // 2, m - 1}):
        assert bytes(got[g]) == oracle.lincomb(sc[g], pts), (m, t, g)
        assert oracle.encode(ext[g]) == bytes(got[g])
    pe = oracle.encode_batch(pts)
    bad = next(bytes([v]) + bytes(31) for v in range(2, 50) if not oracle.decode(bytes([v]) + bytes(31))[1])
    pe[7] = np.frombuffer(bad, dtype=np.uint8)
    pts2 = pts.copy(); pts2[7] = oracle.null()
    got2, ok = engine.lincomb(sc, pts_enc=pe, want_ok=True, public=True)
    assert ok.sum() == t - 1 and ok[7] == 0
    assert np.array_equal(got2, engine.lincomb(sc, pts_ext=pts2))


def test_lincomb_public_small_shapes_take_the_ladder(engine, oracle):
    """shapes below the table threshold (and points that are not shared) go through kyb_lincomb_batch's own path: same bytes"""
    for m, t, shared in ((3, 5, True), (15, 700, True), (40, 9, False)):
        sc = synth.scalars(m * t, 5300 + t).reshape(m, t, 32)
        pts = oracle.mul_base_ext_batch(synth.scalars(t if shared else m * t, 5400 + t, b"point"))
        pts = pts if shared else pts.reshape(m, t, 40)
        assert np.array_equal(engine.lincomb(sc, pts_ext=pts, public=True), engine.lincomb(sc, pts_ext=pts))
