"""The wire-format entry points of SURVEY.md §8f N1: a Deal carries its dealer's commitments as 32-byte encodings
(vss/pedersen/vss.rs:113-124), which the reference unmarshals one at a time (point.rs:43-51) before PubPoly::eval
(poly.rs:457-469) or PubPoly::add (poly.rs:486-507) use them.  kyb_pubpoly_eval_multi_enc_batch / kyb_sum_enc_batch take the
encodings as they arrive.  Checked against the oracle's own decode + eval / add on every path (one group or evaluation per
wavefront, the batch kernels), with encodings the reference accepts in unusual forms and encodings it rejects."""
import json
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
P = (1 << 255) - 19
KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.json")))


def _points(oracle, n, seed):
    ext = oracle.mul_base_ext_batch(synth.scalars(n, seed, b"wire"))
    enc = np.stack([np.frombuffer(oracle.encode(p), dtype=np.uint8) for p in ext])
    return ext, enc


def _spoil(oracle, enc, ext):
    """Every 37th encoding does not decode, every 41st is a non-canonical form of the neutral element's y (y = p + 1), every 43rd a
    point of small order, every 47th has bit 255 of a y with x = 0 set (accepted by the reference, ge.rs:124-179)."""
    enc, ext = enc.copy(), ext.copy()
    bad = next(bytes([v]) + bytes(31) for v in range(2, 60) if not oracle.decode(bytes([v]) + bytes(31))[1])
    weak = [bytes.fromhex(h) for h in KATS["weak_keys"]]
    want_ok = np.ones(len(enc), dtype=np.uint8)
    for i in range(len(enc)):
        raw = None
        if i % 37 == 5:
            raw = bad
        elif i % 41 == 7:
            raw = (P + 1).to_bytes(32, "little")
        elif i % 43 == 11:
            raw = weak[(i // 43) % len(weak)]
        elif i % 47 == 13:
            raw = (1 | (1 << 255)).to_bytes(32, "little")
        if raw is None:
            continue
        enc[i] = np.frombuffer(raw, dtype=np.uint8)
        p, ok = oracle.decode(raw)
        want_ok[i] = 1 if ok else 0
        ext[i] = p if ok else oracle.null()
    return enc, ext, want_ok


@pytest.mark.parametrize("m,t,k", [(1, 1, 1), (5, 7, 3), (24, 17, 1), (40, 30, 2), (300, 40, 30), (64, 150, 1)])
def test_eval_from_wire_encodings(engine, oracle, m, t, k):
    ext, enc = _points(oracle, m * t, 700 + m)
    enc, ext, want_ok = _spoil(oracle, enc, ext)
    rng = np.random.default_rng(m * 1000 + t)
    idx = rng.integers(0, 2000, (m, k), dtype=np.uint64).astype(np.uint32)
    got, ok = engine.pubpoly_eval_multi_enc(enc.reshape(m, t, 32), idx)
    assert np.array_equal(ok.reshape(-1), want_ok)
    assert np.array_equal(got, engine.pubpoly_eval_multi(ext.reshape(m, t, 40), idx))      # same kernels behind a host-side decode
    for g in sorted({0, m // 2, m - 1}):
        for j in sorted({0, k - 1}):
            assert bytes(got[g, j]) == oracle.pubpoly_eval(ext.reshape(m, t, 40)[g], int(idx[g, j]))
    got2, ext2, _ = engine.pubpoly_eval_multi_enc(enc.reshape(m, t, 32), idx, want_ext=True)
    assert np.array_equal(got2, got)
    assert np.array_equal(engine.encode(ext2.reshape(-1, 40)), got.reshape(-1, 32))


@pytest.mark.parametrize("m,t", [(1, 1), (1, 2), (17, 24), (3, 32), (4, 33), (40, 50), (200, 16), (683, 64), (5000, 3)])
def test_sums_from_wire_encodings(engine, oracle, m, t):
    ext, enc = _points(oracle, m * t, 800 + t)
    enc, ext, want_ok = _spoil(oracle, enc, ext)
    want = engine.sum_points(ext.reshape(m, t, 40))
    for g in sorted({0, m // 3, m - 1}):
        acc = ext.reshape(m, t, 40)[g, 0]
        for j in range(1, t):
            acc = oracle.add(acc, ext.reshape(m, t, 40)[g, j])
        assert bytes(want[g]) == oracle.encode(acc)
    got, ok = engine.sum_points_enc(enc.reshape(m, t, 32))
    assert np.array_equal(ok.reshape(-1), want_ok)
    assert np.array_equal(got, want)
    # as received: dealer-major (t dealers x m coefficients), summed over the dealers without a transposition on the host
    as_received = np.ascontiguousarray(enc.reshape(m, t, 32).transpose(1, 0, 2))
    got_t, ext_t, ok_t = engine.sum_points_enc(as_received, item_major=True, want_ext=True)
    assert np.array_equal(got_t, want)
    assert np.array_equal(ok_t, want_ok.reshape(m, t).T)
    assert np.array_equal(engine.encode(ext_t), want)


def test_wire_format_bad_arguments(engine):
    lib = engine.lib
    z = np.zeros((4, 32), dtype=np.uint8)
    out = np.zeros((4, 32), dtype=np.uint8)
    idx = np.zeros(4, dtype=np.uint32)
    p = lambda a: a.ctypes.data
    assert lib.kyb_sum_enc_batch(None, 2, 2, 0, p(out), None, None) != 0
    assert lib.kyb_sum_enc_batch(p(z), 2, 0, 0, p(out), None, None) != 0
    assert lib.kyb_sum_enc_batch(p(z), 2, 2, 0, None, None, None) != 0
    assert lib.kyb_sum_enc_batch(p(z), 0, 2, 0, p(out), None, None) == 0                     # nothing to do
    assert lib.kyb_pubpoly_eval_multi_enc_batch(None, 2, 2, p(idx), 1, p(out), None, None) != 0
    assert lib.kyb_pubpoly_eval_multi_enc_batch(p(z), 0, 2, p(idx), 1, p(out), None, None) != 0
    assert lib.kyb_pubpoly_eval_multi_enc_batch(p(z), 2, 2, p(idx), 1, None, None, None) != 0
    idx[0] = 0xffffffff
    assert lib.kyb_pubpoly_eval_multi_enc_batch(p(z), 2, 2, p(idx), 1, p(out), None, None) != 0   # index + 1 must fit 32 bits


@pytest.mark.parametrize("m,t", [(1, 1), (5, 3), (16, 11), (64, 43), (300, 201), (40, 700)])
def test_dkg_round_in_one_call(engine, oracle, m, t):
    """kyb_dkg_verify_round_enc: the m x t commitments decoded once, every dealer's polynomial evaluated at the node's index AND the columns
    summed == the two separate wire-format calls == the oracle (decode, then PubPoly::eval / add), with encodings the reference rejects,
    non-canonical forms it accepts and small-order points in the mix"""
    ext, enc = _points(oracle, m * t, 900 + m)
    enc, ext, want_ok = _spoil(oracle, enc, ext)
    index = (m * 7 + t) % 1500
    ev, sums, ok = engine.dkg_verify_round_enc(enc.reshape(m, t, 32), index)
    assert np.array_equal(ok.reshape(-1), want_ok)
    ev2, ok2 = engine.pubpoly_eval_multi_enc(enc.reshape(m, t, 32), np.full((m, 1), index, dtype=np.uint32))
    assert np.array_equal(ev, ev2[:, 0]) and np.array_equal(ok2, ok)
    sums2, _ = engine.sum_points_enc(enc.reshape(m, t, 32), item_major=True)
    assert np.array_equal(sums, sums2)
    E = ext.reshape(m, t, 40)
    for g in sorted({0, m // 2, m - 1}):
        assert bytes(ev[g]) == oracle.pubpoly_eval(E[g], index), (g, "eval")
    for j in sorted({0, t // 2, t - 1}):
        acc = E[0, j]
        for g in range(1, m):
            acc = oracle.add(acc, E[g, j])
        assert bytes(sums[j]) == oracle.encode(acc), (j, "sum")
    # evaluations only
    ev3, none, _ = engine.dkg_verify_round_enc(enc.reshape(m, t, 32), index, want_sums=False)
    assert none is None and np.array_equal(ev3, ev)
