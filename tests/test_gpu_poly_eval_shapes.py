"""PubPoly::eval (share/poly.rs:457-469) through its three launch shapes — one evaluation per wavefront, per lane, and one SEGMENT of
the Horner chain per lane recombined by the variable-base ladder (k_poly_eval_part) — must give the reference's point whatever
the shape: forced segment counts against the unsegmented kernels and the oracle, with commitments that carry small-order
components (the multiplier x^(s len) is only exact mod 8L, not mod L), neutral commitments and indices of every bit length."""
import json
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.json")))


def _commits(oracle, m, t, seed):
    ext = oracle.mul_base_ext_batch(synth.scalars(m * t, seed, b"shape")).reshape(m, t, 40)
    weak = [oracle.decode(bytes.fromhex(h))[0] for h in KATS["weak_keys"]]
    for g in range(m):
        for j in range(t):
            if (g * t + j) % 5 == 2:
                ext[g, j] = oracle.add(ext[g, j], weak[(g + j) % len(weak)])       # mixed order
            elif (g * t + j) % 17 == 3:
                ext[g, j] = weak[(g + j) % len(weak)]                               # small order
            elif (g * t + j) % 19 == 4:
                ext[g, j] = oracle.null()
    return ext


@pytest.fixture
def eng(xengine):
    """the shapes are forced through poly.segments / poly.batch_segments: selectors of the cross-check build (the product chooses by its cost model)"""
    yield xengine
    xengine.set_option("poly.batch_segments", 0)
    xengine.set_option("poly.segments", 0)


@pytest.mark.parametrize("m,t,k", [(3, 29, 2), (40, 50, 1), (1, 97, 33), (260, 12, 1)])
def test_segment_per_lane_matches_the_other_shapes(eng, oracle, m, t, k):
    commits = _commits(oracle, m, t, 900 + t)
    rng = np.random.default_rng(t)
    idx = rng.integers(0, 1 << 11, (m, k), dtype=np.uint64).astype(np.uint32)
    idx[0, 0] = 0                                                                   # x = 1
    eng.set_option("poly.batch_segments", 1)
    want = eng.pubpoly_eval_multi(commits, idx)
    for g in sorted({0, m - 1}):
        for j in sorted({0, k - 1}):
            assert bytes(want[g, j]) == oracle.pubpoly_eval(commits[g], int(idx[g, j]))
    for segs in (2, 3, 5, 8, 64, 256):
        if segs > t:
            continue
        eng.set_option("poly.batch_segments", segs)
        got, ext = eng.pubpoly_eval_multi(commits, idx, want_ext=True)
        assert np.array_equal(got, want), segs
        assert np.array_equal(eng.encode(ext.reshape(-1, 40)), want.reshape(-1, 32)), segs


def test_segment_per_lane_wide_indices_and_single_polynomial(eng, oracle):
    t = 41
    commits = _commits(oracle, 1, t, 77)[0]
    idx = np.array([0, 1, 2, 255, 256, 65535, 65536, (1 << 31) - 1, 1 << 31, 0xfffffffe], dtype=np.uint32)
    eng.set_option("poly.batch_segments", 1)
    want = eng.pubpoly_eval(commits, idx)
    for i in (0, 3, 7, 9):
        assert bytes(want[i]) == oracle.pubpoly_eval(commits, int(idx[i]))
    for segs in (2, 6, 10, 41):
        eng.set_option("poly.batch_segments", segs)
        assert np.array_equal(eng.pubpoly_eval(commits, idx), want), segs


def test_automatic_choice_takes_the_segmented_shape_for_long_polynomials(eng, oracle):
    """1,200 dealers' polynomials of 300 coefficients at one index each: the cost model picks the segmented batch shape (the profile
    shows the ladder); results equal the one-evaluation-per-wavefront kernels'."""
    m, t = 1200, 300
    base = _commits(oracle, 1, t, 5)[0]
    commits = np.tile(base[None, :, :], (m, 1, 1))
    commits[7, 3] = oracle.null()
    idx = np.full((m, 1), 733, dtype=np.uint32)
    idx[5, 0] = 12
    eng.set_option("poly.batch_segments", 1)
    want = eng.pubpoly_eval_multi(commits, idx)
    assert bytes(want[0, 0]) == oracle.pubpoly_eval(commits[0], 733) and bytes(want[7, 0]) == oracle.pubpoly_eval(commits[7], 733)
    assert bytes(want[5, 0]) == oracle.pubpoly_eval(commits[5], 12)
    eng.set_option("poly.batch_segments", 0)
    eng.profile_begin(16)
    got = eng.pubpoly_eval_multi(commits, idx)
    names = [n for n, _ in eng.profile_read(16)]
    assert np.array_equal(got, want)
    assert any(n.startswith("k_mul_ladder") for n in names), names
