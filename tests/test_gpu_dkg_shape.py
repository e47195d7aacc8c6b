"""The batch entry points at the shape a Pedersen DKG round uses them:
n dealers each commit to a degree-(t-1) polynomial (PriPoly::commit, poly.rs:195-206 -> one fixed-base
batch), every verifier checks every dealer's share against the public polynomial
(vss.rs:904-909: fig = mul(fi, None); PubPoly::eval(i) == fig), and the distributed public key is
the sum of the constant-term commitments (dkg.rs:905-953).  Everything runs on the GPU through the C
ABI; a sample is re-checked against the oracle's restatement of the reference's own routines."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
L = synth.L


def test_pedersen_round_shape(engine, oracle):
    n, t = 24, 17
    coeffs = synth.scalars(n * t, 90).reshape(n, t, 32)                       # dealer d, coefficient j
    # commitments of all dealers: n*t fixed-base mults in one call
    enc, ext = engine.mul_base(coeffs.reshape(-1, 32), want_ext=True)
    commits = ext.reshape(n, t, 40)
    assert np.array_equal(enc, oracle.mul_base_batch(coeffs.reshape(-1, 32), nthreads=8))
    # private shares f_d(i) = sum_j c_dj (i+1)^j mod L  (PriPoly::eval, poly.rs:133-141; host-side scalar arithmetic)
    ci = [[int.from_bytes(bytes(coeffs[d, j]), "little") for j in range(t)] for d in range(n)]
    shares = np.zeros((n, n, 32), dtype=np.uint8)
    for d in range(n):
        for i in range(n):
            v = sum(c * pow(i + 1, j, L) for j, c in enumerate(ci[d])) % L
            shares[d, i] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
    # every verifier's check of every dealer: PubPoly::eval(i) on the GPU vs mul(share, None) on the GPU
    idx = np.arange(n, dtype=np.uint32)
    fig = engine.mul_base(shares.reshape(-1, 32)).reshape(n, n, 32)
    for d in range(n):
        ev = engine.pubpoly_eval(commits[d], idx)
        assert np.array_equal(ev, fig[d])                                       # all n checks of dealer d pass
        if d % 8 == 0:
            for i in (0, n - 1):
                assert bytes(ev[i]) == oracle.pubpoly_eval(commits[d], i)       # the reference's own eval
    # the verifier's side in one launch: the deals of all n dealers checked at verifier v's own index (vss.rs:904-909),
    # and the whole n x n matrix of checks as n polynomials x n indices
    for v in (0, 7, n - 1):
        col = engine.pubpoly_eval_multi(commits, np.full((n, 1), v, dtype=np.uint32))
        assert np.array_equal(col[:, 0], fig[:, v])
    allchk = engine.pubpoly_eval_multi(commits, np.tile(idx, (n, 1)))
    assert np.array_equal(allchk, fig)
    # a corrupted share is caught by exactly its check
    bad = shares[3, 5].copy(); bad[0] ^= 1
    assert bytes(engine.mul_base(bad)[0]) != bytes(engine.pubpoly_eval(commits[3], idx[5:6])[0])
    # distributed public key: sum of the constant-term commitments == (sum of the constant terms) * B
    acc = commits[0, 0].copy()
    for d in range(1, n):
        acc = engine.add(acc[None, :], commits[d, 0][None, :])[0]
    secret = sum(ci[d][0] for d in range(n)) % L
    assert bytes(engine.encode(acc[None, :])[0]) == oracle.mul_base(secret.to_bytes(32, "little"))
    assert engine.equal(acc[None, :], oracle.mul_base_ext(secret.to_bytes(32, "little"))[None, :])[0] == 1
    # the whole distributed public polynomial: coefficient-wise sum over the dealers in one launch (t groups of n points)
    dist_poly, dist_ext = engine.sum_points(np.ascontiguousarray(commits.transpose(1, 0, 2)), want_ext=True)
    for j in (0, 1, t - 1):
        sj = sum(ci[d][j] for d in range(n)) % L
        assert bytes(dist_poly[j]) == oracle.mul_base(sj.to_bytes(32, "little"))
    assert engine.equal(dist_ext[0][None, :], acc[None, :])[0] == 1
    # sums with awkward operands: P + P, P + (-P), neutral elements, a single-element group
    odd = np.stack([commits[0, 0], commits[0, 0], oracle.neg(commits[0, 1]), commits[0, 1], oracle.null(), commits[0, 2], oracle.null()])[None, :, :]
    want = oracle.add(oracle.add(commits[0, 0], commits[0, 0]), commits[0, 2])
    assert bytes(engine.sum_points(odd)[0]) == oracle.encode(want)
    assert bytes(engine.sum_points(commits[3, 4][None, None, :])[0]) == bytes(enc.reshape(n, t, 32)[3, 4])
    # Diffie-Hellman of every dealer with every verifier (vss.rs:371-375, dh_impl.rs:74-80): n*n variable-base
    # mults straight from the wire encodings; both directions agree and match the reference's mul
    longterm = synth.scalars(n, 91)
    pubs = engine.mul_base(longterm)
    shared = engine.mul(np.repeat(longterm, n, axis=0), pts_enc=np.tile(pubs, (n, 1))).reshape(n, n, 32)
    assert np.array_equal(shared, shared.transpose(1, 0, 2))
    assert bytes(shared[2, 5]) == oracle.mul(bytes(longterm[2]), oracle.decode(bytes(pubs[5]))[0])
    # recover_commit (poly.rs:566-603): t public shares of every dealer give back its constant-term commitment,
    # all n dealers in one linear-combination launch
    pick = sorted(np.random.default_rng(5).choice(n, size=t, replace=False).tolist())
    xs = [i + 1 for i in pick]
    lam = []
    for xi in xs:
        num = den = 1
        for xj in xs:
            if xj != xi:
                num = num * xj % L
                den = den * (xj - xi) % L
        lam.append(num * pow(den, L - 2, L) % L)
    lam_b = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in lam), dtype=np.uint8).reshape(t, 32)
    pub_shares = np.stack([engine.pubpoly_eval(commits[d], np.array(pick, dtype=np.uint32), want_ext=True)[1] for d in range(n)])
    rec = engine.lincomb(np.broadcast_to(lam_b, (n, t, 32)), pts_ext=pub_shares)
    assert np.array_equal(rec, enc.reshape(n, t, 32)[:, 0])
    assert bytes(rec[7]) == oracle.lincomb(lam_b, pub_shares[7])


def test_lagrange_coefficients_and_recover_commit_on_the_gpu(engine, oracle):
    """kyb_lagrange_coeffs_batch == Python integers (prod x_j / (x_j - x_i) mod L, x = index + 1) for ragged index sets, and the whole
    recover_commit (poly.rs:566-603) — coefficients on the GPU, then kyb_lincomb_public_batch over the share points — gives back the secret
    commitment a_0 * B of the polynomial the shares were made from"""
    import time
    L = synth.L
    rng = np.random.default_rng(321)
    for m, t in ((1, 1), (1, 2), (3, 7), (5, 64), (2, 683)):
        idx = np.stack([np.sort(rng.choice(3 * t + 5, t, replace=False)) for _ in range(m)]).astype(np.uint32)
        if t > 2:
            idx[0, -1] = 0xfffffffe                                       # x = 2^32 - 1: the largest index the ABI takes
        lam = engine.lagrange_coeffs(idx)
        for g in range(m):
            xs = [int(v) + 1 for v in idx[g]]
            for i in sorted({0, t // 2, t - 1}):
                num = den = 1
                for j in range(t):
                    if j != i:
                        num = num * xs[j] % L
                        den = den * (xs[j] - xs[i]) % L
                want = num * pow(den, L - 2, L) % L
                assert int.from_bytes(bytes(lam[g, i]), "little") == want, (m, t, g, i)
    # a repeated index: the reference cannot produce one (shares are keyed by index); the coefficients it touches are 0 (inverse of 0 is 0)
    lam = engine.lagrange_coeffs(np.array([[4, 9, 4, 1]], dtype=np.uint32))
    assert not lam[0, 0].any() and not lam[0, 2].any() and lam[0, 1].any()
    # recover_commit of 40 share sets, threshold 21: shares f(x_i) * B of one secret polynomial per set
    m, t = 40, 21
    coeffs = [[int.from_bytes(bytes(c), "little") for c in synth.scalars(t, 700 + g)] for g in range(m)]
    idx = np.stack([np.sort(rng.choice(64, t, replace=False)) for _ in range(m)]).astype(np.uint32)
    shares = np.frombuffer(b"".join((sum(c * pow(int(i) + 1, k, L) for k, c in enumerate(coeffs[g])) % L).to_bytes(32, "little")
                                    for g in range(m) for i in idx[g]), dtype=np.uint8).reshape(m * t, 32)
    _, share_pts = engine.mul_base(shares, want_ext=True)
    lam = engine.lagrange_coeffs(idx)
    got = engine.lincomb(lam, pts_ext=share_pts.reshape(m, t, 40), public=True)
    want = engine.mul_base(np.frombuffer(b"".join(c[0].to_bytes(32, "little") for c in coeffs), dtype=np.uint8).reshape(m, 32))
    assert np.array_equal(got, want)
    assert bytes(got[3]) == oracle.mul_base(coeffs[3][0].to_bytes(32, "little"))


def test_pripoly_shares_on_the_gpu(engine, oracle):
    """kyb_pripoly_eval_batch (PriPoly::eval / shares, poly.rs:133-152) == the oracle == Python integers: one lane per evaluation and chains cut
    into segments (few evaluations of a long polynomial), several polynomials in one call, unreduced coefficients, index 2^32 - 2; and the
    dealer's identity  share_i * B == PubPoly::eval(i)  of the committed polynomial"""
    L = synth.L
    rng = np.random.default_rng(77)
    for m, t, k in ((1, 1, 1), (1, 2, 5), (1, 43, 64), (3, 17, 9), (1, 683, 1024), (2, 683, 40), (1, 3000, 3), (1, 8, 70000)):
        coeffs = synth.scalars(m * t, 1000 + t).reshape(m, t, 32)
        coeffs[0, t // 2] = synth.raw256(1, 5)[0]                       # one unreduced value
        idx = rng.integers(0, 1 << 16, k, dtype=np.uint64).astype(np.uint32)
        idx[0] = 0
        if k > 2:
            idx[1], idx[2] = 0xfffffffe, k - 1
        got = engine.pripoly_eval(coeffs, idx)
        assert got.shape == (m, k, 32)
        for g in range(m):
            ints = [int.from_bytes(bytes(c), "little") for c in coeffs[g]]
            for i in sorted({0, 1, 2, k // 2, k - 1} & set(range(k))):
                x = int(idx[i]) + 1
                want = sum(c * pow(x, j, L) for j, c in enumerate(ints)) % L
                assert int.from_bytes(bytes(got[g, i]), "little") == want, (m, t, k, g, i)
                if t <= 700:
                    assert bytes(got[g, i]) == oracle.pripoly_eval(coeffs[g], int(idx[i])), (m, t, k, g, i)
    # one polynomial given as (t, 32): shares p(1) .. p(n), and the commitments' evaluation agrees with them
    t, n = 21, 64
    coeffs = synth.scalars(t, 2024)
    shares = engine.pripoly_eval(coeffs, np.arange(n, dtype=np.uint32))
    assert shares.shape == (n, 32)
    _, commits = engine.mul_base(coeffs, want_ext=True)
    assert np.array_equal(engine.mul_base(shares), engine.pubpoly_eval(commits, np.arange(n, dtype=np.uint32)))
