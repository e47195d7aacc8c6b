"""Every device-pointer entry point of include/kyber_ed25519.h that the other tests reach only through its host-pointer twin: the same inputs as
torch tensors on the GPU, called through the C ABI, compared byte for byte with the twin's output (the twins themselves are compared with the oracle
in test_gpu_parity.py / test_gpu_dkg_shape.py / test_gpu_wire_format.py) and, for a sample, with the oracle directly."""
import ctypes

import numpy as np
import pytest
import torch

import kyber_rs_amd
import synth

pytestmark = pytest.mark.gpu
ck = kyber_rs_amd._check
# the operands are torch tensors produced on the device's null stream and the results are read through it: the raw C-ABI calls below name
# that stream (KYB_STREAM_LEGACY) — NULL would be the engine's own non-blocking stream, ordered with neither (include/kyber_ed25519.h)
NULL_STREAM = ctypes.c_void_p(kyber_rs_amd.STREAM_LEGACY)


@pytest.fixture(scope="module")
def engine():
    return kyber_rs_amd.Engine(0)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def dp(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def empty(shape, dtype):
    return torch.empty(shape, dtype=dtype, device="cuda:0")


def host(t):
    return t.cpu().numpy()


@pytest.mark.parametrize("n", [5, 700])
def test_signing_entry_points(engine, oracle, n):
    lib = engine.lib
    seeds = synth.scalars(n, 1201)
    raw = synth.messages(n, 1202)
    msgs = kyber_rs_amd.pack_messages(raw)
    sig_h, pub_h = engine.eddsa_sign(seeds, msgs, want_pub=True)
    assert bytes(sig_h[0]) == oracle.eddsa_sign(bytes(seeds[0]), raw[0]) and bytes(pub_h[n - 1]) == oracle.eddsa_expand(bytes(seeds[n - 1]))[2]
    d_seeds, d_blob, d_off = dev(seeds), dev(msgs.blob if msgs.blob.size else np.zeros(1, np.uint8)), dev(msgs.off)
    sig, pub = empty((n, 64), torch.uint8), empty((n, 32), torch.uint8)
    ck(lib.kyb_eddsa_sign_batch_dev(dp(d_seeds), dp(d_blob), dp(d_off), n, dp(sig), dp(pub), NULL_STREAM), "kyb_eddsa_sign_batch_dev")
    engine.sync()
    assert np.array_equal(host(sig), sig_h) and np.array_equal(host(pub), pub_h)
    sig2 = empty((n, 64), torch.uint8)
    ck(lib.kyb_eddsa_sign_keyed_batch_dev(dp(d_seeds), dp(pub), dp(d_blob), dp(d_off), n, dp(sig2), NULL_STREAM), "kyb_eddsa_sign_keyed_batch_dev")
    engine.sync()
    assert np.array_equal(host(sig2), sig_h)
    assert np.array_equal(engine.verify(pub_h, msgs, sig_h, 0), np.zeros(n, dtype=np.uint8))
    # schnorr with the signer's stored key
    x, k = synth.scalars(n, 1203), synth.scalars(n, 1204, b"k")
    pubs = engine.mul_base(x)
    want = engine.schnorr_sign(x, k, msgs, pubs=pubs)
    sig3 = empty((n, 64), torch.uint8)
    d_x, d_pubs, d_k = dev(x), dev(pubs), dev(k)         # (kept alive: the call is asynchronous)
    ck(lib.kyb_schnorr_sign_keyed_batch_dev(dp(d_x), dp(d_pubs), dp(d_k), dp(d_blob), dp(d_off), n, dp(sig3), NULL_STREAM), "kyb_schnorr_sign_keyed_batch_dev")
    engine.sync()
    assert np.array_equal(host(sig3), want)
    assert bytes(want[n - 1]) == oracle.schnorr_sign(bytes(x[n - 1]), bytes(k[n - 1]), raw[n - 1])


@pytest.mark.parametrize("m,t,k", [(3, 7, 4), (40, 43, 1), (2, 300, 50)])
def test_polynomial_entry_points(engine, oracle, m, t, k):
    lib = engine.lib
    commits = engine.mul_base(synth.scalars(m * t, 1300 + t), ext_only=True).reshape(m, t, 40)
    commits_enc = engine.encode(commits.reshape(-1, 40)).reshape(m, t, 32)
    idx = np.random.default_rng(t).integers(0, 1024, (m, k), dtype=np.uint32)
    d_c, d_ce, d_idx = dev(commits), dev(commits_enc), dev(idx)
    # PubPoly::eval, one polynomial / m polynomials / from the wire
    want1, want1x = engine.pubpoly_eval(commits[0], idx[0], want_ext=True)
    enc, ext = empty((k, 32), torch.uint8), empty((k, 40), torch.int32)
    ck(lib.kyb_pubpoly_eval_batch_dev(dp(d_c), t, dp(d_idx), k, 1023, dp(enc), dp(ext), NULL_STREAM), "kyb_pubpoly_eval_batch_dev")
    engine.sync()
    assert np.array_equal(host(enc), want1) and np.array_equal(engine.encode(host(ext)), want1)
    wantm = engine.pubpoly_eval_multi(commits, idx)
    encm = empty((m, k, 32), torch.uint8)
    ck(lib.kyb_pubpoly_eval_multi_batch_dev(dp(d_c), t, m, dp(d_idx), k, 1023, dp(encm), None, NULL_STREAM), "kyb_pubpoly_eval_multi_batch_dev")
    engine.sync()
    assert np.array_equal(host(encm), wantm)
    wante, ok_h = engine.pubpoly_eval_multi_enc(commits_enc, idx)
    ence, oke = empty((m, k, 32), torch.uint8), empty((m, t), torch.uint8)
    ck(lib.kyb_pubpoly_eval_multi_enc_batch_dev(dp(d_ce), t, m, dp(d_idx), k, 1023, dp(ence), None, dp(oke), NULL_STREAM), "kyb_pubpoly_eval_multi_enc_batch_dev")
    engine.sync()
    assert np.array_equal(host(ence), wante) and np.array_equal(host(oke), ok_h) and np.array_equal(wante, wantm)
    # the oracle on one evaluation: Horner over the commitments (poly.rs:457-469)
    xi = (int(idx[m - 1, k - 1]) + 1).to_bytes(32, "little")
    v = oracle.null()
    for j in range(t - 1, -1, -1):
        v = oracle.add(oracle.mul_ext(xi, v), commits[m - 1, j])
    assert oracle.encode(v) == bytes(wantm[m - 1, k - 1])
    # sums, from limbs and from the wire (both layouts)
    wants = engine.sum_points(commits)
    encs, exts = empty((m, 32), torch.uint8), empty((m, 40), torch.int32)
    ck(lib.kyb_sum_batch_dev(dp(d_c), m, t, dp(encs), dp(exts), NULL_STREAM), "kyb_sum_batch_dev")
    engine.sync()
    assert np.array_equal(host(encs), wants) and np.array_equal(engine.encode(host(exts)), wants)
    for item_major in (0, 1):
        src = commits_enc if not item_major else np.ascontiguousarray(commits_enc.transpose(1, 0, 2))
        wantw, okw = engine.sum_points_enc(src, item_major=bool(item_major))
        encw, okd = empty((m, 32), torch.uint8), empty(src.shape[:2], torch.uint8)
        d_src = dev(src)
        ck(lib.kyb_sum_enc_batch_dev(dp(d_src), m, t, item_major, dp(encw), None, dp(okd), NULL_STREAM), "kyb_sum_enc_batch_dev")
        engine.sync()
        assert np.array_equal(host(encw), wantw) and np.array_equal(host(okd), okw) and np.array_equal(wantw, wants)
    # the verifier's side of a DKG round in one call (device flavour: the index also as m copies in device memory)
    ev_h, sums_h, okr_h = engine.dkg_verify_round_enc(commits_enc, 17)
    ev, sm, okr = empty((m, 32), torch.uint8), empty((t, 32), torch.uint8), empty((m, t), torch.uint8)
    d_index = dev(np.full((m,), 17, dtype=np.uint32))
    ck(lib.kyb_dkg_verify_round_enc_dev(dp(d_ce), t, m, dp(d_index), 17, dp(ev), None, dp(sm), None, dp(okr), NULL_STREAM), "kyb_dkg_verify_round_enc_dev")
    engine.sync()
    assert np.array_equal(host(ev), ev_h) and np.array_equal(host(sm), sums_h) and np.array_equal(host(okr), okr_h)
    assert lib.kyb_dkg_verify_round_enc_dev(dp(d_ce), t, m, None, 17, dp(ev), None, dp(sm), None, dp(okr), NULL_STREAM) == -2      # KYB_E_BAD_ARG: index_dev is required


@pytest.mark.parametrize("m,t", [(4, 9), (33, 43)])
def test_scalar_side_and_linear_combinations(engine, oracle, m, t):
    lib = engine.lib
    rng = np.random.default_rng(m * t)
    # Lagrange coefficients and private shares
    idx = np.stack([rng.choice(2048, t, replace=False).astype(np.uint32) for _ in range(m)])
    want_l = engine.lagrange_coeffs(idx)
    out_l = empty((m, t, 32), torch.uint8)
    d_idx = dev(idx)
    ck(lib.kyb_lagrange_coeffs_batch_dev(dp(d_idx), m, t, dp(out_l), NULL_STREAM), "kyb_lagrange_coeffs_batch_dev")
    engine.sync()
    assert np.array_equal(host(out_l), want_l)
    coeffs = synth.scalars(m * t, 1400 + t).reshape(m, t, 32)
    k_idx = np.arange(0, 12, dtype=np.uint32)
    want_s = engine.pripoly_eval(coeffs, k_idx)
    out_s = empty((m, 12, 32), torch.uint8)
    d_coeffs, d_kidx = dev(coeffs), dev(k_idx)
    ck(lib.kyb_pripoly_eval_batch_dev(dp(d_coeffs), m, t, dp(d_kidx), 12, dp(out_s), NULL_STREAM), "kyb_pripoly_eval_batch_dev")
    engine.sync()
    assert np.array_equal(host(out_s), want_s)
    # sum_i lambda_i * share_i = the secret coefficient (poly.rs:244-290), through the device entry point of the linear combination with public scalars
    pts = engine.mul_base(synth.scalars(m * t, 1500 + t), ext_only=True).reshape(m, t, 40)
    for fn_h, fn_d, public in ((lib.kyb_lincomb_batch, lib.kyb_lincomb_batch_dev, False), (lib.kyb_lincomb_public_batch, lib.kyb_lincomb_public_batch_dev, True)):
        want = engine.lincomb(want_l, pts_ext=pts, public=public)
        out = empty((m, 32), torch.uint8)
        d_sc, d_pts = dev(want_l), dev(pts)
        ck(fn_d(dp(d_sc), None, dp(d_pts), 0, m, t, dp(out), None, None, NULL_STREAM), "kyb_lincomb(_public)_batch_dev")
        engine.sync()
        assert np.array_equal(host(out), want), public
        acc = oracle.null()
        for j in range(t):
            acc = oracle.add(acc, oracle.mul_ext(bytes(want_l[0, j]), pts[0, j]))
        assert oracle.encode(acc) == bytes(want[0])


def test_encode_decode_and_table_image_entry_points(engine):
    lib = engine.lib
    n = 1500
    ext = engine.mul_base(synth.scalars(n, 1601), ext_only=True)
    enc = engine.encode(ext)
    out = empty((n, 32), torch.uint8)
    d_ext = dev(ext)
    engine.encode_dev(d_ext, out)
    back, ok = empty((n, 40), torch.int32), empty((n,), torch.uint8)
    bad = enc.copy()
    bad[::9] = np.frombuffer(bytes([2]) + bytes(31), dtype=np.uint8)
    d_bad = dev(bad)
    engine.decode_dev(d_bad, back, ok)
    engine.sync()
    assert np.array_equal(host(out), enc)
    ext_h, ok_h = engine.decode(bad)
    assert np.array_equal(host(ok), ok_h) and np.array_equal(host(back), ext_h) and not ok_h[0] and ok_h[1]
    image = engine.base_table()
    d_img = empty((image.size,), torch.uint8)
    engine.base_table_export_dev(d_img)
    engine.sync()
    assert np.array_equal(host(d_img), image.view(np.uint8).reshape(-1))
    other = kyber_rs_amd.Engine(0, build_table=False, private=True)
    try:
        other.base_table_import_dev(d_img)
        other.sync()
        s = synth.scalars(64, 1602)
        assert np.array_equal(other.mul_base(s), engine.mul_base(s))
    finally:
        other.close()
