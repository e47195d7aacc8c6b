"""Host-pointer calls on PAGE-LOCKED caller arrays (kyb_host_alloc): a zero-copy call uses them where they lie (host.in_place, csrc/engine.hip
pinned_dev_ptr) instead of copying them into the context's buffer.  Same bytes as the copying path and as the oracle; arrays the path must NOT take
in place (misaligned slices, pageable memory mixed in) fall back without a difference."""
import ctypes

import numpy as np
import pytest

import kyber_rs_amd
import oracle_lib
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    return kyber_rs_amd.Engine(0)


def _pinned_copy(engine, a, lead=0):
    """a copy of `a` in page-locked memory; lead > 0: starting `lead` bytes into the allocation (misaligned for the in-place path)"""
    flat = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    raw = engine.pinned_array((flat.size + lead,), np.uint8)
    raw[lead:] = flat
    return raw[lead:].view(a.dtype).reshape(a.shape)


@pytest.mark.parametrize("n", [2048, 8192, 20000])      # one item per wavefront / two lanes per item (the DKG-sized paths) / one lane per item
def test_pinned_arrays_are_used_in_place_with_the_same_results(engine, n):
    orc = oracle_lib.Oracle()
    lib = engine.lib
    s, k = synth.scalars(n, 901), synth.scalars(n, 902, b"k")
    enc, ext = engine.mul_base(s, want_ext=True)
    raw = synth.messages(n, 903)
    msgs = kyber_rs_amd.pack_messages(raw)
    sigs = engine.schnorr_sign(s, k, msgs)
    sigs[::7, 2] ^= 1
    want_mul = engine.mul(k, pts_enc=enc)
    want_st = engine.verify(enc, msgs, sigs, 1)
    idx = np.random.default_rng(5).choice(n, 24, replace=False)
    for i in idx:
        assert bytes(want_mul[i]) == orc.mul(bytes(k[i]), ext[i])
        assert int(want_st[i]) == orc.verify(1, bytes(enc[i]), raw[int(i)], bytes(sigs[i]))
    P = kyber_rs_amd._ptr
    for lead in (0, 8):                                   # 8: not 16-byte aligned -> the copying path, silently
        pk, penc, pext, psig = (_pinned_copy(engine, a, lead) for a in (k, enc, ext, sigs))
        blob, off = _pinned_copy(engine, msgs.blob, 0), _pinned_copy(engine, msgs.off, 0)
        out = engine.pinned_array((n, 32), np.uint8)
        st = engine.pinned_array((n,), np.uint8)
        ok = engine.pinned_array((n,), np.uint8)
        for in_place in (1, 0, 1):
            engine.set_option("host.in_place", in_place)
            out[:] = 0; ok[:] = 0
            kyber_rs_amd._check(lib.kyb_mul_batch(P(pk), P(penc), None, n, P(out), None, P(ok)), "kyb_mul_batch")
            assert np.array_equal(out, want_mul) and ok.all(), (lead, in_place)
            out[:] = 0
            kyber_rs_amd._check(lib.kyb_mul_batch(P(pk), None, P(pext), n, P(out), None, None), "kyb_mul_batch")
            assert np.array_equal(out, want_mul), (lead, in_place)
            out[:] = 0
            kyber_rs_amd._check(lib.kyb_mul_base_batch(P(pk), n, P(out), None), "kyb_mul_base_batch")
            assert bytes(out[int(idx[0])]) == orc.mul_base(bytes(k[int(idx[0])])), (lead, in_place)
            st[:] = 0xee
            kyber_rs_amd._check(lib.kyb_verify_batch(P(penc), P(blob), P(off), P(psig), n, 1, P(st)), "kyb_verify_batch")
            assert np.array_equal(st, want_st), (lead, in_place)
            # pageable and page-locked arrays in one call
            out2 = np.zeros((n, 32), dtype=np.uint8)
            kyber_rs_amd._check(lib.kyb_mul_batch(P(k), P(penc), None, n, P(out2), None, None), "kyb_mul_batch")
            assert np.array_equal(out2, want_mul), (lead, in_place)
    engine.set_option("host.in_place", 1)
    # the caller's secret scalars were read where they lay and are still there (nothing of the caller's is wiped)
    assert np.array_equal(pk, k)


def _guarded(engine, a, guard=64):
    """a page-locked copy of `a` with `guard` bytes of 0xA5 on either side of it inside the same allocation (start kept 16-byte aligned)"""
    flat = np.ascontiguousarray(a).view(np.uint8).reshape(-1)
    raw = engine.pinned_array((flat.size + 2 * guard,), np.uint8)
    raw[:] = 0xA5
    raw[guard:guard + flat.size] = flat
    return raw, raw[guard:guard + flat.size].view(a.dtype).reshape(a.shape)


@pytest.mark.parametrize("n", [2049, 8191, 20001])      # odd sizes: the last 16-byte word / the last lanes of a shared inversion are partial
def test_in_place_arrays_are_never_read_or_written_past_their_end_and_may_alias(engine, n):
    """(1) guard bytes behind (and in front of) every page-locked caller array stay untouched at odd n; (2) an output array that IS an
    input array of the same call (in-place update: out_ext == pts_ext, out == a) gives the results of separate arrays — the engine stages the
    input when two in-place arrays share bytes and one is written."""
    orc = oracle_lib.Oracle()
    lib = engine.lib
    P = kyber_rs_amd._ptr
    k = synth.scalars(n, 911)
    enc, ext = engine.mul_base(synth.scalars(n, 912), want_ext=True)
    want_ext_enc = engine.mul(k, pts_ext=ext)
    want_sum = engine.encode(engine.add(ext, ext))
    idx = [0, n // 2, n - 1]
    for i in idx:
        assert bytes(want_ext_enc[i]) == orc.mul(bytes(k[i]), ext[i])
    engine.set_option("host.in_place", 1)
    rk, pk = _guarded(engine, k)
    rx, px = _guarded(engine, ext)
    ro, po = _guarded(engine, np.zeros((n, 32), np.uint8))
    rx2, px2 = _guarded(engine, np.zeros((n, 40), np.int32))
    kyber_rs_amd._check(lib.kyb_mul_batch(P(pk), None, P(px), n, P(po), P(px2), None), "kyb_mul_batch")
    assert np.array_equal(po, want_ext_enc)
    assert np.array_equal(engine.encode(np.array(px2)), want_ext_enc)
    for raw in (rk, rx, ro, rx2):
        assert (raw[:64] == 0xA5).all() and (raw[-64:] == 0xA5).all(), "guard bytes around a page-locked caller array were written"
    assert np.array_equal(pk, k) and np.array_equal(px, ext)
    # in-place update: the products land on top of the operands
    kyber_rs_amd._check(lib.kyb_mul_batch(P(pk), None, P(px), n, P(po), P(px), None), "kyb_mul_batch (out_ext == pts_ext)")
    assert np.array_equal(po, want_ext_enc) and np.array_equal(engine.encode(np.array(px)), want_ext_enc)
    assert (rx[:64] == 0xA5).all() and (rx[-64:] == 0xA5).all()
    # a + a -> a
    px[:] = ext
    kyber_rs_amd._check(lib.kyb_add_batch(P(px), P(px), n, P(px), 0), "kyb_add_batch (out == a == b)")
    assert np.array_equal(engine.encode(np.array(px)), want_sum)
    # shifted overlap: the output starts one record into the input
    big = engine.pinned_array(((n + 1) * 40,), np.int32)
    big[:n * 40] = ext.reshape(-1)
    src, dst = big[:n * 40].reshape(n, 40), big[40:].reshape(n, 40)
    kyber_rs_amd._check(lib.kyb_add_batch(P(src), P(src), n, P(dst), 0), "kyb_add_batch (out overlaps a, shifted)")
    assert np.array_equal(engine.encode(np.array(dst)), want_sum)
