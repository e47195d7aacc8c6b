"""Contexts and groups of the C ABI on the one GPU the test box has:

* kyb_group_create([0, 0, 0]): three contexts on device 0, table image built once, moved and checksum-validated;
  kyb_group_*_batch shard [floor(n r / G), floor(n (r+1) / G)) and every output is checked against the oracle;
* private contexts used from fresh threads through the device-pointer API (each entry makes its context's device current);
* an imported table image that fails its checksum is refused;
* 100 short-lived caller streams: scratch slots are recycled / released, no KYB_E_NOMEM;
* kyb_set_option racing with launches (options are atomics; profile begin/read take a lock).
The N > 1-device path proper (RCCL broadcast between distinct GPUs) needs the driver's 8-GPU node; everything above the
transport is what runs here."""
import ctypes
import threading

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def test_group_of_three_contexts_on_one_device(oracle):
    import kyber_rs_amd
    grp = kyber_rs_amd.Group([0, 0, 0])
    try:
        assert grp.size == 3
        assert grp.transport == "host-copy"            # the list repeats a device: RCCL is not attempted
        n = 1000                                        # shards 333 / 333 / 334
        s = synth.raw256(n, 31)
        assert np.array_equal(grp.mul_base(s), oracle.mul_base_batch(s, nthreads=8))
        pts = oracle.mul_base_ext_batch(synth.scalars(n, 32, b"point"))
        enc, ok = grp.mul(s, pts_ext=pts)
        assert ok.all() and np.array_equal(enc, oracle.mul_batch(s, pts, nthreads=8))
        pts_enc = np.stack([np.frombuffer(oracle.encode(p), dtype=np.uint8) for p in pts])
        enc2, ok2 = grp.mul(s, pts_enc=pts_enc)
        assert ok2.all() and np.array_equal(enc2, enc)
        # ragged messages: every shard gets its offsets rebased to its own first message
        msgs = [bytes([i & 255]) * (i % 97) for i in range(n)]
        x, k = synth.scalars(n, 33, b"x"), synth.scalars(n, 33, b"k")
        sig = grp.schnorr_sign(x, k, msgs)
        assert np.array_equal(sig, oracle.schnorr_sign_batch(x, k, msgs, nthreads=8))
        pub = grp.mul_base(x)
        assert not grp.verify(pub, msgs, sig, 1).any()
        bad = sig.copy(); bad[::7, 40] ^= 1
        st = grp.verify(pub, msgs, bad, 1)
        assert np.array_equal(st, oracle.verify_batch(1, pub, msgs, bad, nthreads=8))
        # fewer items than ranks, and none at all
        assert np.array_equal(grp.mul_base(s[:2]), oracle.mul_base_batch(s[:2]))
        assert grp.mul_base(s[:0]).shape == (0, 32)
        # every rank holds the same validated image
        imgs = [grp.engine(r).base_table().tobytes() for r in range(3)]
        assert imgs[0] == imgs[1] == imgs[2]
    finally:
        grp.close()


def test_private_contexts_from_fresh_threads_dev_api(oracle):
    """two private contexts next to the default one; each is driven through the device-pointer API from a thread that
    has never touched HIP before (hipSetDevice inside every entry point)"""
    import torch
    import kyber_rs_amd
    dev = torch.device("cuda", 0)
    n = 3000
    s_np = synth.scalars(n, 41)
    want = oracle.mul_base_batch(s_np, nthreads=8)
    engines = [kyber_rs_amd.Engine(0, private=True) for _ in range(2)]
    results, errors = {}, []

    def work(i):
        try:
            eng = engines[i]
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                sc = torch.from_numpy(s_np).to(dev)
                out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
                for _ in range(5):
                    eng.mul_base_dev(sc, out_enc=out, stream=st.cuda_stream)
                eng.sync(st.cuda_stream)
                results[i] = out.cpu().numpy()
                eng.stream_release(st.cuda_stream)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for i in range(2):
        assert np.array_equal(results[i], want)
    for e in engines:
        e.close()


def test_host_calls_in_flight_share_the_chip(engine, oracle):
    """Six threads with a context each issue mid-size host-pointer calls at the same time: the small-batch thresholds shrink with the number of
    calls in flight (coop.share_by_load; the routing of a call then depends on what else is running), and every call still returns the bytes
    of a call made alone — with the option on and off.  One lone call keeps its one-item-per-wavefront kernels."""
    import kyber_rs_amd
    n = 3000
    s = synth.scalars(n, 71); k = synth.scalars(n, 72, b"k")
    pts = oracle.mul_base_ext_batch(synth.scalars(n, 73, b"p"))
    msgs = kyber_rs_amd.pack_messages(synth.messages(n, 74))
    want_base = oracle.mul_base_batch(s, nthreads=8)
    want_mul = oracle.mul_batch(k, pts, nthreads=8)
    sig = engine.schnorr_sign(s, k, msgs)
    bad = sig.copy(); bad[::5, 3] ^= 1
    want_st = engine.verify(want_base, msgs, bad, 1)
    engine.profile_begin(8)
    assert np.array_equal(engine.mul(k[:2000], pts_ext=pts[:2000]), want_mul[:2000])
    assert [nm for nm, _ in engine.profile_read(8)][0] == "k_mul_coop"           # alone: 2,000 items take the latency kernels
    engine.profile_begin(0)
    for share in (1, 0):
        engines = [kyber_rs_amd.Engine(0, private=True) for _ in range(6)]
        errors = []

        def work(i):
            try:
                e = engines[i]
                e.set_option("coop.share_by_load", share)
                for r in range(12):
                    m = n - 37 * ((i + r) % 5)
                    assert np.array_equal(e.mul_base(s[:m]), want_base[:m]), ("mul_base", i, r)
                    assert np.array_equal(e.mul(k[:m], pts_ext=pts[:m]), want_mul[:m]), ("mul", i, r)
                    assert np.array_equal(e.schnorr_sign(s[:m], k[:m], msgs[:m]), sig[:m]), ("sign", i, r)
                    assert np.array_equal(e.verify(want_base[:m], msgs[:m], bad[:m], 1), want_st[:m]), ("verify", i, r)
            except Exception as ex:  # noqa: BLE001
                errors.append(repr(ex))

        th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in engines:
            e.close()
        assert not errors, errors[:3]


def test_corrupted_table_image_is_refused(engine):
    import kyber_rs_amd
    img = engine.base_table()
    eng2 = kyber_rs_amd.Engine(0, build_table=False, private=True)
    try:
        bad = img.copy()
        bad[200000] ^= 0x10                             # one flipped bit in the radix-64 image
        with pytest.raises(kyber_rs_amd.KyberHipError, match="checksum"):
            eng2.base_table_import(bad)
        with pytest.raises(kyber_rs_amd.KyberHipError, match="KYB_E_NOT_INIT"):
            eng2.mul_base(synth.scalars(1, 1))          # the refused image was not installed
        trunc = img.copy()
        trunc[-4096:] = 0                               # a truncated transfer
        with pytest.raises(kyber_rs_amd.KyberHipError, match="checksum"):
            eng2.base_table_import(trunc)
        eng2.base_table_import(img)                     # the intact image is accepted
        s = synth.scalars(64, 2)
        assert np.array_equal(eng2.mul_base(s), engine.mul_base(s))
    finally:
        eng2.close()


def test_hundred_short_lived_streams(engine, oracle):
    """a service that creates and destroys streams: slots are recycled (LRU, behind the slot's last launch) or released
    explicitly; a recycled stream handle never sees another stream's unfinished scratch"""
    import torch
    dev = torch.device("cuda", 0)
    n = 700
    s_np = synth.scalars(n, 51)
    want = oracle.mul_base_batch(s_np, nthreads=8)
    sc = torch.from_numpy(s_np).to(dev)
    torch.cuda.synchronize()
    outs = []
    for i in range(100):
        st = torch.cuda.Stream(device=dev)
        out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
        st.wait_stream(torch.cuda.current_stream())
        engine.mul_base_dev(sc, out_enc=out, stream=st.cuda_stream)
        if i % 3 == 0:
            engine.stream_release(st.cuda_stream)       # explicit release waits for the launch, then frees the slot
        outs.append((st, out))
        if i % 10 == 9:
            for st_, out_ in outs:
                st_.synchronize()
                assert np.array_equal(out_.cpu().numpy(), want)
            outs = []


def test_set_option_and_profiling_race_with_launches(xengine, oracle):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    n = 2048
    s = synth.scalars(n, 61)
    want = oracle.mul_base_batch(s, nthreads=8)
    stop = threading.Event()
    errors = []

    def flip():
        i = 0
        while not stop.is_set():
            engine.set_option("mul_base.small_chunks", 1 + (i & 1))
            engine.set_option("verify.overlap", i & 1)
            engine.profile_begin(16 if i % 5 == 0 else 0)
            if i % 5 == 1:
                engine.profile_read(16)
            i += 1

    def run():
        try:
            for _ in range(40):
                if not np.array_equal(engine.mul_base(s), want):
                    errors.append("parity")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    t1, t2 = threading.Thread(target=flip), threading.Thread(target=run)
    t1.start(); t2.start()
    t2.join()
    stop.set()
    t1.join()
    engine.profile_begin(0)
    engine.set_option("mul_base.small_chunks", 2)
    engine.set_option("verify.overlap", 1)
    assert not errors, errors


def test_encode_batched_matches_per_item_encode(xengine, oracle):
    engine = xengine          # variant kernels / selectors: the cross-check build (tests/conftest.py)
    """kyb_encode_batch with one shared inversion per 8 points == per-point inversion == the oracle, incl. Z = 0 garbage
    (the reference's 0^(p-2) = 0 answer) isolated from its neighbours, ragged sizes"""
    for n in (1, 7, 8, 9, 1000, 4099):
        pts = oracle.mul_base_ext_batch(synth.scalars(min(n, 300), 70 + n, b"point"))
        pts = np.tile(pts, ((n + len(pts) - 1) // len(pts), 1))[:n].copy()
        if n >= 9:
            pts[3, 20:30] = 0                           # Z = 0
        engine.set_option("encode.batched", 1)
        a = engine.encode(pts)
        engine.set_option("encode.batched", 0)
        b = engine.encode(pts)
        engine.set_option("encode.batched", 1)
        assert np.array_equal(a, b)
        idx = list(range(min(n, 40)))
        assert [bytes(a[i]) for i in idx] == [oracle.encode(pts[i]) for i in idx]


def test_one_inversion_per_wavefront_matches_the_per_lane_forms(xengine, oracle):
    engine = xengine          # finish.four is a selector of the cross-check build (tests/conftest.py); the product runs its default, 2
    """what closes a mid-size launch (engine.hip finish_wave / finish_four): one inversion per WAVEFRONT with Montgomery's trick across the 64
    lanes (k_finish_wave, finish.four = 2, the default) == one per 4 points of a lane (1) == one per 8 (0) == the oracle: marshal_binary of
    projective points, and the end of fixed-base, variable-base and signing calls (encodings and affine limbs); sizes whose last wavefront has
    1, 63 and 64 live lanes; Z = 0 garbage in the first, a middle and the last lane of a wavefront gets the reference's 0^(p-2) = 0 and leaves
    the other 63 points of its wavefront alone"""
    assert engine.get_option("finish.four") == 2
    dm = engine.get_option("coop.decode_max_items")
    base = oracle.mul_base_ext_batch(synth.scalars(300, 910, b"point"))
    proj300 = np.stack([oracle.add(a_, b_) for a_, b_ in zip(base, np.roll(base, 1, axis=0))])        # Z != 1
    want300 = np.stack([np.frombuffer(oracle.encode(e), dtype=np.uint8) for e in proj300])
    try:
        for n in (dm + 1, dm + 63, dm + 64, 4099, 20000, 64 * 8 * engine.get_option("device.cus") - 3):
            reps = (n + 299) // 300
            pts, want = np.tile(proj300, (reps, 1))[:n].copy(), np.tile(want300, (reps, 1))[:n].copy()
            for bad in (0, 64 + 37, n - 1, n - 64):
                pts[bad, 20:30] = 0                              # Z = 0
                want[bad] = 0                                    # (x, y) = (0, 0): the all-zero encoding
            got = {}
            for four in (2, 1, 0):
                engine.set_option("finish.four", four)
                got[four] = engine.encode(pts)
            assert np.array_equal(got[2], got[1]) and np.array_equal(got[2], got[0]), n
            assert np.array_equal(got[2], want), n
        # the end of whole calls: above every one-item-per-wavefront size, below a wavefront per SIMD
        n = engine.get_option("coop.ladder_max_items") + 77
        s, k = synth.raw256(n, 911), synth.scalars(n, 912, b"k")
        x = s.copy(); x[:, 31] &= 0x7f
        p = np.tile(proj300, ((n + 299) // 300, 1))[:n].copy()
        msgs = synth.messages(n, 913)
        res = {}
        for four in (2, 1):
            engine.set_option("finish.four", four)
            res[four] = (engine.mul_base(s, want_ext=True), engine.mul(s, pts_ext=p, want_ext=True), engine.schnorr_sign(x, k, msgs))
        for a_, b_ in zip(res[2], res[1]):
            if isinstance(a_, tuple):
                assert np.array_equal(a_[0], b_[0]) and np.array_equal(a_[1], b_[1])
            else:
                assert np.array_equal(a_, b_)
        assert np.array_equal(res[2][0][0], oracle.mul_base_batch(s, nthreads=8))
        assert np.array_equal(res[2][1][0], oracle.mul_batch(s, p, nthreads=8))
        assert np.array_equal(res[2][2], oracle.schnorr_sign_batch(x, k, msgs, nthreads=8))
    finally:
        engine.set_option("finish.four", 2)


def test_fixed_base_in_quarters_matches_one_lane_per_item(xengine, oracle):
    engine = xengine          # mul_base.quarters is a selector of the cross-check build (tests/conftest.py); the product runs its default, 1
    """mid-size fixed-base launches (engine.hip base_quarters): four wavefronts per 64 items, a quarter of the 43 windows each, the partial points
    added through the staging records (k_mul_base64_quarters) == one lane per item (k_mul_base64) == the oracle — sizes from the first above the
    one-item-per-wavefront kernels to the last that takes this form and one beyond, ragged last groups, the scalars whose digits sit at the
    window cuts (0, 1, L - 1, L, 2^66 - 1, 2^132, 2^198 +- 1, the a[31] quirk values), signing (two scalar arrays in one launch, the seam inside a group)
    and limbs out"""
    assert engine.get_option("mul_base.quarters") == 1
    cus = engine.get_option("device.cus")
    lo = 5 * cus                                        # (engine.hip COOP_BASE_TO_QUARTERS_PER_CU: the last size of the one-item-per-wavefront kernel)
    L = (1 << 252) + 27742317777372353535851937790883648493
    special = [0, 1, 2, 63, 64, L - 1, L, L + 1, (1 << 66) - 1, 1 << 66, 1 << 132, (1 << 198) - 1, (1 << 198) + 1, (1 << 255) - 19, (1 << 255), (1 << 256) - 1,
               0x7f << 248, 0x80 << 248, 0x8f << 248, 0x90 << 248, (0xff << 248) | 12345]
    try:
        for n in (lo + 1, lo + 64, 8192 + 37, 128 * cus, 128 * cus + 1):
            s = synth.raw256(n, 920 + n % 7)
            for j, v in enumerate(special):
                s[(j * 131) % n] = np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)
            want = oracle.mul_base_batch(s, nthreads=8)
            engine.set_option("mul_base.quarters", 1)
            enc1, ext1 = engine.mul_base(s, want_ext=True)
            engine.set_option("mul_base.quarters", 0)
            enc0, ext0 = engine.mul_base(s, want_ext=True)
            assert np.array_equal(enc1, want) and np.array_equal(enc0, want), n
            assert np.array_equal(ext1, ext0), n
        n = lo // 2 + 45                                   # 2 n items in the launch, the seam between nonces and keys inside a group
        x, k, msgs = synth.raw256(n, 925), synth.scalars(n, 926, b"k"), synth.messages(n, 927)
        x[:, 31] &= 0x7f
        want_sig = oracle.schnorr_sign_batch(x, k, msgs, nthreads=8)
        for q in (1, 0):
            engine.set_option("mul_base.quarters", q)
            assert np.array_equal(engine.schnorr_sign(x, k, msgs), want_sig), q
    finally:
        engine.set_option("mul_base.quarters", 1)


def test_four_lane_ladder_at_its_hand_over_sizes(engine, oracle):
    """variable base from points with the DEFAULT routing: the last size of the one-item-per-wavefront kernel (9 per CU) and the first of the four-lane
    ladder, the last of the four-lane ladder (64 per CU, one wavefront per SIMD) and the first of the two-lane one — every output against the oracle;
    unreduced scalars, projective and small-order operands in the mix"""
    cus = engine.get_option("device.cus")
    assert engine.get_option("ladder.quad_max_items") == 64 * cus
    nmax = 64 * cus + 1
    s = synth.raw256(nmax, 931); s[::3] = synth.scalars(len(s[::3]), 932)
    base = oracle.mul_base_ext_batch(synth.scalars(500, 933, b"point"))
    base[7] = oracle.null()
    base[11:400:13] = np.stack([oracle.add(a_, b_) for a_, b_ in zip(base[11:400:13], base[12:401:13])])      # Z != 1
    pts = np.tile(base, ((nmax + 499) // 500, 1))[:nmax].copy()
    want = oracle.mul_batch(s, pts, nthreads=8)
    for n in (9 * cus, 9 * cus + 1, 64 * cus - 2, 64 * cus, 64 * cus + 1):
        assert np.array_equal(engine.mul(s[:n], pts_ext=pts[:n]), want[:n]), n


def test_group_device_resident_shards_and_pool_reuse(oracle):
    """kyb_group_mul(_base)_batch_dev: per-rank device pointers, launches queued on every rank's own stream by ONE thread, kyb_group_sync;
    and the group's persistent worker threads serve many sharded host-pointer calls in a row (nothing is spawned per call)."""
    import torch
    import kyber_rs_amd
    grp = kyber_rs_amd.Group([0, 0, 0])
    try:
        dev = torch.device("cuda", 0)
        sizes = [70000, 1, 4097]                    # batch kernels, one-item kernels, in between
        sc_np = [synth.scalars(m, 50 + r) for r, m in enumerate(sizes)]
        pt_np = [oracle.mul_base_ext_batch(synth.scalars(min(m, 64), 60 + r, b"point")) for r, m in enumerate(sizes)]
        pt_np = [np.tile(p, ((m + len(p) - 1) // len(p), 1))[:m].copy() for p, m in zip(pt_np, sizes)]
        sc = [torch.from_numpy(a).to(dev) for a in sc_np]
        pts = [torch.from_numpy(a).to(dev) for a in pt_np]
        out = [torch.empty((m, 32), dtype=torch.uint8, device=dev) for m in sizes]
        outb = [torch.empty((m, 32), dtype=torch.uint8, device=dev) for m in sizes]
        torch.cuda.synchronize()
        for _ in range(2):
            grp.mul_dev(sc, pts_ext=pts, out_enc=out)
            grp.mul_base_dev(sc, out_enc=outb)
        grp.sync()
        for r in range(3):
            assert np.array_equal(out[r].cpu().numpy(), oracle.mul_batch(sc_np[r], pt_np[r], nthreads=8)), r
            assert np.array_equal(outb[r].cpu().numpy(), oracle.mul_base_batch(sc_np[r], nthreads=8)), r
        # wire encodings in, decode flags out
        penc = [torch.from_numpy(oracle.encode_batch(a)).to(dev) for a in pt_np]
        ok = [torch.zeros((m,), dtype=torch.uint8, device=dev) for m in sizes]
        for t in out:
            t.zero_()
        grp.mul_dev(sc, pts_enc=penc, out_enc=out, ok=ok)
        grp.sync()
        for r in range(3):
            assert bool(ok[r].all()) and np.array_equal(out[r].cpu().numpy(), oracle.mul_batch(sc_np[r], pt_np[r], nthreads=8)), r
        # 200 sharded host-pointer calls through the same three worker threads
        s = synth.scalars(9, 77)
        want = oracle.mul_base_batch(s)
        before = threading.active_count()
        for _ in range(200):
            assert np.array_equal(grp.mul_base(s), want)
        assert threading.active_count() == before
    finally:
        grp.close()


def test_group_rejects_decreasing_message_offsets(oracle):
    """a decrease in msg_off wrapped in the unsigned shard offsets and passed the per-shard check; the group entry points now
    validate the whole array first, like the single-device calls."""
    import kyber_rs_amd
    lib = kyber_rs_amd.load_library()
    grp = kyber_rs_amd.Group([0, 0])
    try:
        n = 6
        x, k = synth.scalars(n, 81, b"x"), synth.scalars(n, 81, b"k")
        blob = np.zeros(1024, dtype=np.uint8)
        off = np.array([0, 10, 100, 40, 50, 60, 70], dtype=np.uint32)          # 100 -> 40: inside rank 0's shard ... and across shards below
        sig = np.zeros((n, 64), dtype=np.uint8)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        assert lib.kyb_group_schnorr_sign_batch(grp.handle, p(x), p(k), p(blob), p(off), n, p(sig)) == -2
        assert b"non-decreasing" in lib.kyb_last_error()
        off2 = np.array([0, 10, 20, 30, 5, 6, 7], dtype=np.uint32)             # the decrease sits exactly on the shard boundary
        assert lib.kyb_group_schnorr_sign_batch(grp.handle, p(x), p(k), p(blob), p(off2), n, p(sig)) == -2
        pub = np.zeros((n, 32), dtype=np.uint8)
        st = np.zeros(n, dtype=np.uint8)
        assert lib.kyb_group_verify_batch(grp.handle, p(pub), p(blob), p(off), p(sig), n, 1, p(st)) == -2
        assert not sig.any()                                                   # nothing was written
        good = np.arange(0, 8 * (n + 1), 8, dtype=np.uint32)
        assert lib.kyb_group_schnorr_sign_batch(grp.handle, p(x), p(k), p(blob), p(good), n, p(sig)) == 0
        assert np.array_equal(sig, oracle.schnorr_sign_batch(x, k, [bytes(8)] * n))
    finally:
        grp.close()


def test_benchmark_diagnostics(engine, oracle):
    """kyb_diag_mad_peak and kyb_diag_wave_stamps: plausible figures, and the stamped kernels still produce the oracle's bytes"""
    import torch
    pk = engine.mad_peak(20.0)
    cus = engine.device_info()["compute_units"]
    assert pk["kernel_ms"] >= 15.0 and 1.2 < pk["clock_ghz"] < 2.7, pk      # (the SGPR-carry pass, usually the one reported, runs a little shorter than the calibrated VCC pass)
    assert 3.9 < pk["simd_cycles_per_mad"] < 6.0, pk                         # quarter rate: never below 4 cycles per wavefront instruction
    nominal = cus * 4 * 64 / 4 * 2.4e9
    assert 0.5 * nominal < pk["mads_per_s"] <= 1.001 * nominal, pk
    dev = torch.device("cuda", 0)
    n = 1 << 17
    s_np = synth.scalars(n, 91)
    sc = torch.from_numpy(s_np).to(dev)
    pts = torch.empty((n, 40), dtype=torch.int32, device=dev)
    out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    outb = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    engine.mul_base_dev(sc, out_ext=pts)
    engine.sync()
    buf = torch.zeros(8, dtype=torch.int64, device=dev)
    engine.wave_stamps(buf)
    try:
        engine.mul_dev(sc, pts_ext=pts, out_enc=out)
        engine.sync()
        a = [int(v) % (1 << 64) for v in buf.cpu().tolist()]
        assert a[4] == n // 64                                                # every wavefront of k_mul_ladder stamped once
        ghz = ((a[2] - a[0]) % (1 << 64)) / ((a[3] - a[1]) % (1 << 64)) * 0.1
        assert 1.2 < ghz < 2.7, ghz
        buf.zero_()
        torch.cuda.synchronize()
        engine.mul_base_dev(sc, out_enc=outb)
        engine.sync()
        b = [int(v) % (1 << 64) for v in buf.cpu().tolist()]
        assert b[4] > 0 and b[4] % 16 == 0                                    # k_mul_base64: whole 1024-thread workgroups
    finally:
        engine.wave_stamps(None)
    idx = np.arange(0, n, 257)
    assert np.array_equal(out.cpu().numpy()[idx], oracle.mul_batch(s_np[idx], pts.cpu().numpy()[idx], nthreads=8))
    assert np.array_equal(outb.cpu().numpy()[idx], oracle.mul_base_batch(s_np[idx], nthreads=8))
    buf.zero_()
    torch.cuda.synchronize()
    engine.mul_dev(sc, pts_ext=pts, out_enc=out)
    engine.sync()
    assert not buf.cpu().any()                                                # switched off: nothing is written any more
