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


def test_set_option_and_profiling_race_with_launches(engine, oracle):
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


def test_encode_batched_matches_per_item_encode(engine, oracle):
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
