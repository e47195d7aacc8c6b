"""Error paths and secret hygiene of the engine (round-4 review item 4: 13 KYB_E_NOMEM returns and ~10 wipe scopes with zero tests).

The trait the engine sits behind is infallible (group.rs:139): a binding can only abort on a failed call, so the engine must not be the reason —
and when a call does fail (device memory exhausted, a launch refused) the context must stay usable.  Test hooks of the ABI
(include/kyber_ed25519.h): `diag.fail_alloc_after = k` / `diag.fail_launch_after = k` make the k-th buffer allocation / kernel launch of a
context fail without being made.  For every scenario below — host-pointer and device-pointer entry points at sizes that reach every lazily
grown buffer — EVERY allocation and EVERY launch of the call is failed in turn on a fresh context: the call must return KYB_E_NOMEM / KYB_E_HIP
with a text that names the buffer / the launch, the NEXT call on the same context must give the bytes of an undisturbed context (checked
against the oracle), and the context must hold exactly the memory an undisturbed one holds afterwards (diag.dev_kib / diag.host_kib: nothing
leaked, nothing lost).  The allocation sites reached are compared with the list in the source.

Secret hygiene: after calls that take private keys and nonces (and kyb_mul_batch, whose result is a Diffie-Hellman shared secret), also after
FAILED ones, kyb_diag_scratch_read copies the context's page-locked and device staging buffers back: no secret scalar (and no shared point) of
the call is found in them."""
import os
import re

import numpy as np
import pytest

import kyber_rs_amd
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def image():
    e = kyber_rs_amd.Engine(0, crosscheck=True)
    return e.base_table()


def fresh(image):
    """a context of its own with nothing allocated yet (the table image imported, not rebuilt)"""
    e = kyber_rs_amd.Engine(0, build_table=False, private=True, crosscheck=True)
    e.base_table_import(image)
    return e


def _inputs(n, seed):
    s, k = synth.scalars(n, seed), synth.scalars(n, seed + 1, b"k")
    return s, k


class Scenario:
    def __init__(self, name, prepare, call, options=()):
        self.name, self.prepare, self.call, self.options = name, prepare, call, options


def _torch():
    import torch
    return torch


def _scenarios(orc):
    """(name, prepare(engine) -> state, call(engine, state) -> bytes of every output, options).  `prepare` runs on a context of its own (it may use
    the engine to make inputs); the returned state holds only host data or device tensors that do not belong to a context."""
    sc = []

    def host(name, n, fn, options=()):
        sc.append(Scenario(name, lambda e: None, lambda e, st: fn(e), options))

    s1, k1 = _inputs(70000, 11)
    pts_small = orc.mul_base_ext_batch(s1[:64])
    host("mul_base 1 item (zero-copy buffer, completion flag)", 1, lambda e: e.mul_base(s1[:1]).tobytes())
    host("mul_base 3,000 items with limbs", 3000, lambda e: b"".join(a.tobytes() for a in e.mul_base(s1[:3000], want_ext=True)))
    host("mul_base 20,000 items (device staging, split finish)", 20000, lambda e: e.mul_base(s1[:20000]).tobytes())
    host("mul_base 70,000 items from pageable memory (pipelined: bounce ring, landing area)", 70000, lambda e: e.mul_base(s1).tobytes())
    host("mul 64 items, limbs in", 64, lambda e: e.mul(k1[:64], pts_ext=pts_small).tobytes())
    msgs = synth.messages(5000, 13)
    host("sign 64", 64, lambda e: e.schnorr_sign(s1[:64], k1[:64], msgs[:64]).tobytes())
    host("sign 5,000", 5000, lambda e: e.schnorr_sign(s1[:5000], k1[:5000], msgs).tobytes())
    host("eddsa_sign 300", 300, lambda e: e.eddsa_sign(s1[:300], msgs[:300]).tobytes())
    host("pripoly_eval 40 x 30", 0, lambda e: e.pripoly_eval(s1[:40 * 30].reshape(40, 30, 32), np.arange(17, dtype=np.uint32)).tobytes())

    def with_points(name, n, fn, options=()):
        def prepare(e):
            enc, ext = e.mul_base(s1[:n], want_ext=True)
            sig = e.schnorr_sign(s1[:n], k1[:n], msgs[:n]) if n <= len(msgs) else None
            return {"enc": enc, "ext": ext, "sig": sig}
        sc.append(Scenario(name, prepare, fn, options))

    with_points("mul 9,000 items, encodings in (mid-size ladder)", 9000, lambda e, st: e.mul(k1[:9000], pts_enc=st["enc"]).tobytes())
    with_points("mul 70,000 items, limbs in (pipelined)", 70000, lambda e, st: e.mul(k1, pts_ext=st["ext"]).tobytes())
    with_points("verify 64", 64, lambda e, st: e.verify(st["enc"], msgs[:64], st["sig"], 1).tobytes())
    with_points("verify 3,000", 3000, lambda e, st: e.verify(st["enc"], msgs[:3000], st["sig"], 0).tobytes())
    with_points("verify_points 3,000 (keys as points: encoding buffer)", 3000, lambda e, st: e.verify_points(st["ext"], msgs[:3000], st["sig"], 1).tobytes())
    with_points("encode / decode / add / equal / point_checks 500", 500,
                lambda e, st: e.encode(st["ext"]).tobytes() + e.decode(st["enc"])[0].tobytes() + e.add(st["ext"], st["ext"][::-1].copy()).tobytes()
                + e.equal(st["ext"], st["ext"][::-1].copy()).tobytes() + e.point_checks(enc=st["enc"]).tobytes())
    with_points("pubpoly_eval 43 commitments at 64 indices", 43, lambda e, st: e.pubpoly_eval(st["ext"], np.arange(64, dtype=np.uint32)).tobytes())
    with_points("pubpoly_eval_multi 8 x 43", 8 * 43, lambda e, st: e.pubpoly_eval_multi(st["ext"].reshape(8, 43, 40), np.arange(8, dtype=np.uint32)[:, None]).tobytes())
    with_points("dkg_verify_round_enc 16 x 11", 16 * 11, lambda e, st: b"".join(a.tobytes() for a in e.dkg_verify_round_enc(st["enc"].reshape(16, 11, 32), 5)))
    with_points("sum_points 43 x 16", 43 * 16, lambda e, st: e.sum_points(st["ext"].reshape(43, 16, 40)).tobytes())
    with_points("lincomb 6 x 9 (small: product staging)", 54, lambda e, st: e.lincomb(k1[:54].reshape(6, 9, 32), pts_ext=st["ext"].reshape(6, 9, 40)).tobytes())
    with_points("lincomb 3 x 900", 2700, lambda e, st: e.lincomb(k1[:2700].reshape(3, 900, 32), pts_ext=st["ext"].reshape(3, 900, 40)).tobytes())
    with_points("lincomb_public 100 outputs over 100 shared points (window tables)", 100,
                lambda e, st: e.lincomb(k1[:100 * 100].reshape(100, 100, 32), pts_ext=st["ext"], public=True).tobytes())
    # device-resident batches: the caller's tensors, the engine's scratch

    def dev(name, n, fn):
        def prepare(e):
            torch = _torch()
            enc, ext = e.mul_base(s1[:n], want_ext=True)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
            return {"s": t(s1[:n]), "k": t(k1[:n]), "enc": t(enc), "ext": t(ext), "out": torch.zeros((n, 32), dtype=torch.uint8, device="cuda"),
                    "oext": torch.zeros((n, 40), dtype=torch.int32, device="cuda")}

        def call(e, st):
            st["out"].zero_(); st["oext"].zero_()
            fn(e, st)
            e.sync()
            return st["out"].cpu().numpy().tobytes() + st["oext"].cpu().numpy().tobytes()
        sc.append(Scenario(name, prepare, call))

    dev("mul_base_dev 50,000", 50000, lambda e, st: e.mul_base_dev(st["s"], out_enc=st["out"], out_ext=st["oext"]))
    dev("mul_dev 30,000, encodings in and out", 30000, lambda e, st: e.mul_dev(st["k"], pts_enc=st["enc"], out_enc=st["out"]))
    dev("mul_dev 2,000, limbs in and out", 2000, lambda e, st: e.mul_dev(st["k"], pts_ext=st["ext"], out_ext=st["oext"]))
    dev("encode_dev / decode_dev 40,000", 40000, lambda e, st: (e.encode_dev(st["ext"], st["out"]), e.decode_dev(st["enc"], st["oext"])))
    return sc


def _attempt(e, sc, st):
    try:
        return sc.call(e, st), None
    except kyber_rs_amd.KyberHipError as err:
        return None, str(err)


def _source_sites():
    src = open(os.path.join(ROOT, "kyber-rs_amd", "csrc", "engine.hip")).read()
    sites = set(re.findall(r'fail\(KYB_E_NOMEM, "([^"]+)", e\)', src))
    return sites - {"table workspace allocation"}          # the windowed-table kernel's workspace: reachable in the cross-check build only


def test_every_allocation_and_every_launch_failed_in_turn(image, oracle):
    scenarios = _scenarios(oracle)
    prep_engine = fresh(image)
    seen_sites, n_alloc_faults, n_launch_faults = set(), 0, 0
    try:
        for sc in scenarios:
            st = sc.prepare(prep_engine)
            control = fresh(image)
            for key, val in sc.options:
                control.set_option(key, val)
            want, err = _attempt(control, sc, st)
            assert err is None, (sc.name, err)
            assert want == _attempt(control, sc, st)[0], sc.name                       # deterministic: the second call gives the same bytes
            held = (control.get_option("diag.dev_kib"), control.get_option("diag.host_kib"))
            control.close()
            for option, code, budget in (("diag.fail_alloc_after", "KYB_E_NOMEM", 24), ("diag.fail_launch_after", "KYB_E_HIP", 24)):
                faults = 0
                for k in range(1, budget + 1):
                    e = fresh(image)
                    try:
                        for key, val in sc.options:
                            e.set_option(key, val)
                        e.set_option(option, k)
                        got, err = _attempt(e, sc, st)
                        left = e.get_option(option)
                        e.set_option(option, 0)
                        if err is None:
                            assert got == want and left > 0, (sc.name, option, k)       # the call has fewer than k allocations / launches: done
                            break
                        faults += 1
                        assert left == 0 and code in err, (sc.name, option, k, err)
                        if code == "KYB_E_NOMEM":
                            site = re.search(r"KYB_E_NOMEM[^:]*: ([^:]+)", err)
                            assert site, err
                            seen_sites.add(site.group(1).strip())
                            assert "out of memory" in err.lower() or "memory" in err.lower(), err
                        else:
                            assert "injected launch failure" in err and "launch::" in err, err
                        # the context is still usable, its answer is the undisturbed one, and it holds what an undisturbed context holds
                        again, err2 = _attempt(e, sc, st)
                        assert err2 is None and again == want, (sc.name, option, k, err2)
                        assert (e.get_option("diag.dev_kib"), e.get_option("diag.host_kib")) == held, (sc.name, option, k, held)
                    finally:
                        e.close()
                else:
                    raise AssertionError(f"{sc.name}: more than {budget} {option} steps in one call")
                assert faults >= 1 or code == "KYB_E_NOMEM", (sc.name, option)           # (a small device-pointer call allocates nothing; every call launches)
                if code == "KYB_E_NOMEM":
                    n_alloc_faults += faults
                else:
                    n_launch_faults += faults
            print(f"{sc.name}: ok")
    finally:
        prep_engine.close()
    # kyb_host_alloc: NULL with KYB_E_NOMEM in kyb_last_error, the next one works
    e = fresh(image)
    try:
        e.set_option("diag.fail_alloc_after", 1)
        with pytest.raises(Exception):
            e.pinned_array((4096,), np.uint8)
        a = e.pinned_array((4096,), np.uint8)
        a[:] = 7
        assert int(a.sum()) == 7 * 4096
    finally:
        e.close()
    want_sites = _source_sites()
    print(f"{n_alloc_faults} allocation faults, {n_launch_faults} launch faults injected over {len(scenarios)} scenarios; sites reached: {sorted(seen_sites)}")
    assert want_sites and want_sites <= seen_sites, f"allocation sites never failed: {sorted(want_sites - seen_sites)}"
    assert n_alloc_faults >= len(want_sites) and n_launch_faults >= 2 * len(scenarios)


def _find_any(haystack: bytes, needles) -> list:
    return [i for i, nd in enumerate(needles) if nd in haystack]


def test_secret_operands_do_not_stay_behind_in_the_contexts_buffers(image, oracle):
    """private keys and nonces (unique 32-byte strings) after kyb_mul_base_batch / kyb_mul_batch / signing calls at the three host-pointer regimes
    (zero-copy, staged, pipelined) — and after a FAILED call of each: the page-locked buffers, the device staging and (for kyb_mul_batch) the
    products in them are searched for every secret of the call"""
    s, k = _inputs(70000, 21)
    msgs = synth.messages(5000, 23)
    prep = fresh(image)
    enc_all, ext_all = prep.mul_base(synth.scalars(70000, 29), want_ext=True)
    prep.close()
    # page-locked zero-copy / bounce buffers, device staging: cleared per call (include/kyber_ed25519.h "secrets"); 7 = the records of the
    # four-workgroup one-item product (secret multiples of the point), which the kernel itself clears behind the last piece
    cleared = (0, 1, 2, 7)
    cases = [
        ("mul_base", (1, 2000, 20000, 70000), lambda e, n: e.mul_base(s[:n]), lambda n, out: [s[i].tobytes() for i in range(0, n, max(1, n // 64))]),
        ("mul", (1, 2000, 20000, 70000), lambda e, n: e.mul(k[:n], pts_ext=ext_all[:n]),
         lambda n, out: [k[i].tobytes() for i in range(0, n, max(1, n // 64))] + [out[i].tobytes() for i in range(0, n, max(1, n // 64))]),      # s * P is a shared secret
        ("sign", (1, 500, 5000), lambda e, n: e.schnorr_sign(s[:n], k[:n], msgs[:n]),
         lambda n, out: [s[i].tobytes() for i in range(0, n, max(1, n // 64))] + [k[i].tobytes() for i in range(0, n, max(1, n // 64))]),
        ("eddsa_sign", (1, 500), lambda e, n: e.eddsa_sign(s[:n], msgs[:n]), lambda n, out: [s[i].tobytes() for i in range(0, n, max(1, n // 64))]),
    ]
    checked = 0
    for name, sizes, call, secrets in cases:
        for n in sizes:
            for fail_launch in (0, 1, 2):
                e = fresh(image)
                try:
                    if fail_launch:
                        e.set_option("diag.fail_launch_after", fail_launch)
                    try:
                        out = call(e, n)
                        failed = False
                    except kyber_rs_amd.KyberHipError:
                        out, failed = None, True
                    e.set_option("diag.fail_launch_after", 0)
                    if fail_launch and not failed:
                        continue                                   # the call has fewer launches than that
                    needles = secrets(n, out if out is not None else np.zeros((n, 64), np.uint8))
                    if out is None:
                        needles = [nd for nd in needles if any(nd)]
                    for which in cleared:
                        buf = e.scratch_read(which)
                        hits = _find_any(buf, needles)
                        assert not hits, f"{name} n={n} (launch fault {fail_launch}): secret #{hits[0]} of the call is still in buffer {which} ({len(buf)} bytes)"
                        checked += 1
                finally:
                    e.close()
    # the hook itself sees what a call leaves where it is NOT cleared (public data: the encodings of a verification), so a clean answer above means something
    e = fresh(image)
    try:
        pub = e.mul_base(s[:300])
        sig = e.schnorr_sign(s[:300], k[:300], msgs[:300])
        e.verify(pub, msgs[:300], sig, 1)
        assert pub[5].tobytes() in e.scratch_read(0) + e.scratch_read(2)
    finally:
        e.close()
    assert checked >= 40
