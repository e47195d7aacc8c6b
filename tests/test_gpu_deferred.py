"""Deferred points (kyb_defer_*, csrc/defer.inc): element-at-a-time callers hand their operations over without asking for results, a flush
evaluates the recorded graph in batches.  Every test records the calls the reference's own loops make, asks for bytes the way the
reference does (marshal_binary / eq), and compares with the oracle; the statistics of the arena show that the batching really happened."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _le(x: int) -> bytes:
    return int(x).to_bytes(32, "little")


@pytest.fixture()
def eng():
    """an engine context of its own: the arena's statistics then belong to this test"""
    import kyber_rs_amd
    e = kyber_rs_amd.Engine(0, private=True)
    yield e
    e.close()


def _delta(eng, before):
    after = eng.defer_stats()
    return {k: after[k] - before[k] for k in after}


def test_commit_is_one_batch_and_its_marshals_are_cache_hits(eng, oracle):
    """PriPoly::commit (poly.rs:195-206): t times mul(coeff_j, Some(base)) — recorded one by one, evaluated by ONE call when the first commitment
    is marshalled (session_id hashes all of them, vss.rs:303-307), the other marshals cost nothing"""
    t = 43
    coeffs = synth.scalars(t, 41)
    base = eng.defer_base()
    before = eng.defer_stats()
    commits = [eng.defer_mul(coeffs[j].tobytes(), base) for j in range(t)]
    assert _delta(eng, before)["engine_calls"] == 0                      # nothing has run
    encs = [eng.defer_get(c) for c in commits]
    d = _delta(eng, before)
    assert d["flushes"] == 1 and d["engine_calls"] <= 2 and d["marshal_cache_hits"] == t      # (base point once per arena) + one batched multiplication
    want = oracle.mul_base_batch(coeffs)
    assert encs == [bytes(w) for w in want]
    # the same through the fixed-base node, and limbs on request
    again = [eng.defer_mul_base(coeffs[j].tobytes()) for j in range(t)]
    eng.defer_flush()
    assert [oracle.encode(eng.defer_get_ext(h)) for h in again] == encs


def test_pubpoly_eval_chain_is_one_fused_call(eng, oracle):
    """PubPoly::eval (poly.rs:457-469): v = null; for j = t-1 .. 0: v = mul(xi, Some(v)); v = add(v, commits[j]) — 2 t dependent calls in the
    reference, one kyb_pubpoly_eval_multi_batch call here; commitments with small-order components included; several verifiers' chains of the
    same length share the call; defer.fuse = 0 gives the same bytes level by level"""
    t = 43
    commits_ext = oracle.mul_base_ext_batch(synth.scalars(t, 42))
    weak = oracle.decode(bytes.fromhex("c7176a703d4dd84fba3c0b760d10670f2a2053fa2c39ccc64ec7fd7792ac037a"))[0]      # a point of order 8
    commits_ext[3] = oracle.add(commits_ext[3], weak)
    cs = [eng.defer_input(c) for c in commits_ext]

    def record_eval(i):
        v = eng.defer_null()
        xi = _le(1 + i)
        for j in reversed(range(t)):
            v = eng.defer_mul(xi, v)
            v = eng.defer_add(v, cs[j])
        return v

    before = eng.defer_stats()
    v = record_eval(6)
    got = eng.defer_get(v)
    d = _delta(eng, before)
    assert got == oracle.pubpoly_eval(commits_ext, 6)
    assert d["horner_fused"] == 1 and d["engine_calls"] == 1 and d["flushes"] == 1
    # n verifiers' evaluations recorded before anybody looks: still one call
    before = eng.defer_stats()
    vs = [record_eval(i) for i in (0, 1, 63, 511, 65535, 2**32 - 2)]
    eng.defer_flush()
    d = _delta(eng, before)
    assert d["horner_fused"] == 6 and d["engine_calls"] == 1
    assert [eng.defer_get(h) for h in vs] == [oracle.pubpoly_eval(commits_ext, i) for i in (0, 1, 63, 511, 65535, 2**32 - 2)]
    # an inner node of a fused chain is still a point of its own: asking for it evaluates it (its own, shorter chain)
    v = eng.defer_null()
    inner = None
    for j in reversed(range(t)):
        v = eng.defer_mul(_le(5), v)
        v = eng.defer_add(v, cs[j])
        if j == 10:
            inner = v
    assert eng.defer_get(v) == oracle.pubpoly_eval(commits_ext, 4)
    assert eng.defer_get(inner) == oracle.pubpoly_eval(commits_ext[10:], 4)
    # a multiplier that is no share index (here: a full-size scalar) is not fused and still right
    big = synth.scalars(1, 43)[0].tobytes()
    v = eng.defer_null()
    for j in reversed(range(4)):
        v = eng.defer_mul(big, v)
        v = eng.defer_add(v, cs[j])
    before = eng.defer_stats()
    got = eng.defer_get(v)
    assert _delta(eng, before)["horner_fused"] == 0
    acc = oracle.null()
    for j in reversed(range(4)):
        acc = oracle.add(oracle.mul_ext(big, acc), commits_ext[j])
    assert got == oracle.encode(acc)
    # fusion off: level by level, same bytes
    eng.set_option("defer.fuse", 0)
    try:
        before = eng.defer_stats()
        v = record_eval(6)
        assert eng.defer_get(v) == oracle.pubpoly_eval(commits_ext, 6)
        d = _delta(eng, before)
        assert d["horner_fused"] == 0 and d["engine_calls"] >= 2 * t - 1
    finally:
        eng.set_option("defer.fuse", 1)


def test_recover_commit_chain_is_one_batch_and_one_sum(eng, oracle):
    """recover_commit (poly.rs:566-603): acc = null; for every share: tmp = mul(num / den, Some(share)); acc = add(acc, tmp) — the t
    multiplications in one call, the additions in one kyb_sum_batch call"""
    t = 20
    lam = synth.scalars(t, 44)
    shares = oracle.mul_base_ext_batch(synth.scalars(t, 45))
    hs = [eng.defer_input(s) for s in shares]
    before = eng.defer_stats()
    acc = eng.defer_null()
    for i in range(t):
        tmp = eng.defer_mul(lam[i].tobytes(), hs[i])
        acc = eng.defer_add(acc, tmp)
    got = eng.defer_get(acc)
    d = _delta(eng, before)
    assert got == oracle.lincomb(lam, shares)
    assert d["sums_fused"] == 1 and d["engine_calls"] == 2


def test_random_graphs_match_the_eager_calls(eng, oracle):
    """random expression graphs over every operation, evaluated through random get / equal / flush requests, against the oracle's eager evaluation;
    a small arena makes old handles stale, and a stale handle is an error, never a wrong answer"""
    import kyber_rs_amd
    rng = np.random.default_rng(46)
    pts = oracle.mul_base_ext_batch(synth.scalars(6, 47))
    pts = np.concatenate([pts, oracle.decode(bytes.fromhex("26e8958fc2b227b045c3f489f2ef98f0d5dfac05d3c63339b13802886d53fc05"))[0][None]])      # order 8
    for trial in range(6):
        nodes = []          # (handle, oracle limbs)
        for p in pts:
            nodes.append((eng.defer_input(p), p))
        nodes.append((eng.defer_null(), oracle.null()))
        nodes.append((eng.defer_base(), oracle.base()))
        for step in range(120):
            op = rng.integers(0, 6)
            a = nodes[rng.integers(0, len(nodes))]
            b = nodes[rng.integers(0, len(nodes))]
            sc = [synth.scalars(1, 1000 * trial + step)[0].tobytes(), _le(int(rng.integers(0, 9))), synth.raw256(1, 1000 * trial + step)[0].tobytes()][rng.integers(0, 3)]
            if op == 0:
                nodes.append((eng.defer_mul_base(sc), oracle.mul_base_ext(sc)))
            elif op == 1:
                nodes.append((eng.defer_mul(sc, a[0]), oracle.mul_ext(sc, a[1])))
            elif op == 2:
                nodes.append((eng.defer_add(a[0], b[0]), oracle.add(a[1], b[1])))
            elif op == 3:
                nodes.append((eng.defer_add(a[0], b[0], subtract=True), oracle.add(a[1], b[1], sub=True)))
            elif op == 4:
                nodes.append((eng.defer_neg(a[0]), oracle.neg(a[1])))
            else:
                k = rng.integers(0, 3)
                if k == 0:
                    assert eng.defer_get(a[0]) == oracle.encode(a[1])
                elif k == 1:
                    assert eng.defer_equal(a[0], b[0]) == (oracle.encode(a[1]) == oracle.encode(b[1]))
                else:
                    eng.defer_flush()
        for h, want in nodes:
            enc, ext = eng.defer_get(h, want_ext=True)
            assert enc == oracle.encode(want) and oracle.encode(ext) == enc
    # a small window: the value of an evaluated node outlives it (the table of kept values) — handles stay good, as answers and as operands;
    # with the table switched off a handle older than the window is refused, never answered wrongly; the host's floor drops values and all
    eng.set_option("defer.max_nodes", 64)
    try:
        first = eng.defer_mul_base(_le(5))
        for i in range(200):
            eng.defer_mul_base(_le(i))
        st = eng.defer_stats()
        assert st["nodes_dropped"] > 0 and st["values_kept"] > 0 and st["nodes_held"] <= 64
        assert eng.defer_get(first) == oracle.mul_base(_le(5))                                        # answered from the table
        again = eng.defer_mul(_le(3), first)                                                          # ... and taken back in as an operand
        assert eng.defer_get(again) == oracle.mul_base(_le(15))
        assert eng.defer_equal(first, eng.defer_mul_base(_le(5))) and not eng.defer_equal(first, again)
        st2 = eng.defer_stats()
        assert st2["kept_hits"] > st["kept_hits"] and st2["operands_readmitted"] > st["operands_readmitted"]
        eng.set_option("defer.keep_mib", 0)
        lost = eng.defer_mul_base(_le(9))
        for i in range(200):
            eng.defer_mul_base(_le(i))
        with pytest.raises(kyber_rs_amd.KyberHipError, match="stale"):
            eng.defer_get(lost)
        eng.set_option("defer.keep_mib", 256)
        m = eng.defer_mark()
        keep = eng.defer_mul_base(_le(7))
        eng.defer_floor(m)
        st3 = eng.defer_stats()
        assert st3["nodes_held"] == 1 and st3["values_kept"] == 0 and eng.defer_get(keep) == oracle.mul_base(_le(7))
        with pytest.raises(kyber_rs_amd.KyberHipError, match="stale"):
            eng.defer_get(first)                                                                      # the floor is the host's word that nothing older is wanted
    finally:
        eng.set_option("defer.max_nodes", 1 << 18)
        eng.set_option("defer.keep_mib", 256)


def test_random_graphs_through_a_small_window_on_the_engine(eng, oracle):
    """the random graphs again, on the real kernels, through a window of 64 nodes (it moves by 16 after a flush of everything pending): values
    kept behind the window — also those a projective flush left with Z != 1 and no bytes —, operands taken back in as leaves, Horner steps cut by
    the window's edge.  A handle may be refused (leaves and never-evaluated inner steps leave nothing behind) — then it is dropped from the pool —
    but whatever is answered equals the oracle's eager evaluation, and no flush fails."""
    import kyber_rs_amd
    rng = np.random.default_rng(4646)
    pts = oracle.mul_base_ext_batch(synth.scalars(6, 4747))
    eng.set_option("defer.max_nodes", 64)
    answers = refused = 0
    try:
        for trial in range(8):
            eng.set_option("defer.fuse", 0 if trial % 4 == 3 else 1)
            nodes = []

            def leaves():
                for p in pts:
                    nodes.append((eng.defer_input(p), p))
                nodes.append((eng.defer_null(), oracle.null()))
                nodes.append((eng.defer_base(), oracle.base()))

            def attempt(fn, used):
                """fn() -> result, or None when an operand was refused as stale (it is then forgotten)"""
                nonlocal refused
                try:
                    return fn()
                except kyber_rs_amd.KyberHipError as e:
                    assert "stale" in str(e), e
                    refused += 1
                    for h in used:
                        try:
                            eng.defer_get(h)
                        except kyber_rs_amd.KyberHipError:
                            nodes[:] = [nd for nd in nodes if nd[0] != h]
                    return None

            leaves()
            for step in range(260):
                if len(nodes) < 4:
                    leaves()
                op = rng.integers(0, 7)
                a = nodes[rng.integers(0, len(nodes))]
                b = nodes[rng.integers(0, len(nodes))]
                sc = [synth.scalars(1, 5000 * trial + step)[0].tobytes(), _le(int(rng.integers(1, 9))), synth.raw256(1, 5000 * trial + step)[0].tobytes()][rng.integers(0, 3)]
                if op == 0:
                    h = attempt(lambda: eng.defer_mul_base(sc), ())
                    if h is not None:
                        nodes.append((h, oracle.mul_base_ext(sc)))
                elif op == 1:
                    h = attempt(lambda: eng.defer_mul(sc, a[0]), (a[0],))
                    if h is not None:
                        nodes.append((h, oracle.mul_ext(sc, a[1])))
                elif op in (2, 3):
                    h = attempt(lambda: eng.defer_add(a[0], b[0], subtract=(op == 3)), (a[0], b[0]))
                    if h is not None:
                        nodes.append((h, oracle.add(a[1], b[1], sub=(op == 3))))
                elif op == 4:                      # a short Horner chain with a share index on top of a
                    x = _le(int(rng.integers(1, 9)))
                    cur = a
                    for _ in range(int(rng.integers(2, 6))):
                        c = nodes[rng.integers(0, len(nodes))]
                        m = attempt(lambda: eng.defer_mul(x, cur[0]), (cur[0],))
                        if m is None:
                            cur = None
                            break
                        mv = oracle.mul_ext(x, cur[1])
                        sm = attempt(lambda: eng.defer_add(m, c[0]), (m, c[0]))
                        if sm is None:
                            cur = None
                            break
                        cur = (sm, oracle.add(mv, c[1]))
                        if rng.integers(0, 4) == 0:
                            nodes.append(cur)          # an inner step is kept and used later
                    if cur is not None:
                        nodes.append(cur)
                else:
                    k = rng.integers(0, 3)
                    if k == 0:
                        got = attempt(lambda: eng.defer_get(a[0]), (a[0],))
                        if got is not None:
                            answers += 1
                            assert got == oracle.encode(a[1])
                    elif k == 1:
                        got = attempt(lambda: eng.defer_equal(a[0], b[0]), (a[0], b[0]))
                        if got is not None:
                            answers += 1
                            assert got == (oracle.encode(a[1]) == oracle.encode(b[1]))
                    else:
                        eng.defer_flush()              # never fails: nothing pending can have lost an operand
            for h, want in list(nodes):
                got = attempt(lambda: eng.defer_get(h, want_ext=True), (h,))
                if got is not None:
                    answers += 1
                    assert got[0] == oracle.encode(want) and oracle.encode(got[1]) == got[0]
        st = eng.defer_stats()
        print(f"small window on the engine: {answers} answers checked, {refused} calls refused as stale; {st}")
        assert answers > 600 and refused * 3 < answers and st["kept_hits"] > 0 and st["operands_readmitted"] > 0
    finally:
        eng.set_option("defer.max_nodes", 1 << 18)
        eng.set_option("defer.fuse", 1)
        eng.defer_floor(eng.defer_mark())


def test_threads_recording_into_one_arena(eng, oracle):
    """four host threads record and ask on ONE context at the same time (the arena has its own lock; a flush evaluates everybody's nodes): every
    thread gets its own results right"""
    import threading
    errors = []

    def work(i):
        try:
            sc = synth.scalars(24, 300 + i)
            want = oracle.mul_base_batch(sc)
            for rnd in range(6):
                hs = [eng.defer_mul_base(sc[j].tobytes()) for j in range(24)]
                acc = hs[0]
                for h in hs[1:]:
                    acc = eng.defer_add(acc, h)
                if [eng.defer_get(h) for h in hs] != [bytes(w) for w in want]:
                    errors.append((i, rnd, "products"))
                ext = [oracle.mul_base_ext(sc[j].tobytes()) for j in range(24)] if rnd == 0 else None
                if ext is not None:
                    s = ext[0]
                    for e in ext[1:]:
                        s = oracle.add(s, e)
                    if eng.defer_get(acc) != oracle.encode(s):
                        errors.append((i, rnd, "sum"))
        except Exception as ex:  # noqa: BLE001
            errors.append((i, repr(ex)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:3]


def test_points_travel_between_contexts(oracle):
    """Point is Copy and Send: a point recorded through one context (thread) may be asked for, or used as an operand, through another.  Handles name
    their arena, so that works — also after the first context is gone (a worker thread that ended before its results were looked at)"""
    import kyber_rs_amd
    e1 = kyber_rs_amd.Engine(0, private=True)
    e2 = kyber_rs_amd.Engine(0, private=True)
    try:
        a = e1.defer_mul_base(_le(4242))
        mine = e2.defer_mul_base(_le(4242))
        assert (a >> 40) != (mine >> 40)                                  # never the same number for two points of two arenas
        b = e2.defer_add(e2.defer_mul(_le(77), a), a)
        pa = oracle.mul_base_ext(_le(4242))
        assert e2.defer_get(b) == oracle.encode(oracle.add(oracle.mul_ext(_le(77), pa), pa))
        assert e2.defer_get(a) == oracle.mul_base(_le(4242)) == e1.defer_get(a)
        assert e2.defer_equal(a, mine) and not e2.defer_equal(a, b)
        late = e1.defer_mul_base(_le(999))                                # recorded, never asked for ...
    finally:
        e1.close()                                                        # ... and its context ends
    try:
        assert e2.defer_get(late) == oracle.mul_base(_le(999))
    finally:
        e2.close()
