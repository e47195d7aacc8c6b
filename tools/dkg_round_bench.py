"""One node's curve work in a Pedersen DKG round, through the host-pointer C ABI, against the
single-thread CPU port: what a kyber-rs node would gain by routing its loops to the batch entry points.

Per node with n participants and threshold t (call sites: dkg.rs / vss/pedersen/vss.rs / poly.rs):
  dealer    commit                 t  x mul(coeff, Some(B))                     vss.rs:303
            shares                 n  x PriPoly::eval (t scalar multiply-adds)  poly.rs:133-152
            deals                  n  x (schnorr::sign + dh_exchange)           vss.rs:361-386
  verifier  process n deals        n  x (schnorr::verify + dh_exchange)         vss.rs:640-660
            verify_deal            n  x (mul(f_i, None) + PubPoly::eval(i))     vss.rs:904-909
  finish    distributed public polynomial: sum of n commitment polynomials      dkg.rs:905-953
Outputs are spot-checked against the oracle."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

# kernel variants are selected here: the cross-check build (csrc/Makefile CROSSCHECK=1) — the product library has no such options
os.environ.setdefault("KYB_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber-rs_amd", "libkyber_ed25519_hip_crosscheck.so"))
import kyber_rs_amd
import oracle_lib
import synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, nargs="+", default=[16, 64, 256, 1024])
args = ap.parse_args()
eng = kyber_rs_amd.Engine(0)
orc = oracle_lib.Oracle()
L = synth.L


def timed(fn):
    fn()
    best = None
    for _ in range(3):                       # the best of three: one sample now and then catches a garbage collection of the big operand arrays
        t0 = time.perf_counter(); out = fn(); dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None or dt < best else best
    return out, best


# single-thread CPU port: per-operation times on this host
s256 = synth.scalars(256, 1); p256 = orc.mul_base_ext_batch(s256); m256 = synth.messages(256, 1)
sig256 = orc.schnorr_sign_batch(s256, synth.scalars(256, 2), m256)
pub256 = np.stack([np.frombuffer(orc.encode(p), dtype=np.uint8) for p in p256])
cpu = {}
for name, fn in (("mul", lambda: orc.mul_batch(s256, p256)), ("mul_base", lambda: orc.mul_base_batch(s256)),
                 ("sign", lambda: orc.schnorr_sign_batch(s256, s256, m256)), ("verify", lambda: orc.verify_batch(1, pub256, m256, sig256))):
    t0 = time.perf_counter(); fn(); cpu[name] = (time.perf_counter() - t0) / 256 * 1e3
t0 = time.perf_counter()
for i in range(8):
    orc.pubpoly_eval(p256[:32], i)
cpu["eval_per_coeff"] = (time.perf_counter() - t0) / 8 / 32 * 1e3
t0 = time.perf_counter()
for e in pub256:
    orc.decode(bytes(e))
cpu["unmarshal"] = (time.perf_counter() - t0) / 256 * 1e3                     # incl. the ctypes call (~2 us)
t0 = time.perf_counter()
for i in range(8):
    orc.pripoly_eval(s256, i)
cpu["share_per_coeff"] = (time.perf_counter() - t0) / 8 / 256 * 1e3
print("CPU port, 1 thread, ms per op:", {k: round(v, 4) for k, v in cpu.items()})
print("n, t, gpu_ms_total, cpu_ms_estimate, speedup, breakdown_ms   [wire: the n*t commitments arrive as 32-byte encodings; GPU total with "
      "kyb_pubpoly_eval_multi_enc_batch / kyb_sum_enc_batch, CPU estimate with the n*t unmarshal_binary calls the reference makes]")
for n in args.n:
    t = n * 2 // 3 + 1
    coeffs = synth.scalars(t, 100 + n)
    longterm = synth.scalars(n, 200 + n)
    pubs = eng.mul_base(longterm)                                            # everybody's long-term public keys (given)
    base = np.tile(orc.base(), (t, 1))
    br = {}
    (commit_enc, commit_ext), br["commit"] = timed(lambda: eng.mul(coeffs, pts_ext=base, want_ext=True))
    ci = [int.from_bytes(bytes(c), "little") for c in coeffs]
    shares, br["shares"] = timed(lambda: eng.pripoly_eval(coeffs, np.arange(n, dtype=np.uint32)))      # PriPoly::shares: n Horner chains of t scalar multiply-adds
    for i in (0, n // 2, n - 1):
        assert int.from_bytes(bytes(shares[i]), "little") == sum(c * pow(i + 1, j, L) for j, c in enumerate(ci)) % L
    msgs = synth.messages(n, n)
    nonces = synth.scalars(n, 300 + n)
    me = np.tile(longterm[0], (n, 1))
    mypub = np.tile(pubs[0], (n, 1))
    sigs, br["sign_deals"] = timed(lambda: eng.schnorr_sign(me, nonces, msgs, pubs=mypub))
    dh, br["dh_out"] = timed(lambda: eng.mul(me, pts_enc=pubs))
    # verifier side: n incoming deals (here: the same dealer n times, which costs the same as n dealers)
    st, br["verify_deals"] = timed(lambda: eng.verify(mypub, msgs, sigs, 1))
    assert not st.any()
    dh2, br["dh_in"] = timed(lambda: eng.mul(me, pts_enc=pubs))
    polys = np.tile(commit_ext[None, :, :], (n, 1, 1))
    me_i = n // 2                                                            # this node's share index (x = me_i + 1: a log2(n)-bit multiplier per Horner step)
    idx = np.full((n, 1), me_i, dtype=np.uint32)
    fig, br["fig"] = timed(lambda: eng.mul_base(np.tile(shares[me_i], (n, 1))))
    ev, br["eval"] = timed(lambda: eng.pubpoly_eval_multi(polys, idx))
    assert np.array_equal(ev[:, 0], fig)
    coop_was = eng.get_option("coop.max_items")
    eng.set_option("coop.max_items", 0)                                       # the one-evaluation-per-lane kernel, for comparison (not part of the total)
    eng.set_option("poly.batch_segments", 1)
    ev_b, eval_batch_ms = timed(lambda: eng.pubpoly_eval_multi(polys, idx))
    eng.set_option("coop.max_items", coop_was)
    ev_c, eval_coop_ms = timed(lambda: eng.pubpoly_eval_multi(polys, idx))   # one evaluation per wavefront (what every n took before the segment-per-lane shape)
    eng.set_option("poly.batch_segments", 0)
    assert np.array_equal(ev_c, ev)
    assert np.array_equal(ev_b, ev)
    by_coeff = np.ascontiguousarray(polys.transpose(1, 0, 2))              # (the node accumulates the incoming polynomials coefficient by coefficient)
    dist, br["dist_poly"] = timed(lambda: eng.sum_points(by_coeff))
    polys_enc = np.tile(commit_enc[None, :, :], (n, 1, 1))                  # as received: dealer-major, 32 bytes per commitment
    (ev_w, ok_w), eval_wire_ms = timed(lambda: eng.pubpoly_eval_multi_enc(polys_enc, idx))
    assert np.array_equal(ev_w, ev) and ok_w.all()
    (dist_w, ok_w), dist_wire_ms = timed(lambda: eng.sum_points_enc(polys_enc, item_major=True))
    assert np.array_equal(dist_w, dist) and ok_w.all()
    (ev_1, dist_1, ok_1), round_wire_ms = timed(lambda: eng.dkg_verify_round_enc(polys_enc, me_i))      # both from ONE transfer and ONE decode
    assert np.array_equal(ev_1, ev[:, 0]) and np.array_equal(dist_1, dist) and ok_1.all()
    assert bytes(commit_enc[1]) == orc.mul(bytes(coeffs[1]), orc.base()) and bytes(dh[3]) == orc.mul(bytes(longterm[0]), orc.decode(bytes(pubs[3]))[0])
    gpu_ms = sum(br.values())
    cpu_ms = t * cpu["mul"] + n * (cpu["sign"] + cpu["mul"]) + n * (cpu["verify"] + cpu["mul"]) + n * (cpu["mul_base"] + t * cpu["eval_per_coeff"]) + n * t * 0.0005 + n * t * cpu["share_per_coeff"]
    print(f"{n}, {t}, {gpu_ms:.2f}, {cpu_ms:.0f}, {cpu_ms / gpu_ms:.0f}x, " + " ".join(f"{k}={v:.2f}" for k, v in br.items()) + f" (eval one per lane: {eval_batch_ms:.2f}, one per wavefront: {eval_coop_ms:.2f})"
          + f"   [wire: {gpu_ms - br['eval'] - br['dist_poly'] + eval_wire_ms + dist_wire_ms:.2f} ms, eval={eval_wire_ms:.2f} dist_poly={dist_wire_ms:.2f}; "
          + f"one call (kyb_dkg_verify_round_enc): {gpu_ms - br['eval'] - br['dist_poly'] + round_wire_ms:.2f} ms, eval+dist_poly={round_wire_ms:.2f}; CPU {cpu_ms + n * t * cpu['unmarshal']:.0f} ms]", flush=True)
