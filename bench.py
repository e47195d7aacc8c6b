#!/usr/bin/env python3
"""Headline benchmark: Ed25519 scalar-mults/sec on MI355X (BASELINE.json `metric`).

  python bench.py --gpus N --steps K --warmup W [--workload mul|mul_base|sign|verify] [--n ITEMS_PER_GPU]

A "step" = one pass of the hot path over one batch that is already resident in HBM:
  mul       2^20 variable-base mults, random scalars + random points   (BASELINE configs[1], default, the line's `value`)
  mul_base  2^20 fixed-base mults                                       (configs[2])
  sign      2^18 Schnorr signatures, 32-byte messages                   (configs[3])
  verify    2^20 Schnorr verifications with the reference's checks      
For N > 1 the driver launches one process per GPU (torch.distributed.run); every rank owns its own
shard of N x ITEMS_PER_GPU independent items (weak scaling, no data-path collective).  The only
collective is the one-time RCCL broadcast of the base-point table image built on rank 0.

Rank 0 prints ONE JSON line.  `value` is whole-job items/s of the PRIMARY workload over the timed K steps
(barrier + synchronize on both sides, max over ranks).  At N = 1 the other single-GPU configurations of
BASELINE.json are then timed the same way, each in its own timed region OUTSIDE the primary one, and
reported under `workloads` (value, ms_per_step, roofline, cpu_baseline, parity_checked_items each), and `small_calls` gives the
wall time of ONE synchronous host-pointer call with 1 / 64 items (what unmodified protocol code sees; a latency, not a throughput).

`roofline` prices the dominant kernel against the v_mad_u64_u32 issue peak: the path is integer-VALU bound
by construction (BASELINE.json north_star), not HBM or MFMA bound, so the object carries
`bound: "valu-int"` and additionally reports the (negligible) algorithmic HBM rate.  Two peaks are given:
`peak` = the rate MEASURED on this chip with every SIMD issuing nothing but v_mad_u64_u32
(tools/microbench/valu_rates.hip, clock settles at ~1.9 GHz under that load) and `peak_nominal` =
1024 SIMDs x 64 lanes / 4 cycles x 2.4 GHz (the data-sheet clock the chip does not hold under this load;
profiles/r02/ladder_clock.json has the in-kernel clock).  `frac` uses the algorithmic multiply-adds of the
REFERENCE's algorithm, `executed_frac` the multiply-adds this repository's kernels really execute.
`cpu_baseline` times the oracle — a C port of the reference algorithm, NOT the Rust binary — on this
box's host cores (rank 0, N = 1 only)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic 32x32->64 multiply(-add)s per unit, fixed by the reference's algorithm:
# whole step (mult + encode) and the dominant kernel alone (mult with projective output; the encode is
# k_finish's work when the batched finish is on)
# verify (eddsa_sig.rs:159-212): 2 decodes (2 x 16,000) + 2 has_small_order encodes (2 x 15,270) + fixed-base
# (46,980) + variable-base (188,640) + add (900) + eq = 2 encodes (2 x 15,270)
PRODUCTS = {"mul": 203_910, "mul_base": 62_250, "sign": 124_500, "verify": 329_600}
PRODUCTS_DOMINANT = {"k_mul": 188_640, "k_mul_ladder": 188_640, "k_mul_base": 46_980, "k_sign": 124_500}
# multiply-adds the kernels of THIS repository actually execute per item (the algorithms differ from the
# reference's: 256-step ladder; 43 radix-64 / 52 radix-32 / 64 radix-16 mixed additions) — reported next to the
# algorithmic figure so that `frac` (algorithmic, may exceed 1 where less work is done) is not mistaken
# for pipe utilisation
EXECUTED = {"k_mul": 188_640, "k_mul_ladder": 256 * (5 * 100 + 4 * 55 + 10) + 2_300, "k_mul_base": {64: 43 * 700, 32: 52 * 700, 16: 64 * 700}, "k_sign": 2 * 64 * 700 + 15_270}
ALG_BYTES = {"mul": 32 + 160 + 32, "mul_base": 32 + 32, "sign": 32 + 32 + 32 + 64, "verify": 32 + 64 + 32 + 1}
UNIT = {"mul": "variable-base scalar-mults/s", "mul_base": "fixed-base scalar-mults/s", "sign": "signatures/s", "verify": "verifications/s"}
METRIC = {"mul": "Ed25519 scalar-mults/sec", "mul_base": "Ed25519 scalar-mults/sec", "sign": "Ed25519 Schnorr signatures/sec", "verify": "Ed25519 Schnorr verifications/sec"}
DOMINANT = {"mul": "k_mul_ladder", "mul_base": "k_mul_base", "sign": "k_mul_base", "verify": "k_mul_ladder"}
WORKLOAD_TEXT = {"mul": "2^20 variable-base scalar-mults, random scalars+points, reference-limb points in, 32-byte encodings out",
                 "mul_base": "2^20 fixed-base (generator) scalar-mults, 32-byte encodings out",
                 "sign": "2^18 Schnorr signs, 32-byte messages, 64-byte signatures out",
                 "verify": "2^20 Schnorr verifications with the reference's checks, 32-byte messages, status bytes out"}
# measured on MI355X: 26.8e12 v_mad_u64_u32 lane-ops/s with every SIMD issuing (8 waves/SIMD, clock
# settles at ~1.9 GHz under this load) — tools/microbench/valu_rates.hip, profiles/r01_valu_rates_mi355x.jsonl
PEAK_MAD_PER_S = 26.8e12
# data-sheet figure: 256 CUs x 4 SIMDs x 64 lanes per 4 cycles (half rate of the SIMD-32 VALU) x 2.4 GHz
PEAK_MAD_NOMINAL = 1024 * 64 / 4 * 2.4e9
HBM_PEAK_GBS = 8000.0
DEFAULT_N = {"mul": 1 << 20, "mul_base": 1 << 20, "sign": 1 << 18, "verify": 1 << 20}
PROFILE_ROUND = "r02"


class Inputs:
    """synthetic inputs of one rank (seed = 1 + rank), generated once and shared by the workloads"""

    def __init__(self, seed, dev):
        self.seed, self.dev, self._np, self._t, self.msg_list = seed, dev, {}, {}, None

    def scalars_np(self, n, tag=b"scalar"):
        import synth
        key = (tag, n)
        if key not in self._np:
            have = [k for k in self._np if k[0] == tag and k[1] >= n]
            self._np[key] = self._np[have[0]][:n] if have else synth.scalars(n, self.seed, tag)
        return self._np[key]

    def scalars(self, n, tag=b"scalar"):
        import torch
        key = (tag, n)
        if key not in self._t:
            self._t[key] = torch.from_numpy(self.scalars_np(n, tag)).to(self.dev)
        return self._t[key]

    def messages(self, n):
        import synth
        if self.msg_list is None or len(self.msg_list) < n:
            self.msg_list = synth.messages(n, self.seed)
        return self.msg_list[:n]


def setup_workload(wl, n, eng, inp, dev, stream, keyed):
    """-> dict with the step closure and everything the parity check / CPU baseline need; inputs end up resident in HBM"""
    import numpy as np
    import torch
    w = {"wl": wl, "n": n, "keyed": keyed}
    sc = inp.scalars(n)
    w["sc_np"] = inp.scalars_np(n)
    w["out"] = out = torch.empty((n, 64 if wl == "sign" else (1 if wl == "verify" else 32)), dtype=torch.uint8, device=dev)
    if wl == "mul":
        psc = inp.scalars(n, b"point")
        w["pts"] = pts = torch.empty((n, 40), dtype=torch.int32, device=dev)        # point_i = (hash mod L) * B, reference limbs
        eng.mul_base_dev(psc, out_ext=pts, stream=stream)
        w["step"] = lambda: eng.mul_dev(sc, pts_ext=pts, out_enc=out, stream=stream)
    elif wl == "mul_base":
        w["step"] = lambda: eng.mul_base_dev(sc, out_enc=out, stream=stream)
    else:
        w["k"] = k = inp.scalars(n, b"k")
        w["msg_list"] = msg_list = inp.messages(n)
        msgs = torch.from_numpy(np.frombuffer(b"".join(msg_list), dtype=np.uint8).copy()).to(dev)
        off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int32, device=dev)
        pubs = None
        if wl == "sign":
            if keyed:
                pubs = torch.empty((n, 32), dtype=torch.uint8, device=dev)
                eng.mul_base_dev(sc, out_enc=pubs, stream=stream)
            w["step"] = lambda: eng.sign_dev(sc, k, msgs, off, out, stream=stream, pubs=pubs)
        else:               # valid signatures to verify: produced on the GPU (untimed), spot-checked against the oracle below
            w["sigs"] = sigs = torch.empty((n, 64), dtype=torch.uint8, device=dev)
            w["pubs"] = pubs = torch.empty((n, 32), dtype=torch.uint8, device=dev)
            eng.sign_dev(sc, k, msgs, off, sigs, stream=stream)
            eng.mul_base_dev(sc, out_enc=pubs, stream=stream)
            w["step"] = lambda: eng.verify_dev(pubs, msgs, off, sigs, out, flavor=1, stream=stream)
        w["_keep"] = (msgs, off, pubs)
    torch.cuda.synchronize()
    return w


def time_workload(w, eng, steps, warmup, barrier):
    """W untimed steps, then exactly K steps between barrier+synchronize; HIP events per step (torch events on the
    launch stream) and per kernel launch (the engine's own event pairs, kyb_profile_begin / kyb_profile_read)"""
    import torch
    for _ in range(warmup):
        w["step"]()
    barrier()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    eng.profile_begin(10 * steps)
    t_start = time.perf_counter()
    for a, b in evs:
        a.record()
        w["step"]()
        b.record()
    barrier()
    elapsed = time.perf_counter() - t_start
    w["step_ms"] = [a.elapsed_time(b) for a, b in evs]
    w["launches"] = eng.profile_read(10 * steps)
    eng.profile_begin(0)
    return elapsed


def check_parity(w, orc, m, threads, dev):
    """m outputs of this rank (seeded sample) against the oracle; raises on the first difference"""
    import numpy as np
    import torch
    wl, n = w["wl"], w["n"]
    m = min(m, n)
    idx = np.sort(np.random.default_rng(0).choice(n, m, replace=False))
    tidx = torch.from_numpy(idx).to(dev)
    got = w["out"][tidx].cpu().numpy()
    sc = w["sc_np"][idx]
    if wl == "mul":
        want = orc.mul_batch(sc, w["pts"][tidx].cpu().numpy(), nthreads=threads)
    elif wl == "mul_base":
        want = orc.mul_base_batch(sc, nthreads=threads)
    elif wl == "sign":
        want = orc.schnorr_sign_batch(sc, w["k"][tidx].cpu().numpy(), [w["msg_list"][i] for i in idx], nthreads=threads)
    else:
        pub_s, sig_s = w["pubs"][tidx].cpu().numpy(), w["sigs"][tidx].cpu().numpy()
        msgs = [w["msg_list"][i] for i in idx]
        want = orc.verify_batch(1, pub_s, msgs, sig_s, nthreads=threads).reshape(-1, 1)
        if want.any() or not np.array_equal(sig_s, orc.schnorr_sign_batch(sc, w["k"][tidx].cpu().numpy(), msgs, nthreads=threads)):
            raise SystemExit("PARITY FAILURE: GPU-made signatures are not what the oracle signs / verifies")
    if not np.array_equal(got, want):
        raise SystemExit(f"PARITY FAILURE ({wl}): GPU output differs from the oracle")
    return m


def cpu_baseline(w, orc, threads):
    """the oracle (C port of the reference algorithm) on a bounded sample of the same workload"""
    import numpy as np
    wl, n = w["wl"], w["n"]
    per_core = {"mul": 1 << 15, "mul_base": 1 << 16, "sign": 1 << 15, "verify": 1 << 14}[wl]
    cnt_all = min(per_core * threads, n)
    pts_cpu = w["pts"][:cnt_all].cpu().numpy() if wl == "mul" else None
    k_cpu = w["k"][:cnt_all].cpu().numpy() if wl == "sign" else None
    pub_cpu = w["pubs"][:cnt_all].cpu().numpy() if wl == "verify" else None
    sig_cpu = w["sigs"][:cnt_all].cpu().numpy() if wl == "verify" else None

    def run(cnt, th):
        sub = np.arange(cnt)
        t1 = time.perf_counter()
        if wl == "mul":
            orc.mul_batch(w["sc_np"][sub], pts_cpu[sub], nthreads=th)
        elif wl == "mul_base":
            orc.mul_base_batch(w["sc_np"][sub], nthreads=th)
        elif wl == "sign":
            orc.schnorr_sign_batch(w["sc_np"][sub], k_cpu[sub], [w["msg_list"][i] for i in sub], nthreads=th)
        else:
            orc.verify_batch(1, pub_cpu[sub], [w["msg_list"][i] for i in sub], sig_cpu[sub], nthreads=th)
        return cnt / (time.perf_counter() - t1)

    one = run(min(per_core, n), 1)
    allc = run(cnt_all, threads)
    return {"value": round(allc, 1), "unit": UNIT[wl], "cores": threads, "kind": "port", "value_1core": round(one, 1),
            "sample": f"oracle/ed25519_oracle.c (C restatement of the reference algorithm, gcc -O3 -march=native; not the Rust binary): "
                      f"{cnt_all} items of the same workload on {threads} threads, {min(per_core, n)} items on 1 thread"}


def roofline(w, eng, steps):
    wl, n, keyed = w["wl"], w["n"], w["keyed"]
    products = dict(PRODUCTS)
    if wl == "sign" and keyed:
        products["sign"] = 62_250 + 2_000        # EdDSA::sign on a key object: one fixed-base mult + encode, two hashes, one sc_mul_add
    per_kernel = {}
    for name, ms in w["launches"]:
        per_kernel.setdefault(name, []).append(ms)
    dom = DOMINANT[wl] if DOMINANT[wl] in per_kernel else max(per_kernel, key=lambda k_: sum(per_kernel[k_]))
    dom_ms = sum(per_kernel[dom]) / len(per_kernel[dom])                  # average duration of ONE launch
    launches_per_step = len(per_kernel[dom]) / steps
    items_per_launch = n * (2 if wl == "sign" and dom == "k_mul_base" and not keyed else 1) / launches_per_step
    split = eng.get_option("finish.batched") and n >= eng.get_option("finish.min_items")
    dom_products = PRODUCTS_DOMINANT.get(dom, products[wl]) if split else products[wl]
    mad_rate = dom_products * items_per_launch / (dom_ms * 1e-3)
    executed = EXECUTED.get(dom, dom_products)
    if isinstance(executed, dict):
        executed = executed[eng.get_option("mul_base.radix") if n >= eng.get_option("finish.min_items") else 16]
    if dom == "k_mul_ladder" and wl == "verify":
        executed -= 3 * (5 * 100 + 4 * 55 + 10)      # the challenge h is < L < 2^253 by construction: the ladder starts three bits lower
    elif dom == "k_mul_ladder" and eng.get_option("ladder.skip_canonical"):
        # the bench's scalars are reduced mod L: every scalar of the launch is below 2^252 (checked on the device per launch) and the
        # ladder starts four bits lower
        executed -= 4 * (5 * 100 + 4 * 55 + 10)
    exec_rate = executed * items_per_launch / (dom_ms * 1e-3)
    avg_step_ms = sum(w["step_ms"]) / len(w["step_ms"])
    # HBM/fabric bytes per launch of the dominant kernel: NOT measured in this run — replayed from the PMC passes of the
    # same command committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc runs as the guide prescribes)
    traffic, traffic_source = None, None
    if n == DEFAULT_N[wl] and not keyed:
        for rnd in (PROFILE_ROUND, "r01"):
            pmc = os.path.join(ROOT, "profiles", rnd, f"{wl}_pmc_summary.json")
            if os.path.exists(pmc):
                d_ = json.load(open(pmc))["_derived"]
                traffic = round(d_["fetch_bytes_per_dispatch_corrected_x2"] + d_["write_bytes_per_dispatch"])
                traffic_source = f"profiles/{rnd}/{wl}_pmc_summary.json (rocprofv3 --pmc passes of this command, not this run)"
                break
    return {"bound": "valu-int", "kernel": dom, "achieved": round(mad_rate / 1e12, 3), "peak": PEAK_MAD_PER_S / 1e12,
            "unit": "T(32x32+64 mad)/s", "frac": round(mad_rate / PEAK_MAD_PER_S, 4),
            "peak_source": "measured: tools/microbench/valu_rates.hip (profiles/r01_valu_rates_mi355x.jsonl), all SIMDs issuing v_mad_u64_u32, ~1.9 GHz under load",
            "peak_nominal": round(PEAK_MAD_NOMINAL / 1e12, 2), "frac_nominal": round(mad_rate / PEAK_MAD_NOMINAL, 4),
            "algorithmic_mads_per_item": dom_products, "items_per_launch": int(items_per_launch),
            "avg_launch_ms": round(dom_ms, 4), "launches_timed": len(per_kernel[dom]),
            "executed_mads_per_item": executed, "executed_frac": round(exec_rate / PEAK_MAD_PER_S, 4),
            "executed_frac_nominal": round(exec_rate / PEAK_MAD_NOMINAL, 4),
            "traffic": traffic, "traffic_source": traffic_source,
            "step": {"avg_step_ms": round(avg_step_ms, 4), "kernels_ms": {k_: round(sum(v_) / steps, 4) for k_, v_ in per_kernel.items()},
                     "algorithmic_mads_per_item": products[wl],
                     "frac": round(products[wl] * n / (avg_step_ms * 1e-3) / PEAK_MAD_PER_S, 4)},
            "hbm": {"achieved": round(ALG_BYTES[wl] * n / (avg_step_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ALG_BYTES[wl] * n / (avg_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_item": ALG_BYTES[wl]}}


def device_identity(torch, local):
    p = torch.cuda.get_device_properties(local)
    ident = {"index": local, "name": p.name, "cus": p.multi_processor_count}
    for key in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id", "gcnArchName"):
        if hasattr(p, key):
            ident[key] = str(getattr(p, key))
    return ident


def small_call_latency(eng, orc):
    """What unmodified protocol code sees: ONE synchronous host-pointer call with 1 / 64 items (the one-item-per-wavefront kernels,
    DESIGN.md section 4 "Small batches"), median wall time in microseconds, outputs checked against the oracle.  Not a throughput figure."""
    import numpy as np
    import synth as _s
    s = _s.scalars(64, 71)
    k = _s.scalars(64, 72, b"k")
    enc, ext = eng.mul_base(s, want_ext=True)
    msgs = _s.messages(64, 73)
    sigs = eng.schnorr_sign(s, k, msgs)
    assert np.array_equal(enc, orc.mul_base_batch(s)) and np.array_equal(sigs, orc.schnorr_sign_batch(s, k, msgs))
    assert np.array_equal(eng.mul(k, pts_ext=ext), orc.mul_batch(k, ext)) and not eng.verify(enc, msgs, sigs, 1).any()

    def med(fn, reps=100):
        fn(); fn()
        ts = []
        for _ in range(reps):
            a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
        return round(sorted(ts)[len(ts) // 2] * 1e6, 1)

    # x = index + 1 of PubPoly::eval (poly.rs:461-464): a call whose multipliers are all below 2^64 starts its ladder below the leading zeros
    idx = np.zeros((64, 32), dtype=np.uint8)
    idx[:, 0] = np.arange(1, 65, dtype=np.uint8); idx[:, 1] = 2
    assert np.array_equal(eng.mul(idx, pts_ext=ext), orc.mul_batch(idx, ext))
    out = {"unit": "us per host-pointer call (median of 100)", "checked_against_oracle": True}
    for n in (1, 64):
        out[f"n={n}"] = {"mul_base": med(lambda: eng.mul_base(s[:n])), "mul": med(lambda: eng.mul(k[:n], pts_ext=ext[:n])),
                         "mul_by_10bit_index": med(lambda: eng.mul(idx[:n], pts_ext=ext[:n])),
                         "sign": med(lambda: eng.schnorr_sign(s[:n], k[:n], msgs[:n])), "verify": med(lambda: eng.verify(enc[:n], msgs[:n], sigs[:n], 1)),
                         "decode": med(lambda: eng.decode(enc[:n])), "encode": med(lambda: eng.encode(ext[:n]))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="mul", choices=["mul", "mul_base", "sign", "verify"])
    ap.add_argument("--keyed", action="store_true", help="sign: the signers hold their public keys (EdDSA objects, DSS long-term keys): "
                    "one fixed-base mult per signature instead of the two of schnorr::sign")
    ap.add_argument("--n", type=int, default=0, help="items per GPU (default 2^20, 2^18 for sign)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", action="store_true", help="time the primary workload only (no `workloads` object)")
    ap.add_argument("--check", type=int, default=16384, help="items verified against the oracle after timing (per workload)")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (kernel variant), repeatable")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import kyber_rs_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        backend = dist.get_backend()

    # ---- engine + base-point table (RCCL broadcast of rank 0's image over xGMI) ----
    from kyber_rs_amd import multi_gpu
    eng = kyber_rs_amd.Engine(local, build_table=(rank == 0))
    multi_gpu.distribute_base_table(eng, rank, world, dev, dist)

    for kv in args.opt:
        key, val = kv.split("=")
        eng.set_option(key, int(val))

    # what the job really ran on: every rank reports its device, rank 0 gathers
    ident = device_identity(torch, local)
    ident["rank"] = rank
    idents = [ident]
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)

    wl = args.workload
    n = args.n or DEFAULT_N[wl]
    # a dedicated (non-null) torch stream: the engine launches on it and the HIP events are recorded on it, so they
    # bracket exactly the kernels of each step
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream
    inp = Inputs(1 + rank, dev)          # every rank gets its own shard of the synthetic stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t0 = time.time()
    w = setup_workload(wl, n, eng, inp, dev, stream, args.keyed)
    gen_s = time.time() - t0
    elapsed = time_workload(w, eng, args.steps, args.warmup, barrier)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        import oracle_lib
        orc = oracle_lib.Oracle()
        # the GPU box gives one GPU a share of 16 host CPUs although os.cpu_count() reports the whole machine
        threads = max(1, min(len(os.sched_getaffinity(0)), 16))
        checked = check_parity(w, orc, args.check, threads, dev)
        cpu = cpu_baseline(w, orc, threads) if (world == 1 and not args.no_cpu_baseline) else None
        value = n * world * args.steps / elapsed
        line = {
            "metric": METRIC[wl], "value": round(value, 1), "unit": UNIT[wl], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 limbs (radix 2^25.5), u64 accumulators", "data": "synthetic",
            "config": {"workload": (WORKLOAD_TEXT[wl] + (", signers hold their public keys (one fixed-base mult per signature)" if wl == "sign" and args.keyed else "")) if not args.n else f"{wl} x {n} per GPU",
                       "items_per_gpu": n, "sharding": f"independent shards x{world}, no data-path collective; one RCCL table broadcast at init",
                       "options": {k_: eng.get_option(k_) for k_ in ("mul.algo", "mul.ladder_waves", "mul.select", "mul_base.radix", "mul_base.select", "mul_base.block", "finish.batched", "finish.min_items")}},
            "roofline": roofline(w, eng, args.steps),
            "cpu_baseline": cpu,
            "parity_checked_items": checked, "input_gen_s": round(gen_s, 2),
            "ranks_seen": len(idents), "dist_backend": backend, "devices": idents,
        }
        # ---- the other single-GPU configurations, each in its own timed region outside the primary one ----
        if world == 1 and not args.only and not args.n:
            others = {}
            del w
            for owl in ("mul_base", "sign", "verify", "mul"):
                if owl == wl:
                    continue
                on = DEFAULT_N[owl]
                ow = setup_workload(owl, on, eng, inp, dev, stream, False)
                oel = time_workload(ow, eng, args.steps, args.warmup, barrier)
                ochk = check_parity(ow, orc, args.check, threads, dev)
                others[owl] = {"metric": METRIC[owl], "value": round(on * args.steps / oel, 1), "unit": UNIT[owl],
                               "ms_per_step": round(oel / args.steps * 1e3, 4), "steps": args.steps, "warmup": args.warmup,
                               "config": {"workload": WORKLOAD_TEXT[owl], "items_per_gpu": on},
                               "roofline": roofline(ow, eng, args.steps),
                               "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(ow, orc, threads),
                               "parity_checked_items": ochk}
                del ow
            line["workloads"] = others
            line["small_calls"] = small_call_latency(eng, orc)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
