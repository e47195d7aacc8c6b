#!/usr/bin/env python3
"""Headline benchmark: Ed25519 scalar-mults/sec on MI355X (BASELINE.json `metric`).

  python bench.py --gpus N --steps K --warmup W [--workload mul|mul_base|sign] [--n ITEMS_PER_GPU]

A "step" = one pass of the hot path over one batch that is already resident in HBM:
  mul       2^20 variable-base mults, random scalars + random points   (BASELINE configs[1], default)
  mul_base  2^20 fixed-base mults                                       (configs[2])
  sign      2^18 Schnorr signatures, 32-byte messages                   (configs[3])
For N > 1 the driver launches one process per GPU (torch.distributed.run); every rank owns its own
shard of N x ITEMS_PER_GPU independent items (weak scaling, no data-path collective).  The only
collective is the one-time RCCL broadcast of the 168 KiB base-point table image built on rank 0.

Rank 0 prints ONE JSON line.  `value` is whole-job items/s over the timed K steps (barrier +
synchronize on both sides, max over ranks).  `roofline` prices the dominant kernel against the
MEASURED v_mad_u64_u32 issue peak of the chip (profiles/r01_valu_rates_mi355x.jsonl): the path is
integer-VALU bound by construction (BASELINE.json north_star), not HBM or MFMA bound, so the object
carries `bound: "valu-int"` and additionally reports the (negligible) algorithmic HBM rate.
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
# reference's: 256-step ladder; 52 radix-32 or 64 radix-16 mixed additions) — reported next to the
# algorithmic figure so that `frac` (algorithmic, may exceed 1 where less work is done) is not mistaken
# for pipe utilisation
EXECUTED = {"k_mul": 188_640, "k_mul_ladder": 256 * (5 * 100 + 4 * 55 + 10) + 2_300, "k_mul_base": {64: 43 * 700, 32: 52 * 700, 16: 64 * 700}, "k_sign": 2 * 64 * 700 + 15_270}
ALG_BYTES = {"mul": 32 + 160 + 32, "mul_base": 32 + 32, "sign": 32 + 32 + 32 + 64, "verify": 32 + 64 + 32 + 1}
UNIT = {"mul": "variable-base scalar-mults/s", "mul_base": "fixed-base scalar-mults/s", "sign": "signatures/s", "verify": "verifications/s"}
DOMINANT = {"mul": "k_mul_ladder", "mul_base": "k_mul_base", "sign": "k_mul_base", "verify": "k_mul_ladder"}
# measured on MI355X: 26.8e12 v_mad_u64_u32 lane-ops/s with every SIMD issuing (8 waves/SIMD, clock
# settles at ~1.9 GHz under this load) — tools/microbench/valu_rates.hip
PEAK_MAD_PER_S = 26.8e12
HBM_PEAK_GBS = 8000.0


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
    ap.add_argument("--check", type=int, default=1024, help="items verified against the oracle after timing")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (kernel variant), repeatable")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import kyber_rs_amd
    import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    # ---- engine + base-point table (RCCL broadcast of rank 0's image over xGMI) ----
    from kyber_rs_amd import multi_gpu
    eng = kyber_rs_amd.Engine(local, build_table=(rank == 0))
    multi_gpu.distribute_base_table(eng, rank, world, dev, dist)

    for kv in args.opt:
        key, val = kv.split("=")
        eng.set_option(key, int(val))

    wl = args.workload
    n = args.n or ((1 << 18) if wl == "sign" else (1 << 20))
    seed = 1 + rank          # every rank gets its own shard of the synthetic stream
    t0 = time.time()
    # a dedicated (non-null) torch stream: the engine launches on it and the HIP events below are
    # recorded on it, so they bracket exactly the kernel of each step
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    stream = tstream.cuda_stream

    # ---- synthetic inputs, resident in HBM before the timed region ----
    sc_np = synth.scalars(n, seed)
    sc = torch.from_numpy(sc_np).to(dev)
    out = torch.empty((n, 64 if wl == "sign" else (1 if wl == "verify" else 32)), dtype=torch.uint8, device=dev)
    pts = k = msgs = off = pubs = sigs = None
    if wl == "mul":
        psc = torch.from_numpy(synth.scalars(n, seed, b"point")).to(dev)
        pts = torch.empty((n, 40), dtype=torch.int32, device=dev)        # point_i = (hash mod L) * B, reference limbs
        eng.mul_base_dev(psc, out_ext=pts, stream=stream)
    elif wl in ("sign", "verify"):
        k = torch.from_numpy(synth.scalars(n, seed, b"k")).to(dev)
        msg_list = synth.messages(n, seed)
        msgs = torch.from_numpy(np.frombuffer(b"".join(msg_list), dtype=np.uint8).copy()).to(dev)
        off = torch.arange(0, 32 * (n + 1), 32, dtype=torch.int32, device=dev)
        if wl == "sign" and args.keyed:
            pubs = torch.empty((n, 32), dtype=torch.uint8, device=dev)
            eng.mul_base_dev(sc, out_enc=pubs, stream=stream)
        if wl == "verify":      # valid signatures to verify: produced on the GPU (untimed), spot-checked below
            sigs = torch.empty((n, 64), dtype=torch.uint8, device=dev)
            pubs = torch.empty((n, 32), dtype=torch.uint8, device=dev)
            eng.sign_dev(sc, k, msgs, off, sigs, stream=stream)
            eng.mul_base_dev(sc, out_enc=pubs, stream=stream)
    torch.cuda.synchronize()
    gen_s = time.time() - t0

    def step():
        if wl == "mul":
            eng.mul_dev(sc, pts_ext=pts, out_enc=out, stream=stream)
        elif wl == "mul_base":
            eng.mul_base_dev(sc, out_enc=out, stream=stream)
        elif wl == "sign":
            eng.sign_dev(sc, k, msgs, off, out, stream=stream, pubs=pubs if args.keyed else None)
        else:
            eng.verify_dev(pubs, msgs, off, sigs, out, flavor=1, stream=stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # HIP events on the launch stream: per step (torch events) and per kernel launch (the engine's own
    # event pairs around every kernel it launches, kyb_profile_begin / kyb_profile_read)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    eng.profile_begin(8 * args.steps)
    t_start = time.perf_counter()
    for a, b in evs:
        a.record()
        step()
        b.record()
    barrier()
    elapsed = time.perf_counter() - t_start
    step_ms = [a.elapsed_time(b) for a, b in evs]
    launches = eng.profile_read(8 * args.steps)
    eng.profile_begin(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- parity spot-check of this rank's outputs against the oracle (outside the timed region) ----
    checked = 0
    cpu = None
    if rank == 0:
        import oracle_lib
        orc = oracle_lib.Oracle()
        m = min(args.check, n)
        idx = np.random.default_rng(0).choice(n, m, replace=False)
        got = out[torch.from_numpy(idx).to(dev)].cpu().numpy()
        # the GPU box gives one GPU a share of 16 host CPUs although os.cpu_count() reports the whole machine
        threads = max(1, min(len(os.sched_getaffinity(0)), 16))
        if wl == "mul":
            want = orc.mul_batch(sc_np[idx], pts[torch.from_numpy(idx).to(dev)].cpu().numpy(), nthreads=threads)
        elif wl == "mul_base":
            want = orc.mul_base_batch(sc_np[idx], nthreads=threads)
        elif wl == "sign":
            want = orc.schnorr_sign_batch(sc_np[idx], k[torch.from_numpy(idx).to(dev)].cpu().numpy(), [msg_list[i] for i in idx], nthreads=threads)
        else:
            tidx = torch.from_numpy(idx).to(dev)
            pub_s, sig_s = pubs[tidx].cpu().numpy(), sigs[tidx].cpu().numpy()
            want = np.array([[orc.verify(1, bytes(pub_s[j]), msg_list[i], bytes(sig_s[j]))] for j, i in enumerate(idx)], dtype=np.uint8)
            if want.any() or not np.array_equal(sig_s, orc.schnorr_sign_batch(sc_np[idx], k[tidx].cpu().numpy(), [msg_list[i] for i in idx], nthreads=threads)):
                raise SystemExit("PARITY FAILURE: GPU-made signatures are not what the oracle signs / verifies")
        if not np.array_equal(got, want):
            raise SystemExit("PARITY FAILURE: GPU output differs from the oracle")
        checked = m

        # ---- CPU baseline: the oracle (C port of the reference algorithm) on a bounded sample ----
        if world == 1 and not args.no_cpu_baseline:
            per_core = {"mul": 1 << 15, "mul_base": 1 << 16, "sign": 1 << 15, "verify": 1 << 14}[wl]

            def run(cnt, th):
                sub = np.arange(cnt) % n
                t1 = time.perf_counter()
                if wl == "mul":
                    orc.mul_batch(sc_np[sub], pts_cpu[sub], nthreads=th)
                elif wl == "mul_base":
                    orc.mul_base_batch(sc_np[sub], nthreads=th)
                elif wl == "sign":
                    orc.schnorr_sign_batch(sc_np[sub], k_cpu[sub], [msg_list[i] for i in sub], nthreads=th)
                else:
                    orc.verify_batch(1, pub_cpu[sub], [msg_list[i] for i in sub], sig_cpu[sub], nthreads=th)
                return cnt / (time.perf_counter() - t1)

            cnt_all = per_core * threads
            pts_cpu = pts[: min(n, cnt_all)].cpu().numpy() if wl == "mul" else None
            k_cpu = k[: min(n, cnt_all)].cpu().numpy() if wl == "sign" else None
            pub_cpu = pubs[: min(n, cnt_all)].cpu().numpy() if wl == "verify" else None
            sig_cpu = sigs[: min(n, cnt_all)].cpu().numpy() if wl == "verify" else None
            one = run(min(per_core, n), 1)
            allc = run(min(cnt_all, n), threads)
            cpu = {"value": round(allc, 1), "unit": UNIT[wl], "cores": threads, "kind": "port",
                   "value_1core": round(one, 1),
                   "sample": f"oracle/ed25519_oracle.c (C restatement of the reference algorithm, gcc -O3 -march=native; not the Rust binary): "
                             f"{min(cnt_all, n)} items of the same workload on {threads} threads, {min(per_core, n)} items on 1 thread"}

    if wl == "sign" and args.keyed:
        PRODUCTS["sign"] = 62_250 + 2_000        # EdDSA::sign on a key object: one fixed-base mult + encode, two hashes, one sc_mul_add
    if rank == 0:
        total_items = n * world * args.steps
        value = total_items / elapsed
        per_kernel = {}
        for name, ms in launches:
            per_kernel.setdefault(name, []).append(ms)
        dom = DOMINANT[wl] if DOMINANT[wl] in per_kernel else max(per_kernel, key=lambda k_: sum(per_kernel[k_]))
        dom_ms = sum(per_kernel[dom]) / len(per_kernel[dom])                  # average duration of ONE launch
        launches_per_step = len(per_kernel[dom]) / args.steps
        items_per_launch = n * (2 if wl == "sign" and dom == "k_mul_base" and not args.keyed else 1) / launches_per_step
        dom_products = PRODUCTS_DOMINANT.get(dom, PRODUCTS[wl]) if eng.get_option("finish.batched") and n >= eng.get_option("finish.min_items") else PRODUCTS[wl]
        mad_rate = dom_products * items_per_launch / (dom_ms * 1e-3)
        executed = EXECUTED.get(dom, dom_products)
        if isinstance(executed, dict):
            executed = executed[eng.get_option("mul_base.radix") if n >= eng.get_option("finish.min_items") else 16]
        if wl == "verify" and dom == "k_mul_ladder":
            executed -= 3 * (5 * 100 + 4 * 55 + 10)      # the challenge h is < L < 2^253: the ladder starts three bits lower
        avg_step_ms = sum(step_ms) / len(step_ms)
        # HBM/fabric bytes per launch from the PMC passes of the same command (profiles/, FETCH_SIZE x2 + WRITE_SIZE)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01", f"{wl}_pmc_summary.json")
        if os.path.exists(pmc) and n == ((1 << 18) if wl == "sign" else (1 << 20)) and not args.keyed:      # the sizes the PMC passes ran at
            d_ = json.load(open(pmc))["_derived"]
            traffic = round(d_["fetch_bytes_per_dispatch_corrected_x2"] + d_["write_bytes_per_dispatch"])
        line = {
            "metric": {"mul": "Ed25519 scalar-mults/sec", "mul_base": "Ed25519 scalar-mults/sec", "sign": "Ed25519 Schnorr signatures/sec", "verify": "Ed25519 Schnorr verifications/sec"}[wl],
            "value": round(value, 1), "unit": UNIT[wl], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32 limbs (radix 2^25.5), u64 accumulators", "data": "synthetic",
            "config": {"workload": {"mul": "2^20 variable-base scalar-mults, random scalars+points, reference-limb points in, 32-byte encodings out",
                                    "mul_base": "2^20 fixed-base (generator) scalar-mults, 32-byte encodings out",
                                    "sign": "2^18 Schnorr signs, 32-byte messages, 64-byte signatures out",
                                    "verify": "2^20 Schnorr verifications with the reference's checks, 32-byte messages, status bytes out"}[wl] + (", signers hold their public keys (one fixed-base mult per signature)" if wl == "sign" and args.keyed else "") if not args.n else f"{wl} x {n} per GPU",
                       "items_per_gpu": n, "sharding": f"independent shards x{world}, no data-path collective; one RCCL table broadcast at init",
                       "options": {k_: eng.get_option(k_) for k_ in ("mul.algo", "mul.ladder_waves", "mul.select", "mul_base.radix", "mul_base.select", "mul_base.block", "finish.batched", "finish.min_items")}},
            "roofline": {"bound": "valu-int", "kernel": dom, "achieved": round(mad_rate / 1e12, 3), "peak": PEAK_MAD_PER_S / 1e12,
                         "unit": "T(32x32+64 mad)/s", "frac": round(mad_rate / PEAK_MAD_PER_S, 4),
                         "algorithmic_mads_per_item": dom_products, "items_per_launch": int(items_per_launch),
                         "avg_launch_ms": round(dom_ms, 4), "launches_timed": len(per_kernel[dom]),
                         "executed_mads_per_item": executed, "executed_frac": round(executed * items_per_launch / (dom_ms * 1e-3) / PEAK_MAD_PER_S, 4),
                         "traffic": traffic,
                         "step": {"avg_step_ms": round(avg_step_ms, 4), "kernels_ms": {k_: round(sum(v_) / args.steps, 4) for k_, v_ in per_kernel.items()},
                                  "algorithmic_mads_per_item": PRODUCTS[wl],
                                  "frac": round(PRODUCTS[wl] * n / (avg_step_ms * 1e-3) / PEAK_MAD_PER_S, 4)},
                         "hbm": {"achieved": round(ALG_BYTES[wl] * n / (avg_step_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(ALG_BYTES[wl] * n / (avg_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_item": ALG_BYTES[wl]}},
            "cpu_baseline": cpu,
            "parity_checked_items": checked, "input_gen_s": round(gen_s, 2),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
