#!/usr/bin/env python3
"""Headline benchmark: Ed25519 scalar-mults/sec on MI355X (BASELINE.json `metric`).

  python bench.py --gpus N --steps K --warmup W [--mode ranks|group] [--workload mul|mul_enc|mul_base|sign|verify] [--n ITEMS_PER_GPU]

A "step" = one pass of the hot path over one batch that is already resident in HBM:
  mul       2^20 variable-base mults, random scalars + random points   (BASELINE configs[1], default, the line's `value`)
  mul_enc   the same from 32-byte wire encodings: unmarshal_binary inside the timed region (primary input form)
  mul_base  2^20 fixed-base mults                                       (configs[2])
  sign      2^18 Schnorr signatures, 32-byte messages                   (configs[3])
  verify    2^20 Schnorr verifications with the reference's checks      

N > 1 = BASELINE configs[4]: 2^24 variable-base scalar-mults sharded across the N GPUs — rank r owns items [floor(r 2^24 / N),
floor((r + 1) 2^24 / N)), no data-path collective; the only collective is the one-time broadcast of the base-point table image built on
rank 0 (RCCL over xGMI).  The total is fixed as N grows (`scaling: "strong"`); `--scaling weak` gives every GPU the single-GPU batch
instead (2^20 items each).  EVERY rank checks a sample of its own shard against the oracle and the verdicts are gathered before rank 0
prints: one failing rank fails the job.  `python bench.py --gpus 1 --n 16777216` is the same total on one GPU.
  --mode ranks (default, the mode `value` is quoted in): one process per GPU over torch.distributed / RCCL.  Started either by
      the driver's launcher (torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or plainly as
      `python bench.py --gpus N ...`: the parent then spawns the N rank processes itself BEFORE it makes any GPU call (it never
      imports torch), relays rank 0's JSON line and exits with the ranks' status.
  --mode group: ONE process, kyb_group_create(range(N)) — the in-library layout a Rust host uses: one context per GPU, the table
      image moved by the library's own ncclBroadcast (the line reports `table_transport`), device-resident shards queued on all
      GPUs through kyb_group_mul_batch_dev.

Rank 0 prints ONE JSON line.  `value` is whole-job items/s of the PRIMARY workload over the timed K steps (barrier + synchronize on
both sides, max over ranks).  At N = 1 the other single-GPU configurations are then timed the same way, each in its own timed
region OUTSIDE the primary one, and reported under `workloads`; `small_calls` (and `mid_size_calls`, 8,192 items) give the wall time of ONE synchronous host-pointer
call with 1 / 64 items, and the `mul` workload carries `host_pinned` / `host_pageable`: the rate of kyb_mul_batch on 2^20 items in
host memory (PCIe-inclusive; never `value`).

`roofline` prices the dominant kernel against the v_mad_u64_u32 issue peak: the path is integer-VALU bound by construction
(BASELINE.json north_star), not HBM or MFMA bound, so the object carries `bound: "valu-int"` and additionally the (negligible)
algorithmic HBM rate.  `peak` is MEASURED IN THIS RUN on this chip (kyb_diag_mad_peak: every SIMD issuing nothing but
v_mad_u64_u32 for >= 50 ms, outside every timed region) together with the clock the chip held under that load; `peak_nominal` =
1024 SIMDs x 64 lanes / 4 cycles x 2.4 GHz.  `kernel_clock_ghz` is the clock the dominant kernel itself ran at (wave stamps,
csrc/diag_stamp.h, in extra untimed steps) and `issue_share` = wavefront multiply-adds per SIMD x measured SIMD cycles per
multiply-add / (launch duration x kernel clock): the share of the kernel's own SIMD cycles in which the multiplier issues.
`achieved` / `frac` price the multiply-adds of the REFERENCE's algorithm where this repository's kernel executes about as many
(`priced: "algorithmic"`); where it executes far fewer (fixed base: 43 additions instead of 64) the headline pair prices the
EXECUTED multiply-adds (`priced: "executed"`) and the algorithmic figure is kept as `frac_vs_reference_algorithm`.
`cpu_baseline` times the oracle — a C port of the reference algorithm, NOT the Rust binary — on this box's host cores."""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic 32x32->64 multiply(-add)s per unit, fixed by the reference's algorithm:
# whole step (mult + encode) and the dominant kernel alone (mult with projective output; the encode is
# k_finish's work when the batched finish is on)
# verify (eddsa_sig.rs:159-212): 2 decodes (2 x 16,000) + 2 has_small_order encodes (2 x 15,270) + fixed-base
# (46,980) + variable-base (188,640) + add (900) + eq = 2 encodes (2 x 15,270)
PRODUCTS = {"mul": 203_910, "mul_enc": 203_910 + 16_000, "mul_base": 62_250, "sign": 124_500, "verify": 329_600}
PRODUCTS_DOMINANT = {"k_mul": 188_640, "k_mul_ladder": 188_640, "k_mul_base": 46_980, "k_sign": 124_500}
# multiply-adds the kernels of THIS repository actually execute per item (the algorithms differ from the
# reference's: 256-step ladder; 43 radix-64 / 52 radix-32 / 64 radix-16 mixed additions)
EXECUTED = {"k_mul": 188_640, "k_mul_ladder": 256 * (5 * 100 + 4 * 55 + 10) + 2_300, "k_mul_base": {64: 43 * 700, 32: 52 * 700, 16: 64 * 700}, "k_sign": 2 * 64 * 700 + 15_270}
ALG_BYTES = {"mul": 32 + 160 + 32, "mul_enc": 32 + 32 + 32 + 1, "mul_base": 32 + 32, "sign": 32 + 32 + 32 + 64, "verify": 32 + 64 + 32 + 1}
UNIT = {"mul": "variable-base scalar-mults/s", "mul_enc": "variable-base scalar-mults/s", "mul_base": "fixed-base scalar-mults/s", "sign": "signatures/s",
        "verify": "verifications/s"}
METRIC = {"mul": "Ed25519 scalar-mults/sec", "mul_enc": "Ed25519 scalar-mults/sec", "mul_base": "Ed25519 scalar-mults/sec", "sign": "Ed25519 Schnorr signatures/sec",
          "verify": "Ed25519 Schnorr verifications/sec"}
DOMINANT = {"mul": "k_mul_ladder", "mul_enc": "k_mul_ladder", "mul_base": "k_mul_base", "sign": "k_mul_base", "verify": "k_mul_ladder"}
WORKLOAD_TEXT = {"mul": "2^20 variable-base scalar-mults, random scalars+points, reference-limb points in, 32-byte encodings out",
                 "mul_enc": "2^20 variable-base scalar-mults, random scalars+points, 32-byte wire encodings in (decoded on the GPU inside the step), 32-byte encodings out",
                 "mul_base": "2^20 fixed-base (generator) scalar-mults, 32-byte encodings out",
                 "sign": "2^18 Schnorr signs, 32-byte messages, 64-byte signatures out",
                 "verify": "2^20 Schnorr verifications with the reference's checks, 32-byte messages, status bytes out"}
# data-sheet figure: 256 CUs x 4 SIMDs x 64 lanes per 4 cycles (quarter rate of the 16-lane VALU pass) x 2.4 GHz
PEAK_MAD_NOMINAL = 1024 * 64 / 4 * 2.4e9
HBM_PEAK_GBS = 8000.0
DEFAULT_N = {"mul": 1 << 20, "mul_enc": 1 << 20, "mul_base": 1 << 20, "sign": 1 << 18, "verify": 1 << 20}
PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01")
OPTION_KEYS = ("device.cus", "coop.max_items", "coop.base_max_items", "coop.ladder_max_items", "ladder.skip_canonical", "ladder.pair_max_items")      # (kernel variants are not options of the product library)


# ------------------------------------------------------------------------------------------------------------------------------
# N > 1 started without a launcher: the parent spawns the ranks (and touches neither torch nor the GPU)
# ------------------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n_ranks):
    """One child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0's stdout relayed; returns the exit status.
    This process has made no HIP call (no torch import, no kyb_init): nothing that owns a GPU is forked or replaced."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), KYB_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))

    def relay():                     # stdout carries the ONE JSON line; whatever else a library prints there (gloo, RCCL banners) goes to stderr
        for line in procs[0].stdout:
            out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            out.write(line)
            out.flush()
    t = threading.Thread(target=relay, daemon=True)
    t.start()
    status = 0
    live = set(range(n_ranks))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and status == 0:
                status = rc
                print(f"bench.py: rank {r} exited with status {rc}; stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()          # exactly the processes this parent started
        time.sleep(0.05)
    t.join(timeout=10)
    return status


# ------------------------------------------------------------------------------------------------------------------------------
# device runtime: the GPU (torch for memory / streams / events), or the CPU stand-in of tests/test_multi_gpu_cpu.py
# ------------------------------------------------------------------------------------------------------------------------------
class GpuRuntime:
    standin = False

    def __init__(self, local):
        import torch
        self.torch = torch
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        # a dedicated (non-null) torch stream: the engine launches on it and the HIP events are recorded on it, so they
        # bracket exactly the kernels of each step
        self.tstream = torch.cuda.Stream(device=self.dev)
        torch.cuda.set_stream(self.tstream)
        self.stream = self.tstream.cuda_stream

    def sync(self):
        self.torch.cuda.synchronize()

    def event_pairs(self, k):
        ev = self.torch.cuda.Event
        return [(ev(enable_timing=True), ev(enable_timing=True)) for _ in range(k)]

    @staticmethod
    def elapsed_ms(a, b):
        return a.elapsed_time(b)


class StandinRuntime:
    """no GPU: CPU tensors, wall-clock 'events' — only for the plumbing test (spawn, rendezvous, broadcast, shards, one JSON line)"""
    standin = True

    class _Ev:
        def record(self):
            self.t = time.perf_counter()

    def __init__(self, local):
        import torch
        self.torch = torch
        self.dev = torch.device("cpu")
        self.stream = 0

    def sync(self):
        pass

    def event_pairs(self, k):
        return [(self._Ev(), self._Ev()) for _ in range(k)]

    @staticmethod
    def elapsed_ms(a, b):
        return (b.t - a.t) * 1e3


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


def setup_workload(wl, n, eng, inp, rt, keyed):
    """-> dict with the step closure and everything the parity check / CPU baseline need; inputs end up resident in HBM"""
    import numpy as np
    torch, dev, stream = rt.torch, rt.dev, rt.stream
    w = {"wl": wl, "n": n, "keyed": keyed}
    sc = inp.scalars(n)
    w["sc_np"] = inp.scalars_np(n)
    w["out"] = out = torch.empty((n, 64 if wl == "sign" else (1 if wl == "verify" else 32)), dtype=torch.uint8, device=dev)
    if wl in ("mul", "mul_enc"):
        psc = inp.scalars(n, b"point")
        w["pts"] = pts = torch.empty((n, 40), dtype=torch.int32, device=dev)        # point_i = (hash mod L) * B, reference limbs
        if wl == "mul":
            eng.mul_base_dev(psc, out_ext=pts, stream=stream)
            w["step"] = lambda: eng.mul_dev(sc, pts_ext=pts, out_enc=out, stream=stream)
        else:
            w["penc"] = penc = torch.empty((n, 32), dtype=torch.uint8, device=dev)  # the same points as they travel: marshal_binary
            w["ok"] = ok = torch.empty((n,), dtype=torch.uint8, device=dev)
            eng.mul_base_dev(psc, out_enc=penc, out_ext=pts, stream=stream)
            w["step"] = lambda: eng.mul_dev(sc, pts_enc=penc, out_enc=out, ok=ok, stream=stream)
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
            w["pubs_ext"] = pubs_ext = torch.empty((n, 40), dtype=torch.int32, device=dev)
            eng.mul_base_dev(sc, out_enc=pubs, out_ext=pubs_ext, stream=stream)
            w["step"] = lambda: eng.verify_dev(pubs, msgs, off, sigs, out, flavor=1, stream=stream)
            # the same verifications with the public keys as points (schnorr::verify's own signature, kyb_verify_points_batch): timed after the
            # primary region and reported next to it
            w["step_points"] = lambda: eng.verify_points_dev(pubs_ext, msgs, off, sigs, out, flavor=1, stream=stream)
        w["_keep"] = (msgs, off, pubs)
    rt.sync()
    return w


def time_workload(w, eng, rt, steps, warmup, barrier):
    """W untimed steps, then exactly K steps between barrier+synchronize; HIP events per step (torch events on the
    launch stream) and per kernel launch (the engine's own event pairs, kyb_profile_begin / kyb_profile_read)"""
    for _ in range(warmup):
        w["step"]()
    barrier()
    evs = rt.event_pairs(steps)
    eng.profile_begin(10 * steps)
    t_start = time.perf_counter()
    for a, b in evs:
        a.record()
        w["step"]()
        b.record()
    barrier()
    elapsed = time.perf_counter() - t_start
    w["step_ms"] = [rt.elapsed_ms(a, b) for a, b in evs]
    w["launches"] = eng.profile_read(10 * steps)
    eng.profile_begin(0)
    return elapsed


def kernel_clock(w, eng, rt, steps=3, warm=10):
    """the clock the stamped kernels of this step (k_mul_ladder, k_mul_base64) run at: extra UNTIMED steps with the wave stamps on, behind
    at least `warm` plain steps and 0.25 s of load (the parity check before this leaves the GPU idle for seconds, and the first launches
    after an idle period run at a lower clock than the timed region saw: tools/base_clock_probe.py, 1.9 GHz cold, 2.36 after 50 ms)"""
    torch = rt.torch
    buf = torch.zeros(8, dtype=torch.int64, device=rt.dev)
    t0 = time.perf_counter()
    done = 0
    while done < warm or time.perf_counter() - t0 < 0.25:      # at least a quarter of a second of load: the clock needs ~50 ms to come up from idle
        for _ in range(warm):
            w["step"]()
        rt.sync()
        done += warm
    eng.wave_stamps(buf)
    try:
        for _ in range(steps):
            w["step"]()
        rt.sync()
    finally:
        eng.wave_stamps(None)
    s = [int(v) % (1 << 64) for v in buf.cpu().tolist()]
    cyc, ticks, waves = (s[2] - s[0]) % (1 << 64), (s[3] - s[1]) % (1 << 64), s[4]
    if not waves or not ticks:
        return None
    return {"ghz": cyc / ticks * 0.1, "waves_stamped": waves, "steps": steps, "mean_wave_cycles": cyc / waves}


def check_parity(w, orc, m, threads, dev):
    """m outputs of this rank (seeded sample) against the oracle; raises on the first difference"""
    import numpy as np
    import torch
    wl, n = w["wl"], w["n"]
    m = min(m, n)
    if m <= 0:
        return 0
    idx = np.sort(np.random.default_rng(0).choice(n, m, replace=False))
    tidx = torch.from_numpy(idx).to(dev)
    got = w["out"][tidx].cpu().numpy()
    sc = w["sc_np"][idx]
    if wl == "mul":
        want = orc.mul_batch(sc, w["pts"][tidx].cpu().numpy(), nthreads=threads)
    elif wl == "mul_enc":
        # the oracle goes the reference's way: unmarshal_binary of the wire bytes, then mul; and the wire bytes the step consumed are what
        # marshal_binary gives for the sampled points
        want, ok_want = orc.mul_enc_batch(sc, w["penc"][tidx].cpu().numpy(), nthreads=threads)
        if not np.array_equal(w["ok"][tidx].cpu().numpy(), ok_want) or not np.array_equal(w["penc"][tidx].cpu().numpy(), orc.encode_batch(w["pts"][tidx].cpu().numpy(), nthreads=threads)):
            raise SystemExit("PARITY FAILURE (mul_enc): decode flags / wire encodings of the operands differ from the oracle's")
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
    per_core = {"mul": 1 << 15, "mul_enc": 1 << 15, "mul_base": 1 << 16, "sign": 1 << 15, "verify": 1 << 14}[wl]
    cnt_all = min(per_core * threads, n)
    pts_cpu = w["pts"][:cnt_all].cpu().numpy() if wl == "mul" else None
    enc_cpu = w["penc"][:cnt_all].cpu().numpy() if wl == "mul_enc" else None
    k_cpu = w["k"][:cnt_all].cpu().numpy() if wl == "sign" else None
    pub_cpu = w["pubs"][:cnt_all].cpu().numpy() if wl == "verify" else None
    sig_cpu = w["sigs"][:cnt_all].cpu().numpy() if wl == "verify" else None

    def run(cnt, th):
        sub = np.arange(cnt)
        t1 = time.perf_counter()
        if wl == "mul":
            orc.mul_batch(w["sc_np"][sub], pts_cpu[sub], nthreads=th)
        elif wl == "mul_enc":       # unmarshal_binary of every operand, then the multiplication (what the reference does with a received point)
            orc.mul_enc_batch(w["sc_np"][sub], enc_cpu[sub], nthreads=th)
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


def roofline(w, eng, steps, peak, clock, cus):
    """peak: kyb_diag_mad_peak of this run; clock: kernel_clock() of this workload (or None)"""
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
    split = True          # the product library has one fixed-base / finish form: radix 64, projective staging, batched inversion
    dom_products = PRODUCTS_DOMINANT.get(dom, products[wl]) if split else products[wl]
    alg_rate = dom_products * items_per_launch / (dom_ms * 1e-3)
    executed = EXECUTED.get(dom, dom_products)
    if isinstance(executed, dict):
        executed = executed[64]
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
        for rnd in PROFILE_ROUNDS:
            pmc = os.path.join(ROOT, "profiles", rnd, f"{wl}_pmc_summary.json")
            if os.path.exists(pmc):
                d_ = json.load(open(pmc))["_derived"]
                import kyber_rs_amd
                here, there = kyber_rs_amd.kernel_sources_id(), d_.get("kernel_sources_id")
                if there == here:
                    traffic = round(d_["fetch_bytes_per_dispatch_corrected_x2"] + d_["write_bytes_per_dispatch"])
                    traffic_source = f"profiles/{rnd}/{wl}_pmc_summary.json (rocprofv3 --pmc passes of this command on this kernel code, id {here}; not this run)"
                else:      # counters of other code say nothing about this build: no number rather than a stale one
                    traffic_source = f"none: the newest PMC summary (profiles/{rnd}/{wl}_pmc_summary.json) was collected on kernel code {there}, this is {here}"
                break
    pk = peak["mads_per_s"]
    # the kernel executes about the reference's work (ladder: 0.98 of it) -> price the algorithmic figure; far less (43 of 64 additions)
    # -> the algorithmic fraction would exceed what the multiplier can issue: price the executed multiply-adds, keep the other beside it
    priced = "algorithmic" if executed >= 0.9 * dom_products else "executed"
    head_rate = alg_rate if priced == "algorithmic" else exec_rate
    r = {"bound": "valu-int", "kernel": dom, "priced": priced, "achieved": round(head_rate / 1e12, 3), "peak": round(pk / 1e12, 3),
         "unit": "T(32x32+64 mad)/s", "frac": round(head_rate / pk, 4),
         "peak_source": "this run: kyb_diag_mad_peak (every SIMD issuing v_mad_u64_u32 only, 8 wavefronts per SIMD, outside the timed regions)",
         "peak_clock_ghz": round(peak["clock_ghz"], 4), "peak_simd_cycles_per_mad": round(peak["simd_cycles_per_mad"], 4), "peak_kernel_ms": round(peak["kernel_ms"], 2),
         "peak_nominal": round(PEAK_MAD_NOMINAL / 1e12, 2), "frac_nominal": round(head_rate / PEAK_MAD_NOMINAL, 4),
         "algorithmic_mads_per_item": dom_products, "executed_mads_per_item": executed, "items_per_launch": int(items_per_launch),
         "avg_launch_ms": round(dom_ms, 4), "launches_timed": len(per_kernel[dom]),
         "executed_achieved": round(exec_rate / 1e12, 3), "executed_frac": round(exec_rate / pk, 4), "executed_frac_nominal": round(exec_rate / PEAK_MAD_NOMINAL, 4),
         "frac_vs_reference_algorithm": round(alg_rate / pk, 4),
         "traffic": traffic, "traffic_measured_in_this_run": False, "traffic_replayed_from_profiles": traffic is not None, "traffic_source": traffic_source}
    if clock is not None:
        simd_cycles = dom_ms * 1e-3 * clock["ghz"] * 1e9
        wave_mads_per_simd = executed * items_per_launch / 64.0 / (cus * 4)
        r["kernel_clock_ghz"] = round(clock["ghz"], 4)
        r["kernel_clock_source"] = f"this run: wave stamps (s_memtime / s_memrealtime) of {clock['waves_stamped']} wavefronts in {clock['steps']} extra untimed steps"
        r["issue_share"] = round(wave_mads_per_simd * peak["simd_cycles_per_mad"] / simd_cycles, 4)
        r["issue_share_at_4_cycles"] = round(wave_mads_per_simd * 4.0 / simd_cycles, 4)
    step_rate = products[wl] * n / (avg_step_ms * 1e-3)
    r["step"] = {"avg_step_ms": round(avg_step_ms, 4), "kernels_ms": {k_: round(sum(v_) / steps, 4) for k_, v_ in per_kernel.items()},
                 "algorithmic_mads_per_item": products[wl], "frac_vs_reference_algorithm": round(step_rate / pk, 4)}
    r["hbm"] = {"achieved": round(ALG_BYTES[wl] * n / (avg_step_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ALG_BYTES[wl] * n / (avg_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "algorithmic_bytes_per_item": ALG_BYTES[wl]}
    return r


def device_identity(torch, local):
    p = torch.cuda.get_device_properties(local)
    ident = {"index": local, "name": p.name, "cus": p.multi_processor_count}
    for key in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id", "gcnArchName"):
        if hasattr(p, key):
            ident[key] = str(getattr(p, key))
    return ident


def verify_points_rate(w, eng, rt, steps, warmup):
    """the verify workload's signatures checked through kyb_verify_points_batch (keys handed over as points): K steps after W warm-up steps,
    statuses compared with the byte form's"""
    torch = rt.torch
    w["step"](); rt.sync()
    ref = w["out"].clone()
    for _ in range(warmup):
        w["step_points"]()
    rt.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        w["step_points"]()
    rt.sync()
    dt = time.perf_counter() - t0
    if not torch.equal(w["out"], ref):
        raise SystemExit("PARITY FAILURE (verify, keys as points): statuses differ from the byte form")
    return {"value": round(w["n"] * steps / dt, 1), "unit": UNIT["verify"], "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
            "entry_point": "kyb_verify_points_batch_dev (schnorr::verify takes &Point, schnorr_sig.rs:114-127)", "statuses_equal_to_byte_form": True}


def mid_size_calls(eng, orc, threads, n=8192):
    """A DKG-round-sized call (n = 8,192: more SIMDs than wavefronts): ONE synchronous host-pointer call per operation, median wall time in
    milliseconds, and the same variable-base call with the two-lane ladder switched off (k_mul_ladder_pair).  Every output of
    mul / mul_base, and the verification statuses, are checked against the oracle.  Not a throughput figure."""
    import numpy as np
    import synth as _s
    s = _s.scalars(n, 81)
    k = _s.scalars(n, 82, b"k")
    enc, ext = eng.mul_base(s, want_ext=True)
    msg_list = _s.messages(n, 83)
    import kyber_rs_amd as _k
    msgs = _k.pack_messages(msg_list)                  # blob + offsets once: the timed calls below contain no Python loop over the messages
    sigs = eng.schnorr_sign(s, k, msgs)
    sigs[::7, 33] ^= 1
    if not (np.array_equal(enc, orc.mul_base_batch(s, nthreads=threads)) and np.array_equal(eng.mul(k, pts_ext=ext), orc.mul_batch(k, ext, nthreads=threads))
            and np.array_equal(eng.verify(enc, msgs, sigs, 1), orc.verify_batch(1, enc, msg_list, sigs, nthreads=threads))):
        raise SystemExit("PARITY FAILURE (mid-size calls): GPU output differs from the oracle")

    def med(fn, reps=15):
        fn(); fn()
        ts = []
        for _ in range(reps):
            a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
        return round(sorted(ts)[len(ts) // 2] * 1e3, 4)

    out = {"unit": "ms per host-pointer call (median of 15)", "items": n, "checked_against_oracle": True,
           "mul_base": med(lambda: eng.mul_base(s)), "mul": med(lambda: eng.mul(k, pts_ext=ext)), "mul_enc": med(lambda: eng.mul(k, pts_enc=enc)),
           "sign": med(lambda: eng.schnorr_sign(s, k, msgs)), "verify": med(lambda: eng.verify(enc, msgs, sigs, 1))}
    # the same calls on arrays in page-locked memory (kyb_host_alloc): the kernels read and write them where they lie (host.in_place)
    P, ck, lib = _k._ptr, _k._check, eng.lib

    def pin(a):
        b = eng.pinned_array(a.shape, a.dtype)
        b[...] = a
        return b

    pk, penc, pext, psig, pblob, poff = pin(k), pin(enc), pin(ext), pin(sigs), pin(msgs.blob), pin(msgs.off)
    pout, pst = eng.pinned_array((n, 32), np.uint8), eng.pinned_array((n,), np.uint8)
    want_mul, want_st = eng.mul(k, pts_enc=enc), eng.verify(enc, msgs, sigs, 1)
    out["page_locked_arrays"] = {
        "mul": med(lambda: ck(lib.kyb_mul_batch(P(pk), None, P(pext), n, P(pout), None, None), "kyb_mul_batch")),
        "mul_enc": med(lambda: ck(lib.kyb_mul_batch(P(pk), P(penc), None, n, P(pout), None, None), "kyb_mul_batch")),
        "verify": med(lambda: ck(lib.kyb_verify_batch(P(penc), P(pblob), P(poff), P(psig), n, 1, P(pst)), "kyb_verify_batch"))}
    if not (np.array_equal(pout, want_mul) and np.array_equal(pst, want_st)):
        raise SystemExit("PARITY FAILURE (mid-size calls on page-locked arrays): outputs differ from the pageable calls'")
    pair = eng.get_option("ladder.pair_max_items")
    eng.set_option("ladder.pair_max_items", 0)
    try:
        out["mul_one_lane_per_item"] = med(lambda: eng.mul(k, pts_ext=ext))
        out["verify_one_lane_per_item"] = med(lambda: eng.verify(enc, msgs, sigs, 1))
    finally:
        eng.set_option("ladder.pair_max_items", pair)
    return out


def table_digest(eng):
    """what this rank's engine holds as the base-point table after the broadcast: every rank must report the same"""
    return hashlib.sha256(eng.base_table().tobytes()).hexdigest()[:16]


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

    # x = index + 1 of PubPoly::eval (poly.rs:461-464): multipliers the caller declares public (kyb_mul_public_batch) start their ladder below the leading zeros
    idx = np.zeros((64, 32), dtype=np.uint8)
    idx[:, 0] = np.arange(1, 65, dtype=np.uint8); idx[:, 1] = 2
    assert np.array_equal(eng.mul(idx, pts_ext=ext, public=True), orc.mul_batch(idx, ext))
    out = {"unit": "us per host-pointer call (median of 100)", "checked_against_oracle": True}
    for n in (1, 64):
        out[f"n={n}"] = {"mul_base": med(lambda: eng.mul_base(s[:n])), "mul": med(lambda: eng.mul(k[:n], pts_ext=ext[:n])),
                         "mul_public_10bit_index": med(lambda: eng.mul(idx[:n], pts_ext=ext[:n], public=True)),
                         "sign": med(lambda: eng.schnorr_sign(s[:n], k[:n], msgs[:n])), "verify": med(lambda: eng.verify(enc[:n], msgs[:n], sigs[:n], 1)),
                         "decode": med(lambda: eng.decode(enc[:n])), "encode": med(lambda: eng.encode(ext[:n]))}
    # the same single operations on the CPU port (one oracle call each through ctypes): what a batch-of-1 engine call competes with
    s0, k0, e0, x0, m0 = s[0].tobytes(), k[0].tobytes(), enc[0].tobytes(), ext[0], msgs[:1]
    out["cpu_port_n=1"] = {"mul_base": med(lambda: orc.mul_base(s0)), "mul": med(lambda: orc.mul(k0, x0)), "decode": med(lambda: orc.decode(e0)), "encode": med(lambda: orc.encode(x0)),
                           "sign": med(lambda: orc.schnorr_sign_batch(s[:1], k[:1], m0)), "verify": med(lambda: orc.verify_batch(1, enc[:1], m0, sigs[:1]))}
    out["call_by_call_t43"] = call_by_call_sequences(eng, orc, med)
    return out


def call_by_call_sequences(eng, orc, med, t=43):
    """The sequences unmodified protocol code runs, element at a time (share/poly.rs:195-206 commit, :457-469 PubPoly::eval, vss.rs:904-909 verify_deal),
    EAGER = every trait call one batch-of-1 engine call, against DEFERRED = the same calls recorded in the engine's arena (kyb_defer_*, csrc/defer.inc)
    and evaluated when the bytes are asked for.  Microseconds per whole sequence, median of 20; outputs of both forms checked against the oracle.
    (The full Pedersen dealer round, n = 64, t = 43, in C++: tests/cpp/test_vss_round.cpp, profiles/r04/vss_round.log.)"""
    import numpy as np
    import synth as _s
    coeffs = _s.scalars(t, 81)
    one = (1).to_bytes(32, "little")
    base_ext = eng.mul_base(np.frombuffer(one, dtype=np.uint8).reshape(1, 32), ext_only=True)[0]
    commits_ext = orc.mul_base_ext_batch(coeffs)
    index = 6
    xi = np.frombuffer((index + 1).to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32)
    share = np.frombuffer(orc.pripoly_eval(coeffs, index), dtype=np.uint8).reshape(1, 32)
    null_ext = np.zeros(40, dtype=np.int32); null_ext[10] = 1; null_ext[20] = 1
    want_commits = orc.mul_base_batch(coeffs)
    want_eval = orc.pubpoly_eval(commits_ext, index)

    def commit_eager():       # t x mul(coeff, Some(base)), then marshal_binary of each (session_id hashes them)
        pts = [eng.mul(coeffs[j:j + 1], pts_ext=base_ext[None], ext_only=True)[0] for j in range(t)]
        return [eng.encode(p[None])[0].tobytes() for p in pts]

    def commit_deferred():
        b = eng.defer_base()
        hs = [eng.defer_mul(coeffs[j].tobytes(), b) for j in range(t)]
        return [eng.defer_get(h) for h in hs]

    def eval_eager():         # v = null; t x { v = mul(xi, Some(v)); v = add(v, commits[j]) }
        v = null_ext
        for j in reversed(range(t)):
            v = eng.mul(xi, pts_ext=v[None], ext_only=True)[0]
            v = eng.add(v[None], commits_ext[j][None])[0]
        return v

    def verify_deal_eager():  # fig = base().mul(fi.v, None); pub_share = eval(fi.i); fig == pub_share.v
        fig = eng.mul_base(share, ext_only=True)[0]
        return bool(eng.equal(fig[None], eval_eager()[None])[0])

    hc = [eng.defer_input(c) for c in commits_ext]

    def eval_deferred_handle():
        v = eng.defer_input(null_ext)
        x = xi.tobytes()
        for j in reversed(range(t)):
            v = eng.defer_mul(x, v)
            v = eng.defer_add(v, hc[j])
        return v

    def verify_deal_deferred():
        fig = eng.defer_mul_base(share.tobytes())
        return eng.defer_equal(fig, eval_deferred_handle())

    # dist_key_share (dkg.rs:905-953): pubb = pubb.add(poly) for 8 dealers' commitment polynomials, t one-at-a-time Point::add per fold (poly.rs:486-507)
    dealers = 8
    polys = [orc.mul_base_ext_batch(_s.scalars(t, 90 + d)) for d in range(dealers)]
    want_fold = []
    for j in range(t):
        acc = polys[0][j]
        for d in range(1, dealers):
            acc = orc.add(acc, polys[d][j])
        want_fold.append(orc.encode(acc))

    def fold_eager():
        pubb = [polys[0][j] for j in range(t)]
        for d in range(1, dealers):
            pubb = [eng.add(pubb[j][None], polys[d][j][None])[0] for j in range(t)]
        return [eng.encode(p[None])[0].tobytes() for p in pubb]

    def fold_deferred():
        pubb = [eng.defer_input(polys[0][j]) for j in range(t)]
        for d in range(1, dealers):
            pubb = [eng.defer_add(pubb[j], eng.defer_input(polys[d][j])) for j in range(t)]
        return [eng.defer_get(h) for h in pubb]

    # the same four sequences call by call on the CPU port (the oracle through ctypes, one thread): the column the engine's two are read against
    base_cpu = orc.base()

    def commit_cpu():
        return [orc.encode(orc.mul_ext(coeffs[j].tobytes(), base_cpu)) for j in range(t)]

    def eval_cpu():
        v = orc.null()
        for j in reversed(range(t)):
            v = orc.add(orc.mul_ext(xi.tobytes(), v), commits_ext[j])
        return v

    def verify_deal_cpu():
        return orc.encode(orc.mul_base_ext(share.tobytes())) == orc.encode(eval_cpu())

    def fold_cpu():
        pubb = [polys[0][j] for j in range(t)]
        for d in range(1, dealers):
            pubb = [orc.add(pubb[j], polys[d][j]) for j in range(t)]
        return [orc.encode(p) for p in pubb]

    assert fold_cpu() == want_fold and commit_cpu() == [bytes(w) for w in want_commits] and orc.encode(eval_cpu()) == want_eval and verify_deal_cpu()
    assert fold_eager() == want_fold == fold_deferred()
    assert commit_eager() == [bytes(w) for w in want_commits] == commit_deferred()
    assert orc.encode(eval_eager()) == want_eval == eng.defer_get(eval_deferred_handle())
    assert verify_deal_eager() and verify_deal_deferred()
    mark = eng.defer_mark()
    res = {"unit": "us per whole sequence (median of 20), t = 43", "checked_against_oracle": True,
           "cpu_port": "the oracle (C restatement of the reference algorithm) call by call through ctypes, one thread, same box",
           "commit_then_marshal": {"cpu_port": med(commit_cpu, 10), "eager": med(commit_eager, 20), "deferred": med(commit_deferred, 20)},
           "pubpoly_eval": {"cpu_port": med(eval_cpu, 10), "eager": med(eval_eager, 20), "deferred": med(lambda: eng.defer_get_ext(eval_deferred_handle()), 20)},
           "verify_deal": {"cpu_port": med(verify_deal_cpu, 10), "eager": med(verify_deal_eager, 20), "deferred": med(verify_deal_deferred, 20)},
           "pubpoly_add_fold_8_dealers": {"cpu_port": med(fold_cpu, 10), "eager": med(fold_eager, 10), "deferred": med(fold_deferred, 10)}}
    res["arena"] = eng.defer_stats()
    eng.defer_floor(mark)
    return res


def protocol_phases(n=64, t=43):
    """Unmodified call-by-call protocol code, phase by phase, with the CPU beside it (round-4 review item 3): tests/cpp/test_vss_round.cpp (a Pedersen dealer
    round: vss.rs:287-337, 361-386, 904-909), tests/cpp/test_dkg_finish.cpp (dkg.rs:905-953, poly.rs:566-603) and tests/cpp/test_dss_round.cpp (a DSS
    signing round at one participant: dss_sig.rs:173-326) are compiled twice — against the engine
    (eager = every trait call a batch-of-1 engine call; deferred = calls recorded and evaluated in batches; batched = the same round written with the batch
    entry points) and against tests/cpp/cpu_port_abi.cpp, the oracle behind the same ABI: the IDENTICAL sequence on one host core (cpu_port_ms).  All four
    transcripts of a program are byte-identical (checked here; against the oracle item by item in tests/test_gpu_vss_round.py).  Milliseconds, one run each
    after one untimed pass.  Phases where deferred loses to the CPU are sequences of one-item requests, each waited for before the next is made
    (encrypted_deals: key, signature, shared point per verifier; new_dealer: n marshals) — the price is the GPU's one-item latency, see DESIGN.md."""
    import importlib
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        rounds = importlib.import_module("test_gpu_vss_round")
        out = {"n": n, "t": t, "unit": "ms per phase", "cpu_port": "the C restatement of the reference algorithm (oracle/) behind the same ABI, one thread"}
        out["repeats"] = "best of 3 runs per program and phase (host-side latencies: a busy host core shows at once)"
        for prog in ("test_vss_round", "test_dkg_finish", "test_dss_round"):
            gpu_bin, cpu_bin = rounds.build(prog), rounds.build(prog, cpu_port=True)
            lines, timing = rounds.run_program(gpu_bin, n, t)
            cpu_lines, cpu = rounds.run_program(cpu_bin, n, t, "eager")
            for _ in range(2):
                _l, t2 = rounds.run_program(gpu_bin, n, t)
                _c, c2 = rounds.run_program(cpu_bin, n, t, "eager")
                for form in ("eager_ms", "deferred_ms", "batched_ms"):
                    if form in timing:
                        timing[form] = {ph: min(ms, t2[form][ph]) for ph, ms in timing[form].items()}
                cpu["eager_ms"] = {ph: min(ms, c2["eager_ms"][ph]) for ph, ms in cpu["eager_ms"].items()}
            same = lines["E"] == lines["D"] == cpu_lines["E"] and (not lines["B"] or lines["B"] == lines["E"])
            phases = {}
            for ph, ms in cpu["eager_ms"].items():
                # default_ms: what a caller gets who never chooses a mode — the program reports which mode its binding starts in
                # (host/edwards25519.hpp: deferred unless KYBER_HIP_EAGER is set; the Rust module the same)
                dflt = timing.get("default_mode", "eager")
                phases[ph] = {"cpu_port_ms": ms, "eager_ms": timing["eager_ms"][ph], "deferred_ms": timing["deferred_ms"][ph], "default_ms": timing[dflt + "_ms"][ph]}
                if "batched_ms" in timing:
                    phases[ph]["batched_ms"] = timing["batched_ms"][ph]
                phases[ph]["deferred_vs_cpu"] = round(ms / timing["deferred_ms"][ph], 2)
            out[prog] = {"phases": phases, "transcripts_identical": same, "deferred_stats": timing["deferred_stats"], "default_mode": timing.get("default_mode", "eager")}
        return out
    except Exception as e:      # a missing g++ on the bench box must not cost the bench line
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def host_pointer_rates(w, eng, orc, threads, calls=5):
    """kyb_mul_batch on the same 2^20 items in HOST memory (what a Rust caller holds): page-locked buffers (kyb_host_alloc) and ordinary
    pageable numpy arrays; PCIe-inclusive, synchronous calls, 256 outputs of each against the oracle.  Never `value`."""
    import numpy as np
    n = w["n"]
    sc, pts = w["sc_np"], w["pts"].cpu().numpy()
    idx = np.sort(np.random.default_rng(5).choice(n, 256, replace=False))
    want = orc.mul_batch(sc[idx], pts[idx], nthreads=threads)
    ps = eng.pinned_array((n, 32), np.uint8); ps[:] = sc
    pe = eng.pinned_array((n, 40), np.int32); pe[:] = pts
    po = eng.pinned_array((n, 32), np.uint8)
    outp = np.empty((n, 32), dtype=np.uint8)
    res = {}
    for name, fn, out in (("host_pinned", lambda: eng.mul_into(ps, pe, po), po), ("host_pageable", lambda: eng.mul_into(sc, pts, outp), outp)):
        fn(); fn()                                       # the first calls size the staging buffers and wake the copy threads
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[len(ts) // 2]                    # median call
        if not np.array_equal(out[idx], want):
            raise SystemExit(f"PARITY FAILURE ({name}): host-pointer output differs from the oracle")
        res[name] = {"value": round(n / dt, 1), "unit": UNIT["mul"], "ms_per_call": round(dt * 1e3, 3), "calls": calls, "statistic": "median call after two warm-up calls", "items": n,
                     "bytes_over_pcie_per_item": 224, "pcie_gb_s": round(224 * n / dt / 1e9, 2), "parity_checked_items": 256}
    return res


# ------------------------------------------------------------------------------------------------------------------------------
CFG5_TOTAL = 1 << 24      # BASELINE.json configs[4]


def shard_bounds(total, world, rank):
    """rank r's slice [floor(r total / world), floor((r + 1) total / world)) of a job of `total` independent items"""
    return rank * total // world, (rank + 1) * total // world


def plan_items(args, wl, world, rank):
    """-> (items of this rank, total items of the job, scaling, workload text).  N = 1: the single-GPU configuration (or --n).  N > 1: the
    variable-base workloads run configs[4] — 2^24 items cut into `world` shards — unless --scaling weak or --n say otherwise."""
    if args.n:
        return args.n, args.n * world, "weak", f"{wl} x {args.n} per GPU"
    text = WORKLOAD_TEXT[wl] + (", signers hold their public keys (one fixed-base mult per signature)" if wl == "sign" and args.keyed else "")
    if world > 1 and wl in ("mul", "mul_enc") and args.scaling != "weak":
        job = args.total or CFG5_TOTAL
        lo, hi = shard_bounds(job, world, rank)
        form = "reference-limb points in" if wl == "mul" else "32-byte wire encodings in (decoded on the GPU inside the step)"
        size = "2^24" if job == CFG5_TOTAL else str(job)
        return hi - lo, job, "strong", (f"{size} variable-base scalar-mults sharded across {world}xMI355X ({job // world} per GPU), random scalars+points, "
                                        f"{form}, 32-byte encodings out, table image broadcast from rank 0 at init")
    return DEFAULT_N[wl], DEFAULT_N[wl] * world, "weak", text


def common_line(args, wl, n, world, elapsed, eng, total=None, scaling="weak", text=None):
    total = n * world if total is None else total
    return {"metric": METRIC[wl], "value": round(total * args.steps / elapsed, 1), "unit": UNIT[wl], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "u32 limbs (radix 2^25.5), u64 accumulators", "data": "synthetic",
            "config": {"workload": text if text is not None else plan_items(args, wl, world, 0)[3],
                       "items_per_gpu": n, "total_items": total, "mode": args.mode,
                       "sharding": f"independent shards x{world}, no data-path collective; one table broadcast at init",
                       "options": {k_: eng.get_option(k_) for k_ in OPTION_KEYS} if eng is not None else None}}


def run_ranks(args):
    """one process = one rank = one GPU (world 1: the plain single-GPU run)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.standin and os.environ.get("KYB_BENCH_FAIL_RANK") == str(rank):      # tests/test_multi_gpu_cpu.py: a rank that dies before the rendezvous
        raise SystemExit(f"rank {rank}: asked to fail (KYB_BENCH_FAIL_RANK)")
    import torch
    import torch.distributed as dist
    if args.same_device:                 # tests: every rank on GPU 0 (the one-GPU box; with --dist-backend gloo, RCCL wants a device per rank)
        local = 0
    rt = StandinRuntime(local) if args.standin else GpuRuntime(local)
    dev = rt.dev
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.standin or args.dist_backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        backend = dist.get_backend()

    # ---- engine + base-point table (RCCL broadcast of rank 0's image over xGMI) ----
    from kyber_rs_amd import multi_gpu
    if args.standin:
        import standin_engine                        # tests/standin_engine.py: records calls, computes nothing
        eng = standin_engine.StandinEngine(build_table=(rank == 0))
    else:
        import kyber_rs_amd
        eng = kyber_rs_amd.Engine(local, build_table=(rank == 0))       # raises without the HIP library or a gfx950 device
    multi_gpu.distribute_base_table(eng, rank, world, dev, dist)
    for kv in args.opt:
        key, val = kv.split("=")
        eng.set_option(key, int(val))

    # what the job really ran on: every rank reports its device and the table it holds, rank 0 gathers
    ident = {"index": local, "name": "standin (no GPU)"} if args.standin else device_identity(torch, local)
    ident["rank"] = rank
    ident["table_sha256_16"] = table_digest(eng)
    idents = [ident]
    if world > 1:
        idents = [None] * world
        dist.all_gather_object(idents, ident)

    wl = args.workload
    n, total, scaling, text = plan_items(args, wl, world, rank)
    inp = Inputs(1 + rank, dev)          # every rank gets its own shard of the synthetic stream

    def barrier():
        if world > 1:
            dist.barrier()
        rt.sync()

    t0 = time.time()
    w = setup_workload(wl, n, eng, inp, rt, args.keyed)
    gen_s = time.time() - t0
    elapsed = time_workload(w, eng, rt, args.steps, args.warmup, barrier)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- parity: EVERY rank checks a sample of its own shard against the oracle; the verdicts are gathered before anything is printed ----
    parity = {"rank": rank, "ok": True, "checked": 0, "error": None}
    orc = None
    # the GPU box gives one GPU a share of 16 host CPUs although os.cpu_count() reports the whole machine
    threads = max(1, min(len(os.sched_getaffinity(0)), 16 * world) // world)
    if args.standin:
        parity["checked"] = min(args.check, n)           # the stand-in computes nothing: the plumbing of the verdict is what is exercised
        if os.environ.get("KYB_BENCH_PARITY_FAIL_RANK") == str(rank):
            parity.update(ok=False, error=f"rank {rank}: asked to report a parity failure (KYB_BENCH_PARITY_FAIL_RANK)")
    else:
        import oracle_lib
        orc = oracle_lib.Oracle()
        try:
            parity["checked"] = check_parity(w, orc, args.check, threads, dev)
        except SystemExit as e:
            parity.update(ok=False, error=f"rank {rank}: {e}")
    parities = [parity]
    if world > 1:
        parities = [None] * world
        dist.all_gather_object(parities, parity)
    failed = [p_ for p_ in parities if not p_["ok"]]
    if failed:                                            # every rank leaves with a failure status; nothing that looks like a result is printed
        if rank == 0:
            print("PARITY FAILURE: " + "; ".join(p_["error"] for p_ in failed), file=sys.stderr, flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        raise SystemExit(3)

    if rank == 0:
        if len({d["table_sha256_16"] for d in idents}) != 1:
            raise SystemExit(f"TABLE MISMATCH: the ranks hold different base-point table images: {[d['table_sha256_16'] for d in idents]}")
        line = common_line(args, wl, n, world, elapsed, eng, total, scaling, text)
        line["parity_checked_items"] = sum(p_["checked"] for p_ in parities)
        line["parity_checked_items_per_rank"] = [p_["checked"] for p_ in sorted(parities, key=lambda p_: p_["rank"])]
        line["items_per_rank"] = [shard_bounds(total, world, r_)[1] - shard_bounds(total, world, r_)[0] for r_ in range(world)] if scaling == "strong" else [n] * world
        line.update({"input_gen_s": round(gen_s, 2), "ranks_seen": len(idents), "dist_backend": backend, "devices": idents,
                     "table_identical_on_all_ranks": len({d["table_sha256_16"] for d in idents}) == 1,
                     "launched_by": "self-spawned ranks" if os.environ.get("KYB_BENCH_SPAWNED") else ("external launcher" if world > 1 else "single process")})
        if args.standin:
            line.update({"metric": "STANDIN plumbing run (no GPU, no arithmetic): NOT a measurement", "data": "none (stand-in engine)", "roofline": None, "cpu_baseline": None,
                         "standin_calls": eng.calls})
        else:
            cus = eng.device_info()["compute_units"]
            clock = kernel_clock(w, eng, rt)
            peak = eng.mad_peak(50.0)                     # the roofline's denominator, measured on this chip in this run (warm from the steps above)
            line["roofline"] = roofline(w, eng, args.steps, peak, clock, cus)
            line["cpu_baseline"] = cpu_baseline(w, orc, threads) if (world == 1 and not args.no_cpu_baseline) else None
            # ---- the other single-GPU configurations, each in its own timed region outside the primary one ----
            if world == 1 and not args.only and not args.n:
                if wl == "mul":
                    line.update(host_pointer_rates(w, eng, orc, threads))         # host_pinned / host_pageable
                others = {}
                del w
                for owl in ("mul_enc", "mul_base", "sign", "verify", "mul"):
                    if owl == wl:
                        continue
                    on = DEFAULT_N[owl]
                    ow = setup_workload(owl, on, eng, inp, rt, False)
                    oel = time_workload(ow, eng, rt, args.steps, args.warmup, barrier)
                    ochk = check_parity(ow, orc, args.check, threads, dev)
                    oclock = kernel_clock(ow, eng, rt)
                    others[owl] = {"metric": METRIC[owl], "value": round(on * args.steps / oel, 1), "unit": UNIT[owl],
                                   "ms_per_step": round(oel / args.steps * 1e3, 4), "steps": args.steps, "warmup": args.warmup,
                                   "config": {"workload": WORKLOAD_TEXT[owl], "items_per_gpu": on},
                                   "roofline": roofline(ow, eng, args.steps, peak, oclock, cus),
                                   "cpu_baseline": None if args.no_cpu_baseline else cpu_baseline(ow, orc, threads),
                                   "parity_checked_items": ochk}
                    if owl == "mul":
                        others[owl].update(host_pointer_rates(ow, eng, orc, threads))
                    if owl == "verify":
                        others[owl]["public_keys_as_points"] = verify_points_rate(ow, eng, rt, args.steps, args.warmup)
                    del ow
                line["workloads"] = others
                line["small_calls"] = small_call_latency(eng, orc)
                line["mid_size_calls"] = mid_size_calls(eng, orc, threads)
                line["protocol_phases"] = protocol_phases()
                peak2 = eng.mad_peak(50.0)
                line["roofline"]["peak_repeat_at_end_of_run"] = {"peak": round(peak2["mads_per_s"] / 1e12, 3), "clock_ghz": round(peak2["clock_ghz"], 4)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_group(args):
    """ONE process, all GPUs: kyb_group_create (one context per GPU, the table image moved by the library's own ncclBroadcast),
    device-resident shards, every step queued on all GPUs by this thread through kyb_group_mul(_base)_batch_dev"""
    import numpy as np
    import torch
    import kyber_rs_amd
    import oracle_lib
    world = args.gpus
    wl = args.workload
    if wl not in ("mul", "mul_enc", "mul_base"):
        raise SystemExit("--mode group drives the scalar-multiplication workloads (mul, mul_enc, mul_base)")
    n, total, scaling, text = plan_items(args, wl, world, 0)
    if scaling == "strong":                      # one shard size for every context of the group: floor(total / world) items each
        total = n * world
    have = torch.cuda.device_count()
    if have < world:
        raise SystemExit(f"--gpus {world}: this host shows {have} GPU(s)")
    # distinct devices: the native RCCL broadcast is REQUIRED — on the first multi-GPU node a broken transport must be an error, not a quiet
    # "host-copy" in the line (KYB_GROUP_REQUIRE_RCCL; --allow-host-copy lifts it)
    grp = kyber_rs_amd.Group(list(range(world)), flags=0 if (world == 1 or args.allow_host_copy) else kyber_rs_amd.Group.REQUIRE_RCCL)
    engs = [grp.engine(r) for r in range(world)]
    for e in engs:
        for kv in args.opt:
            key, val = kv.split("=")
            e.set_option(key, int(val))
    idents = []
    for r in range(world):
        ident = device_identity(torch, r)
        ident.update({"rank": r, "table_sha256_16": table_digest(engs[r])})
        idents.append(ident)
    # shards: rank r's inputs live on GPU r (seed 1 + r, as in ranks mode)
    t0 = time.time()
    sc, pts, penc, ok, out, sc_np = [], [], [], [], [], []
    for r in range(world):
        dev = torch.device("cuda", r)
        inp = Inputs(1 + r, dev)
        sc_np.append(inp.scalars_np(n))
        sc.append(inp.scalars(n))
        out.append(torch.empty((n, 32), dtype=torch.uint8, device=dev))
        if wl != "mul_base":
            psc = inp.scalars(n, b"point")
            pts.append(torch.empty((n, 40), dtype=torch.int32, device=dev))
            penc.append(torch.empty((n, 32), dtype=torch.uint8, device=dev))
            ok.append(torch.empty((n,), dtype=torch.uint8, device=dev))
            torch.cuda.synchronize(r)
            engs[r].mul_base_dev(psc, out_enc=penc[r], out_ext=pts[r])
    grp.sync()
    gen_s = time.time() - t0

    def step():
        if wl == "mul":
            grp.mul_dev(sc, pts_ext=pts, out_enc=out)
        elif wl == "mul_enc":
            grp.mul_dev(sc, pts_enc=penc, out_enc=out, ok=ok)
        else:
            grp.mul_base_dev(sc, out_enc=out)

    def sync_all():
        grp.sync()
        for r in range(world):
            torch.cuda.synchronize(r)

    for _ in range(args.warmup):
        step()
    sync_all()
    engs[0].profile_begin(10 * args.steps)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t_start
    launches = engs[0].profile_read(10 * args.steps)
    engs[0].profile_begin(0)

    orc = oracle_lib.Oracle()
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))
    checked = 0
    m = max(1, args.check // world) if args.check else 0
    for r in range(world):                       # every rank's outputs against the oracle
        if not m:
            break
        idx = np.sort(np.random.default_rng(r).choice(n, min(m, n), replace=False))
        tidx = torch.from_numpy(idx).to(out[r].device)
        got = out[r][tidx].cpu().numpy()
        want = orc.mul_base_batch(sc_np[r][idx], nthreads=threads) if wl == "mul_base" else orc.mul_batch(sc_np[r][idx], pts[r][tidx].cpu().numpy(), nthreads=threads)
        if not np.array_equal(got, want):
            raise SystemExit(f"PARITY FAILURE (group rank {r}, {wl}): GPU output differs from the oracle")
        checked += len(idx)
    if len({d["table_sha256_16"] for d in idents}) != 1:
        raise SystemExit(f"TABLE MISMATCH: the group's contexts hold different base-point table images: {[d['table_sha256_16'] for d in idents]}")
    line = common_line(args, wl, n, world, elapsed, engs[0], total, scaling, text)
    per_kernel = {}
    for name, ms in launches:
        per_kernel.setdefault(name, []).append(ms)
    line.update({"input_gen_s": round(gen_s, 2), "ranks_seen": world, "dist_backend": None, "table_transport": grp.transport, "table_transport_note": grp.transport_note or None, "devices": idents,
                 "table_identical_on_all_ranks": len({d["table_sha256_16"] for d in idents}) == 1,
                 "launched_by": "one process, kyb_group (one context and stream per GPU, launches queued by one thread)",
                 "rank0_kernels_ms_per_step": {k_: round(sum(v_) / args.steps, 4) for k_, v_ in per_kernel.items()},
                 "roofline": None, "cpu_baseline": None, "parity_checked_items": checked})
    print(json.dumps(line), flush=True)
    grp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", default="ranks", choices=["ranks", "group"], help="N > 1: one process per GPU over torch.distributed/RCCL (default), "
                    "or one process driving all GPUs through kyb_group (native ncclBroadcast of the table image)")
    ap.add_argument("--workload", default="mul", choices=["mul", "mul_enc", "mul_base", "sign", "verify"])
    ap.add_argument("--keyed", action="store_true", help="sign: the signers hold their public keys (EdDSA objects, DSS long-term keys): "
                    "one fixed-base mult per signature instead of the two of schnorr::sign")
    ap.add_argument("--n", type=int, default=0, help="items per GPU (default: 2^20, 2^18 for sign; with N > 1 GPUs the variable-base workloads default to 2^24 / N)")
    ap.add_argument("--allow-host-copy", action="store_true", help="--mode group: accept the host-copy fallback for the table image when RCCL is unavailable")
    ap.add_argument("--total", type=int, default=0, help="N > 1: items of the whole sharded job (default 2^24 = BASELINE configs[4])")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak"], help="N > 1: auto = BASELINE configs[4], 2^24 items in all (strong scaling); "
                    "weak = the single-GPU batch on every GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only", action="store_true", help="time the primary workload only (no `workloads` object)")
    ap.add_argument("--check", type=int, default=16384, help="items verified against the oracle after timing (per workload)")
    ap.add_argument("--opt", action="append", default=[], help="engine option key=value (kernel variant), repeatable")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help=argparse.SUPPRESS)     # tests: real engines, gloo instead of RCCL
    ap.add_argument("--same-device", action="store_true", help=argparse.SUPPRESS)                           # tests: every rank on GPU 0
    ap.add_argument("--standin", action="store_true", help=argparse.SUPPRESS)      # tests only: CPU stand-in engine over gloo, prints a line marked as no measurement
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    # concurrent small calls want more hardware queues than the runtime's default of 4; the variable is read when the HIP runtime starts
    # and belongs to the host program (the library never sets it): here, before anything initialises HIP
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

    if args.mode == "group":
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--mode group is one process for all GPUs: start it without a launcher")
        return run_group(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        # started plainly: become the launcher.  Nothing in this process has touched the GPU (torch is not even imported).
        sys.exit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    run_ranks(args)


if __name__ == "__main__":
    main()
