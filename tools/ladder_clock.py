#!/usr/bin/env python3
"""In-kernel clock of the ladder kernel (MI355X_MICROARCH.md 'DVFS give-back' item 6).

Runs a DIAGNOSTIC build of the library (-DKYB_DIAG_STAMPS: k_mul_ladder reads s_memtime / s_memrealtime right
before and after its 256-step loop and writes the two differences per wave to a buffer of their own; the product
library executes no stamp) back to back for >= 2.5 s on random data, then reports from the LAST launch:

  in_kernel_ghz              median over waves of  d(s_memtime) / d(s_memrealtime) x 100 MHz
  cycles_per_wave_step       median over waves of  d(s_memtime) / steps           (a wave shares its SIMD with others)
  simd_cycles_per_wave_step  kernel time (HIP events) x in-kernel clock / (wave-steps one SIMD executes)
  mad_issue_share_*          739 v_mad_u64_u32 per step x {4.0 nominal half-rate cycles, 4.63 measured in
                             tools/microbench/valu_rates.hip} / simd_cycles_per_wave_step

  python tools/ladder_clock.py [--n 1048576] [--seconds 2.5] [--out profiles/r02/ladder_clock.json]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG_LIB = os.path.join(ROOT, "tools", "_build", "libkyber_ed25519_hip_stamps.so")
MADS_PER_STEP = 5 * 100 + 4 * 55 + 10 + 9          # 5 M + 4 S + the a24 multiplication + 9 carry folds


def build_diag():
    sys.path.insert(0, ROOT)
    import __graft_entry__
    os.makedirs(os.path.dirname(DIAG_LIB), exist_ok=True)
    return __graft_entry__.build_hip(extra_flags=("-DKYB_DIAG_STAMPS",), out=DIAG_LIB, objdir="_obj_stamps")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02", "ladder_clock.json"))
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--no-build", action="store_true", help="use the diagnostic library as it is (it was built before the snapshot left for the GPU box)")
    args = ap.parse_args()
    if not (args.no_build and os.path.exists(DIAG_LIB)):
        build_diag()
    if args.build_only:
        return
    os.environ["KYB_HIP_LIB"] = DIAG_LIB
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import torch
    import kyber_rs_amd
    import synth

    eng = kyber_rs_amd.Engine(0)
    dev = torch.device("cuda", 0)
    n = args.n
    rng = np.random.default_rng(7)
    # random scalars below L's bit length and random points (s_i * B): same distribution as bench.py, cheaper to make
    sc_np = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    sc_np[:, 31] &= 0x0f
    sc = torch.from_numpy(sc_np).to(dev)
    psc = torch.from_numpy(synth.scalars(4096, 3, b"point")).to(dev).repeat((n + 4095) // 4096, 1)[:n].contiguous()
    pts = torch.empty((n, 40), dtype=torch.int32, device=dev)
    out = torch.empty((n, 32), dtype=torch.uint8, device=dev)
    tstream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(tstream)
    st = tstream.cuda_stream
    eng.mul_base_dev(psc, out_ext=pts, stream=st)
    waves = (n + 63) // 64
    stamps = torch.zeros((waves, 2), dtype=torch.int64, device=dev)
    lib = kyber_rs_amd.load_library()      # the raw ctypes library (the engine sees it through its context proxy)
    lib.kyb_diag_set_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.kyb_diag_set_stamp_buffer.restype = ctypes.c_int
    assert lib.kyb_diag_set_stamp_buffer(ctypes.c_void_p(stamps.data_ptr()), waves) == 0
    torch.cuda.synchronize()
    t0 = time.time()
    launches = 0
    while time.time() - t0 < args.seconds:
        for _ in range(16):
            eng.mul_dev(sc, pts_ext=pts, out_enc=out, stream=st)
        launches += 16
        torch.cuda.synchronize()
    eng.profile_begin(8)
    eng.mul_dev(sc, pts_ext=pts, out_enc=out, stream=st)
    torch.cuda.synchronize()
    prof = dict(eng.profile_read(8))
    eng.profile_begin(0)
    s = stamps.cpu().numpy().astype(np.float64)
    cyc, rt = s[:, 0], s[:, 1]
    ok = rt > 0
    ghz = np.median(cyc[ok] / rt[ok] * 0.1)
    steps = 256
    cu = eng.device_info()["compute_units"]
    simds = cu * 4
    ladder_ms = prof["k_mul_ladder"]
    wave_steps_per_simd = waves * steps / simds
    simd_cyc = ladder_ms * 1e-3 * ghz * 1e9 / wave_steps_per_simd
    res = {
        "what": "k_mul_ladder, diagnostic build with s_memtime/s_memrealtime stamps around the 256-step loop (tools/ladder_clock.py)",
        "items": n, "waves": waves, "back_to_back_launches_before_sample": launches, "seconds_of_load": round(time.time() - t0, 2),
        "in_kernel_ghz": round(float(ghz), 4),
        "in_kernel_ghz_p05_p95": [round(float(np.percentile(cyc[ok] / rt[ok] * 0.1, q)), 4) for q in (5, 95)],
        "wave_loop_cycles_median": float(np.median(cyc)), "wave_loop_cycles_min_max": [float(cyc.min()), float(cyc.max())],
        "cycles_per_wave_step_median": round(float(np.median(cyc)) / steps, 1),
        "k_mul_ladder_ms_this_launch_hip_events": round(ladder_ms, 4),
        "simd_cycles_per_wave_step": round(float(simd_cyc), 1),
        "mads_per_step": MADS_PER_STEP,
        "mad_issue_share_at_4.00_cyc": round(MADS_PER_STEP * 4.0 / simd_cyc, 4),
        "mad_issue_share_at_4.63_cyc": round(MADS_PER_STEP * 4.63 / simd_cyc, 4),
        "mad_rate_T_per_s": round(MADS_PER_STEP * steps * n / (ladder_ms * 1e-3) / 1e12, 3),
        "mad_peak_at_in_kernel_clock_T_per_s_4.00_cyc": round(simds * 64 / 4.0 * ghz * 1e9 / 1e12, 2),
        "note": "stamped build: the stamps themselves and the extra fe_copy cost < 0.1 % of the loop; never compare its wall time with the product build's",
    }
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
