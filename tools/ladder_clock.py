#!/usr/bin/env python3
"""In-kernel clock of the ladder kernel (MI355X_MICROARCH.md 'DVFS give-back').

The PRODUCT library carries optional wave stamps (csrc/diag_stamp.h, kyb_diag_wave_stamps): while a buffer is set every wavefront of
k_mul_ladder adds s_memtime / s_memrealtime at its start and end into five 64-bit sums.  This tool runs the 2^20-item step back to
back for >= 2.5 s, then reports from the LAST launch:

  in_kernel_ghz              (sum of cycle differences) / (sum of 100 MHz tick differences) x 100 MHz
  simd_cycles_per_wave_step  kernel time (HIP events) x in-kernel clock / (wave-steps one SIMD executes)
  mad_issue_share_*          739 v_mad_u64_u32 per step x {4.0 nominal quarter-rate cycles, the cycles kyb_diag_mad_peak measures in
                             this run} / simd_cycles_per_wave_step

  python tools/ladder_clock.py [--n 1048576] [--seconds 2.5] [--out profiles/r03/ladder_clock.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MADS_PER_STEP = 5 * 100 + 4 * 55 + 10 + 9          # 5 M + 4 S + the a24 multiplication + 9 carry folds


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03", "ladder_clock.json"))
    args = ap.parse_args()
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
    stamps = torch.zeros(8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    peak = eng.mad_peak(50.0)
    t0 = time.time()
    launches = 0
    while time.time() - t0 < args.seconds:
        for _ in range(16):
            eng.mul_dev(sc, pts_ext=pts, out_enc=out, stream=st)
        launches += 16
        torch.cuda.synchronize()
    eng.wave_stamps(stamps)
    eng.profile_begin(8)
    eng.mul_dev(sc, pts_ext=pts, out_enc=out, stream=st)
    torch.cuda.synchronize()
    prof = dict(eng.profile_read(8))
    eng.profile_begin(0)
    eng.wave_stamps(None)
    s = [int(v) & ((1 << 64) - 1) for v in stamps.cpu().tolist()]
    cyc, rt, seen = (s[2] - s[0]) % (1 << 64), (s[3] - s[1]) % (1 << 64), s[4]
    assert seen == waves, (seen, waves)
    ghz = cyc / rt * 0.1
    steps = 256 - (4 if eng.get_option("ladder.skip_canonical") else 0)      # the scalars above are below 2^252
    cu = eng.device_info()["compute_units"]
    simds = cu * 4
    ladder_ms = prof["k_mul_ladder"]
    wave_steps_per_simd = waves * steps / simds
    simd_cyc = ladder_ms * 1e-3 * ghz * 1e9 / wave_steps_per_simd
    res = {
        "what": "k_mul_ladder of the product library, wave stamps on (kyb_diag_wave_stamps: s_memtime / s_memrealtime at wavefront start and end; tools/ladder_clock.py)",
        "items": n, "waves": waves, "back_to_back_launches_before_sample": launches, "seconds_of_load": round(time.time() - t0, 2),
        "in_kernel_ghz": round(float(ghz), 4),
        "wave_lifetime_cycles_mean": round(cyc / waves, 1), "ladder_steps": steps,
        "cycles_per_wave_step_mean": round(cyc / waves / steps, 1),
        "k_mul_ladder_ms_this_launch_hip_events": round(ladder_ms, 4),
        "simd_cycles_per_wave_step": round(float(simd_cyc), 1),
        "mads_per_step": MADS_PER_STEP,
        "mad_issue_share_at_4.00_cyc": round(MADS_PER_STEP * 4.0 / simd_cyc, 4),
        "mad_peak_this_run": {k_: round(v_, 4) if k_ != "mads_per_s" else round(v_ / 1e12, 3) for k_, v_ in peak.items()},
        "mad_issue_share_at_measured_cyc": round(MADS_PER_STEP * peak["simd_cycles_per_mad"] / simd_cyc, 4),
        "mad_rate_T_per_s": round(MADS_PER_STEP * steps * n / (ladder_ms * 1e-3) / 1e12, 3),
        "mad_peak_at_in_kernel_clock_T_per_s_4.00_cyc": round(simds * 64 / 4.0 * ghz * 1e9 / 1e12, 2),
        "note": "the stamps span the whole wavefront (operand load, ladder, y-recovery, store): the ladder loop is 97 % of it",
    }
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
