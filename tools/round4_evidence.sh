#!/bin/bash
# Run ON THE GPU BOX (via gpurun): this round's evidence files other than the rocprofv3 summaries (tools/collect_profiles.sh) — see profiles/r04/README.md
set -u
out=gpurun_out/r04; mkdir -p $out
lib=kyber-rs_amd
mkdir -p tests/cpp/_build
g++ -O2 -std=c++17 -Wall -Wno-unused-function -o tests/cpp/_build/test_vss_round tests/cpp/test_vss_round.cpp -L $lib -lkyber_ed25519_hip -Wl,-rpath,$PWD/$lib -Wl,-rpath,/opt/rocm/lib || exit 1
for i in 1 2 3; do tests/cpp/_build/test_vss_round 64 43 | grep TIMING; done > $out/vss_round.log 2>&1
tests/cpp/_build/test_vss_round 16 11 | grep TIMING >> $out/vss_round.log 2>&1
tests/cpp/_build/test_vss_round 256 171 | grep TIMING >> $out/vss_round.log 2>&1
echo "vss rounds done"
timeout -k 10 200 python3 tools/concurrent_mid_calls.py > $out/concurrent_mid_calls.log 2>&1; echo "concurrent mid calls rc=$?"
timeout -k 10 300 python3 tools/fuzz_concurrent.py 120 6 42 > $out/fuzz_concurrent_shared_ctx.log 2>&1; echo "fuzz concurrent rc=$?"
timeout -k 10 100 python3 tools/small_ops_latency.py > $out/small_ops_latency.log 2>&1; echo "small ops rc=$?"
