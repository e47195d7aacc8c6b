"""tests/cpp/test_long_running.cpp on the GPU: 110 Pedersen dealer rounds at n = 64, t = 43 — more than 600,000 recorded nodes, the arena's window of
2^18 is left behind twice over — by a handle-only client (the Rust binding's Copy point through the raw C ABI: nothing cached, no floor, no
materialize) and by the C++ mirror in its default mode, both holding a distributed key that is marshalled and multiplied in every round and one
commitment per round to the end (round-5 review item 2: the Rust drop-in aborted with KYB_E_STALE after about 43 rounds).  The small-window CPU run
of the same program with the product's evaluator: tests/test_long_running_cpu_port.py."""
import json
import subprocess

import pytest

from test_gpu_vss_round import build

pytestmark = pytest.mark.gpu


def test_a_hundred_dealer_rounds_past_the_window_with_handle_only_points():
    rounds = 110
    r = subprocess.run([build("test_long_running"), "64", "43", str(rounds)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "OK", r.stdout[-2000:] + r.stderr[-2000:]
    soak = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("SOAK ")][0][5:])
    print("SOAK " + json.dumps(soak))
    for client in ("handle_only", "cpp_mirror"):
        st = soak[client]
        assert st["ok"] and st["rounds"] == rounds
        assert st["nodes"] > (1 << 18) * 2 and st["left_the_window"] > (1 << 18)          # well past defer.max_nodes
        assert st["answers_from_kept_values"] >= rounds and st["operands_taken_back_in"] >= rounds // 2      # the key lives in the table for most of the run
        assert st["values_pushed_out"] == 0
        assert st["horner_fused"] >= 64 * rounds                                            # still one call per PubPoly::eval
