"""tests/cpp/test_long_running.cpp on the CPU: the product's deferred-point evaluator (csrc/defer.inc) over the CPU port, a window of 256 nodes,
300 dealer rounds — the arena's window is crossed more than a hundred times while two clients (the Rust binding's handle-only shape through the raw
C ABI; the C++ mirror in its default mode) hold a distributed key and one commitment per round and never call floor or materialize.  Nothing goes
stale, every answer is right.  With the table of kept values switched off (defer.keep_mib = 0) the same program ends the way round 5's binding did:
KYB_E_STALE, abort.  The GPU run at n = 64, t = 43 past the real window of 2^18 nodes: tests/test_gpu_long_running.py."""
import json
import subprocess

from test_gpu_vss_round import build


def _run(args, timeout=600):
    # (under AddressSanitizer + UBSan: the C++ mirror and the product's defer.inc — window moves, kept values, leaves taken back in)
    return subprocess.run([build("test_long_running", cpu_defer=True, sanitize=True)] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout)


def test_three_hundred_rounds_through_a_small_window(oracle):
    r = _run([6, 4, 300, 256])
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "OK", r.stdout[-2000:] + r.stderr[-2000:]
    soak = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("SOAK ")][0][5:])
    for client in ("handle_only", "cpp_mirror"):
        st = soak[client]
        assert st["ok"] and st["rounds"] == 300
        assert st["nodes"] > 17000 and st["left_the_window"] > st["nodes"] - 600 and st["in_window_now"] <= 256
        assert st["answers_from_kept_values"] >= 2 * 300 and st["operands_taken_back_in"] >= 300 - 2      # the key: marshalled and multiplied every round
        assert st["values_pushed_out"] == 0 and st["horner_fused"] >= 6 * 300


def test_without_the_kept_values_the_same_client_aborts_as_in_round_5(oracle):
    r = _run([6, 4, 300, 256, 0])
    assert r.returncode != 0
    assert "ABORT" in r.stdout and "stale handle" in r.stdout, r.stdout[-1500:]
