"""Unmodified, element-at-a-time DSS code on the engine: tests/cpp/test_dss_round.cpp restates one distributed-Schnorr signing round at one participant
(n = 64, t = 43) call by call as dss_sig.rs:173-326 makes the curve calls — schnorr::verify trait call by trait call, two PubPoly::eval Horner chains,
a variable-base and a fixed-base multiplication, an addition and a comparison per received partial signature — eagerly, recorded (kyb_defer_*), and
written with the batch entry points.  Here: the three transcripts and the CPU port's are equal, every byte in them is what the oracle and Python
integers say (tests/dss_check.py), the signature verifies as plain EdDSA under the distributed key, and the recorded run beats the CPU port."""
import json

import pytest

import dss_check
from test_gpu_vss_round import build, run_program

pytestmark = pytest.mark.gpu


def test_dss_round_call_by_call_eager_deferred_and_batched(oracle):
    n, t = 64, 43
    lines, timing = run_program(build("test_dss_round"), n, t)
    assert lines["E"] == lines["D"] == lines["B"] and len(lines["E"]) > 5 * n
    dss_check.check_transcript(lines["E"], n, t, oracle)
    cpu_lines, cpu = run_program(build("test_dss_round", cpu_port=True), n, t, "eager")
    assert cpu_lines["E"] == lines["E"]                              # the CPU port walks the identical sequence to the identical bytes
    timing["cpu_port_ms"] = cpu["eager_ms"]
    print("PHASES " + json.dumps(timing))
    st = timing["deferred_stats"]
    assert timing["eager_stats_nodes"] == 0                          # the eager run records nothing
    assert st["horner_fused"] >= 2 * (n - 1)                         # both PubPoly::eval chains of every partial signature were fused
    assert st["engine_calls"] <= 24 * n                              # against ~ (4 t + 20) (n - 1) batch-of-1 calls of the eager run
    # the phase that carries the round: n - 1 partial signatures, each 4 t + 5 curve calls in the reference
    assert timing["deferred_ms"]["process_partial_sigs"] * 3 <= timing["cpu_port_ms"]["process_partial_sigs"], timing
    assert timing["deferred_ms"]["process_partial_sigs"] * 10 <= timing["eager_ms"]["process_partial_sigs"], timing
    assert timing["batched_ms"]["round"] * 20 <= timing["cpu_port_ms"]["round"], timing


def test_small_dss_rounds_with_odd_shapes(oracle):
    """t = 1 (constant polynomials: no chain to fuse), t = n, the smallest group"""
    for n, t in ((2, 1), (3, 3), (5, 2), (7, 4)):
        lines, _ = run_program(build("test_dss_round"), n, t)
        assert lines["E"] == lines["D"] == lines["B"]
        dss_check.check_transcript(lines["E"], n, t, oracle)
