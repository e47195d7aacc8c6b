"""tests/cpp/test_dss_round.cpp on the CPU port (tests/cpp/cpu_port_abi.cpp: the oracle behind the element-level entry points of the C ABI): the
call-by-call DSS round of dss_sig.rs:173-326 through the C++ mirror of the trait surface — its host side (Scalar arithmetic mod L, canonical checks,
SHA-512, Lagrange recovery) and the sequence itself — against Python integers and the oracle's own verification.  CPU only; the GPU run of the
same program is tests/test_gpu_dss_round.py."""
import dss_check
from test_gpu_vss_round import build, run_program


def test_dss_round_on_the_cpu_port(oracle):
    for n, t in ((6, 4), (2, 1), (5, 5)):
        lines, timing = run_program(build("test_dss_round", cpu_port=True), n, t, "eager")
        dss_check.check_transcript(lines["E"], n, t, oracle)
        assert set(timing["eager_ms"]) == {"new_dss", "partial_sig", "process_partial_sigs", "round"}
