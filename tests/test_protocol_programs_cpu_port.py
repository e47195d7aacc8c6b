"""tests/cpp/test_vss_round.cpp and test_dkg_finish.cpp on the CPU port (tests/cpp/cpu_port_abi.cpp: the oracle behind the element-level entry points
of the C ABI): the call-by-call sequences of vss.rs:287-337, 361-386, 904-909 and dkg.rs:905-953 / poly.rs:566-603 through the C++ mirror of the trait
surface — the mirror's host side (Scalar arithmetic mod L, Lagrange coefficients, the encoding a Point keeps, SHA-512) and the sequences themselves —
against the oracle and Python integers.  CPU only; the GPU runs of the same programs are tests/test_gpu_vss_round.py."""
import synth
from test_gpu_vss_round import _check_against_oracle, build, run_program


def test_dealer_round_on_the_cpu_port(oracle):
    for n, t in ((6, 4), (3, 2), (2, 1)):
        lines, timing = run_program(build("test_vss_round", cpu_port=True), n, t, "eager")
        _check_against_oracle(lines["E"], n, t, oracle)
        assert set(timing["eager_ms"]) == {"new_dealer", "encrypted_deals", "verify_deals", "round"}


def test_end_of_a_dkg_on_the_cpu_port(oracle):
    L = synth.L
    for n, t in ((5, 3), (2, 1), (3, 5)):
        lines, _ = run_program(build("test_dkg_finish", cpu_port=True), n, t, "eager")
        by = {}
        for ln in lines["E"]:
            tag, val = ln.split()
            by.setdefault(tag, []).append(val)
        coeffs = [int.from_bytes(bytes.fromhex(c), "little") for c in by["COEFF"]]
        assert len(coeffs) == n * t and len(by["DISTCOMMIT"]) == t
        for j in range(t):
            total = sum(coeffs[d * t + j] for d in range(n)) % L
            assert by["DISTCOMMIT"][j] == oracle.mul_base(total.to_bytes(32, "little")).hex(), (n, t, j)
        assert by["RECOVERED"] == [oracle.mul_base(coeffs[0].to_bytes(32, "little")).hex()]


def test_the_three_programs_in_the_bindings_default_mode_on_the_cpu(oracle):
    """The C++ mirror in its DEFAULT (deferred) mode over the product's own evaluator (csrc/defer.inc compiled for the CPU on top of the CPU port): a
    dealer round, the end of a DKG and a DSS round recorded and evaluated in batches give, byte for byte, the transcript of the call-by-call run —
    and the chains are recognised (Horner per verifier, one sum per coefficient)."""
    for prog, n, t in (("test_vss_round", 6, 4), ("test_dkg_finish", 5, 3), ("test_dss_round", 6, 4), ("test_vss_round", 3, 2)):
        lines, timing = run_program(build(prog, cpu_defer=True), n, t, "all")
        assert timing["default_mode"] == "deferred"
        assert lines["E"] == lines["D"] and len(lines["E"]) > 10, prog
        st = timing["deferred_stats"]
        assert timing["eager_stats_nodes"] == 0 and st["nodes"] > 0 and st["engine_calls"] < st["nodes"]
        if prog == "test_vss_round":
            _check_against_oracle(lines["D"], n, t, oracle)
            assert st["horner_fused"] == n + 1
        if prog == "test_dkg_finish":
            assert st["sums_fused"] == t + 1
