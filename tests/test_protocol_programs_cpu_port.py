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
