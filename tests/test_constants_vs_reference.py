"""The curve constants this repo DERIVES (tools/gen_constants.py for the HIP side, oracle_init() for
the oracle) equal, by value, the literals the reference carries in
src/group/edwards25519/constants.rs (D :60-63, D2 :65-68, SQRT_M1 :56-58, BASEEXT :70-87,
BASE :89-3738, PRIME / PRIME_ORDER :16-27).  The reference file is read as TEXT, in the build
container only; on a box without /root/reference the test is skipped."""
import os
import re
import sys

import pytest

REF = "/root/reference/src/group/edwards25519/constants.rs"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present on this box")

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bigint_model as M  # noqa: E402

BITS = [26, 25] * 5
P = M.P


def limbs_value(nums):
    v, off = 0, 0
    for x, b in zip(nums, BITS):
        v += x << off
        off += b
    return v % P


def ints(txt):
    return [int(x) for x in re.findall(r"-?\d+", txt)]


@pytest.fixture(scope="module")
def src():
    return open(REF).read()


def const_block(src, name):
    m = re.search(r"pub const %s: FieldElement = \[(.*?)\];" % name, src, flags=re.S)
    return limbs_value(ints(m.group(1)))


def test_field_constants(src, oracle):
    import gen_constants as G
    d, d2, sm1 = const_block(src, "D"), const_block(src, "D2"), const_block(src, "SQRT_M1")
    assert d == M.D == G.D and d2 == (2 * M.D) % P == G.D2 and sm1 == M.SQRT_M1 == G.SQRT_M1
    assert int.from_bytes(oracle.const_bytes(0), "little") == d
    assert int.from_bytes(oracle.const_bytes(1), "little") == d2
    assert int.from_bytes(oracle.const_bytes(2), "little") == sm1
    assert "57896044618658097711785492504343953926634992332820282019728792003956564819949" in src and int("57896044618658097711785492504343953926634992332820282019728792003956564819949") == P
    assert "7237005577332262213973186563042994240857116359379907606001950938285454250989" in src and int("7237005577332262213973186563042994240857116359379907606001950938285454250989") == M.L


def test_generated_header_is_current():
    import subprocess
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_constants.py"), "--check"]) == 0


def test_baseext_is_the_base_point(src, oracle):
    m = re.search(r"pub const BASEEXT: ExtendedGroupElement = ExtendedGroupElement \{(.*?)\n\};", src, flags=re.S)
    coords = re.findall(r"[xyzt]: \[(.*?)\]", m.group(1), flags=re.S)
    X, Y, Z, T = (limbs_value(ints(c)) for c in coords)
    zi = pow(Z, P - 2, P)
    assert (X * zi % P, Y * zi % P) == M.B            # Z != 1 in the reference, same point
    assert T * Z % P == X * Y % P
    assert oracle.encode(oracle.base()) == M.encode(M.B)


def test_base_table_values(src, oracle):
    """BASE[i][j] == (j+1) * 256^i * B as (y+x, y-x, 2dxy): all 256 entries against the oracle's
    regenerated table, and a sample against the big-int model."""
    body = src[src.index("pub const BASE:"):src.index("pub const WEAK_KEYS")]
    ents = re.findall(r"PreComputedGroupElement \{\s*y_plus_x: \[(.*?)\],\s*y_minus_x: \[(.*?)\],\s*xy2d: \[(.*?)\],\s*\}", body, flags=re.S)
    assert len(ents) == 256
    for n, (a, b, c) in enumerate(ents):
        i, j = divmod(n, 8)
        want = (limbs_value(ints(a)), limbs_value(ints(b)), limbs_value(ints(c)))
        t = oracle.base_table_bytes(i, j)
        got = tuple(int.from_bytes(t[32 * k:32 * k + 32], "little") for k in range(3))
        assert got == want, (i, j)
        if n % 37 == 0:
            x, y = M.mul_int((j + 1) * 256**i, M.B)
            assert want == ((y + x) % P, (y - x) % P, 2 * M.D * x * y % P)
