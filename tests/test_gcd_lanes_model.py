"""tools/gcd_lanes_model.py — the limb-exact model of the one-wavefront inversion of csrc/kernels_coop.hip (safegcd with the nine limbs of d, e, f, g
in lanes 0..8): every bound the device code relies on is an assertion there (64-bit columns, exact division by 2^30, loose limbs in [-1, 2^30 + 1],
|d| < 8p), checked on edge values and 3,400 random ones against x^(p-2).  CPU only; the device runs the same inputs in tests/test_gpu_coop.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_the_model_inverts_and_its_bounds_hold():
    import gcd_lanes_model as M
    M.main()
    P = M.P
    for z in (0, 1, P - 1, 2**255 - 20, 5, 2**128 + 3):
        assert M.inverse(z) == pow(z, P - 2, P)


def test_the_device_source_uses_the_constants_of_the_model():
    src = open(os.path.join(ROOT, "kyber-rs_amd", "csrc", "kernels_coop.hip")).read()
    p8 = 2**258 - 152                                        # 8p in 30-bit limbs, as the final reduction adds it
    limbs = [(p8 >> (30 * i)) & ((1 << 30) - 1) for i in range(8)] + [p8 >> 240]
    want = "{" + ", ".join(hex(x) for x in limbs) + "}"
    assert want in src, want
    assert sum(x << (30 * i) for i, x in enumerate(limbs)) == 8 * (2**255 - 19)
