"""Register / scratch budgets of the compiled kernels, from the compiler's own metadata (tools/kernel_resources.py: hipcc -S for gfx950,
no GPU needed).

The kernels a default call runs must not spill: scratch traffic in a loop that executes 3 x 10^5 VALU instructions per lane is the
first thing that would hurt, and the occupancy each kernel is launched for (waves per SIMD) fixes its VGPR cap.  Variants that are only
reachable through kyb_set_option (cross-checks, A/B leftovers) and the one-off table builders are listed with the scratch they have
today, so that a change there is at least noticed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def rows():
    """every kernel of the CROSS-CHECK build (the product's kernels plus the variants: -DKYB_CROSSCHECK)"""
    import kernel_resources
    return {r["kernel"]: r for r in kernel_resources.collect(["-DKYB_CROSSCHECK"])}


@pytest.fixture(scope="module")
def product_rows():
    """the kernels of the product build: the translation units of csrc/Makefile without CROSSCHECK=1"""
    import kernel_resources
    return {r["kernel"]: r for r in kernel_resources.collect([], units=kernel_resources.PRODUCT_UNITS)}


def test_nothing_in_the_product_library_spills_except_the_table_builders(product_rows):
    spilling = {k: r["scratch"] for k, r in product_rows.items() if r["scratch"]}
    assert all(k.startswith(("_Z12k_base_tablePj", "_Z14k_base_table32Pj", "_Z14k_base_table64Pj")) for k in spilling), spilling
    assert not any(k.startswith(("_Z5k_mulI", "_Z10k_mul_baseI", "_Z12k_mul_base32", "_Z6k_signI", "_Z12k_mul_ladderILi2E", "_Z12k_mul_ladderILi4E", "_Z12k_mul_base64ILb0E"))
                   for k in product_rows), "a variant kernel is compiled into the product"


def _find(rows, prefix):
    hit = [r for k, r in rows.items() if k.startswith(prefix)]
    assert len(hit) == 1, (prefix, [r["kernel"] for r in hit])
    return hit[0]


# (mangled-name prefix, VGPR cap = 512 / waves per SIMD the kernel is launched for, rounded down to the allocation granule of 8)
DEFAULT_PATH = [
    ("_Z12k_mul_ladderILi3E", 168),                       # variable base, mul.ladder_waves = 3 (default)
    ("_Z12k_mul_ladderILi2E", 256),
    ("_Z17k_mul_ladder_pair", 256),                       # two lanes per item, batches of at most one wavefront per SIMD
    ("_Z17k_mul_ladder_quad", 256),                       # four lanes per item (ge_ladder_quad.h), the same bound
    ("_Z19k_mul_ladder_pair_yILi1EE", 256), ("_Z19k_mul_ladder_pair_yILi2EE", 256), ("_Z16k_ladder_recover", 256),      # the same from wire encodings (two | four lanes per item): ladder on y, recovery after the decode
    ("_Z23k_mul_ladder_pair_y_decILi1EE", 256), ("_Z23k_mul_ladder_pair_y_decILi2EE", 256),      # ... with the decode as further workgroups of the same launch (ladder.y_only = 2)
    ("_Z19k_mul_ladder_pair_rILi1EE", 256), ("_Z19k_mul_ladder_pair_rILi2EE", 256),      # ... or decode R beside a ladder from points (verification with the keys given as points)
    ("_Z17k_verify_ladder_yILi1EE", 256), ("_Z17k_verify_ladder_yILi2EE", 256), ("_Z22k_verify_recover_final", 256), ("_Z13k_sig_scalars", 512),      # a DKG-sized verification in two launches
    ("_Z13k_finish_wavePK", 512), ("_Z21k_mul_base64_quarters", 512),      # the two mid-size kernels of round 6 (one wavefront per SIMD)
    ("_Z12k_mul_base64ILb1ELi1024E", 128),                # fixed base, full batches: 4 waves/SIMD, the table owns the LDS
    ("_Z12k_mul_base64ILb1ELi256E", 512),                 # fixed base, batches that do not fill the chip: 1 wave/SIMD
    ("_Z11k_mont_prepPKim", 256), ("_Z8k_finishPK", 256), ("_Z9k_finish4PK", 256), ("_Z16k_encode_batchedPKimPh", 256), ("_Z17k_encode_batched4PKimPh", 256),
    ("_Z20k_decode_or_identityPKhmPiPh", 256), ("_Z16k_decode_to_projPKhm", 256), ("_Z10k_pair_sumP", 256), ("_Z13k_ext_to_projPKim", 256),
    ("_Z13k_verify_prepPKhS0_S0_PKjmPhS3_S3_Pi", 256), ("_Z15k_verify_prep_rPKhmPh", 256), ("_Z13k_verify_hashPKh", 256), ("_Z14k_verify_finalPK", 256), ("_Z13k_verify_diffP", 256),
    ("_Z18k_verify_final_encPK", 256), ("_Z19k_verify_final_enc4PK", 256), ("_Z14k_verify_fixupPKhmS0_iPh", 256), ("_Z11k_sign_hashPKhS0_S0_PKjm", 256), ("_Z12k_eddsa_prepPKhS0_PKjm", 256),
    ("_Z11k_poly_evalILb1EE", 256), ("_Z16k_poly_eval_partPKii", 256), ("_Z7k_equalPKiS0_mPh", 256), ("_Z5k_addPKiS0_mPii", 256), ("_Z8k_decodePKhmPiPh", 256),
    ("_Z10k_mul_coopPKhPKim", 256), ("_Z15k_mul_base_coopPKhS0_mm", 256), ("_Z14k_mul_enc_coopPKhS0_m", 256), ("_Z13k_verify_coopPKhS0_S0_PKjmi", 256),
    ("_Z11k_sign_coopPKhS0_S0_S0_PKjm", 256), ("_Z13k_decode_coopPKhmPiPhi", 256), ("_Z13k_finish_coopPK", 256), ("_Z10k_sum_coopPKjPKimm", 256),
    ("_Z16k_poly_eval_coopPKiiPKjmim", 256), ("_Z15k_poly_eval_segPKiiPKjmmii", 256), ("_Z15k_poly_eval_sumPKjmi", 256),
    ("_Z19k_pripoly_eval_part", 256), ("_Z18k_pripoly_eval_sum", 256),
    ("_Z15k_diag_mad_peakILb0E", 64), ("_Z15k_diag_mad_peakILb1E", 64),      # the peak microbenchmark runs 8 wavefronts per SIMD
    ("_Z12k_mul_base64ILb1ELi768E", 168), ("_Z12k_mul_base64ILb1ELi512E", 256),   # selectable workgroup sizes of the fixed-base kernel
]


@pytest.mark.parametrize("prefix,cap", DEFAULT_PATH)
def test_default_path_kernels_do_not_spill(rows, prefix, cap):
    r = _find(rows, prefix)
    assert r["scratch"] == 0, (r["kernel"], r["scratch"])
    assert r["vgpr"] <= cap, (r["kernel"], r["vgpr"], cap)


# not on any default path: selectable cross-checks and the table builders that run once per context.  prefix -> bytes of scratch today.
KNOWN_SPILLS = {
    "_Z12k_base_tablePj": 1328, "_Z14k_base_table32Pj": 1328, "_Z14k_base_table64Pj": 1328,      # per-thread entry tables on the stack; 8 + 13 + 22 wavefronts at init
    "_Z12k_mul_ladderILi4E": 188,                          # mul.ladder_waves = 4 (128 VGPRs): measured slower, kept as the A/B's other arm
    "_Z12k_mul_base64ILb0ELi1024E": 68,                    # fused per-item inversion (finish.batched = 0)
    "_Z12k_mul_base32ILb1E": 12,                           # mul_base.radix = 32
    "_Z10k_mul_baseILi1ELi512ELb1E": 8, "_Z10k_mul_baseILi0ELi512ELb0E": 36, "_Z10k_mul_baseILi1ELi512ELb0E": 48,      # radix 16, mul_base.block = 512
    "_Z5k_mulILi0ELb1ELb1E": 76, "_Z5k_mulILi1ELb1ELb1E": 164, "_Z5k_mulILi1ELb0ELb1E": 140,        # windowed variable base (mul.algo = 0)
    "_Z5k_mulILi0ELb1ELb0E": 76, "_Z5k_mulILi0ELb0ELb0E": 44, "_Z5k_mulILi1ELb1ELb0E": 172, "_Z5k_mulILi1ELb0ELb0E": 188,
}


def test_only_the_listed_variants_spill(rows):
    spilling = {k: r["scratch"] for k, r in rows.items() if r["scratch"]}
    unlisted = {k: v for k, v in spilling.items() if not any(k.startswith(p) for p in KNOWN_SPILLS)}
    assert not unlisted, f"kernels outside the known list use scratch: {unlisted}"
    for prefix, was in KNOWN_SPILLS.items():
        r = _find(rows, prefix)
        assert r["scratch"] <= 2 * was + 64, (r["kernel"], r["scratch"], was)       # a listed variant that got much worse is worth a look too


def test_fixed_base_kernel_owns_the_lds(rows):
    r = _find(rows, "_Z12k_mul_base64ILb1ELi1024E")
    assert r["lds"] == 163200 and r["lds"] <= 160 * 1024
