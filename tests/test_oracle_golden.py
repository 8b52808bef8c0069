"""CPU: the oracle (oracle/rt_oracle.c) against the golden vectors generated from
the reference's own kernel source (tests/golden/make_golden.py).  Scene arrays
come from the product's mesh loader + BVH builder, so this also pins those."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, bits, options_for

# everything except the two 1080p AO-sized frames stays in the default CPU run
FAST = [
    "bunny_256_s1_a0", "bunny_256_s1_a3", "bunny_1080p_s1_a0", "bunny_64_s1_a3", "bunny_101x77_s9_a2",
    "bunny_50x40_s5_a1_f15", "bunny_96x54_s1_a4_alpha", "blob_128x96_s4_a3", "blob_128x96_s4_a3_sah",
    "blob_80_s1_a5_noshade", "blob_33x17_s1_a0", "ties_33_s1_a3", "ties_33_s1_a3_sah", "ties_64_s4_a3",
    "ties_5x3_s1_a1", "single_32_s1_a3", "interior_hard_160x90_s4_a3",
]
SLOW = ["bunny_1080p_s1_a3", "bunny_600_defaults"]


def run_case(rt, oracle, golden, scene_for, name):
    import orc

    c = golden["renders"][name]
    opt = options_for(rt, c)
    _, arrays = scene_for(c["mesh"], c["bvh"])
    img, counters, _ = oracle.render(orc.params_from_options(opt), arrays)
    assert hashlib.sha256(img.tobytes()).hexdigest() == c["float_sha256"]
    u8 = oracle.resize(img, opt.width, opt.height, opt.n_super_samples)
    assert hashlib.md5(rt.pgm_bytes(u8)).hexdigest() == c["pgm_md5"]
    assert counters == c["counters"]
    return opt, img, u8


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_golden(rt, oracle, golden, scene_for, name):
    opt, img, u8 = run_case(rt, oracle, golden, scene_for, name)
    # the product's host-side resize agrees with the oracle's
    assert np.array_equal(rt.resize_cpu(opt, img), u8)
    dump = os.path.join(GOLDEN_DIR, f"render_{name}.npz")
    if os.path.exists(dump):
        with np.load(dump) as z:
            assert np.array_equal(bits(z["image"]), bits(img))
            assert np.array_equal(z["u8"], u8)


@pytest.mark.parametrize("name", SLOW)
def test_oracle_matches_golden_full_size(rt, oracle, golden, scene_for, name):
    run_case(rt, oracle, golden, scene_for, name)


def test_survey_md5_table(golden):
    """The five digests SURVEY.md 8c lists, measured there independently."""
    expect = {
        "bunny_256_s1_a0": "d9c663519f96ac30f7de4aa1640622a3",
        "bunny_256_s1_a3": "a056eb1c4096c7276a1feadda6d8c760",
        "bunny_1080p_s1_a0": "4a55458b7f5ee60d59665880feebfb78",
        "bunny_1080p_s1_a3": "e02eaeb53640206897ce798cda5cbd83",
        "bunny_600_defaults": "24bb1b1a645e057309983d43f5e2ad00",
    }
    for name, md5 in expect.items():
        assert golden["renders"][name]["pgm_md5"] == md5


def test_ao_table_default(oracle, golden):
    """The 28-direction UNIFORM table is the only libm-dependent quantity of the
    default path: a different libm on this box would show here first."""
    import orc

    class O:
        width = height = 8
        focal_length = 1.0
        n_super_samples = 1
        enable_shading = enable_ao = 1
        ao_max_distance = 0.2
        ao_num_samples = 3
        ao_method = 0
        ao_alpha_min, ao_alpha_max = 4, 90

    table = oracle.ao_table(orc.params_from_options(O))
    assert table.shape == (28, 3)
    got = [[float(x).hex() for x in row] for row in table]
    assert got == golden["ao_table_default_hex"]


def test_band_rendering_matches_full(rt, oracle, golden, scene_for):
    import orc

    c = golden["renders"]["blob_128x96_s4_a3"]
    opt = options_for(rt, c)
    _, arrays = scene_for(c["mesh"], c["bvh"])
    p = orc.params_from_options(opt)
    full, _, _ = oracle.render(p, arrays)
    parts = np.zeros_like(full)
    for y0, y1 in ((0, 50), (50, 51), (51, p.height)):
        oracle.render(p, arrays, rows=(y0, y1), image=parts)
    assert np.array_equal(bits(full), bits(parts))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference tree only exists in the build container")
def test_reference_kernel_agrees_here(rt, oracle, golden, scene_for):
    """Where the reference tree is mounted: compile its kernel for one more macro
    set that is NOT in golden.json and compare floats with the oracle."""
    import orc

    opt = rt.Options.defaults(width=40, height=24, n_super_samples=4, ao_num_samples=2, ao_max_distance=0.3,
                              focal_length=0.8)
    _, arrays = scene_for("blob", "longest")
    p = orc.params_from_options(opt)
    lib = orc.ref_kernel(p, opt.n_super_samples)
    ref_img, _ = orc.ref_render(lib, p, arrays)
    img, _, _ = oracle.render(p, arrays)
    assert np.array_equal(bits(ref_img), bits(img))
    assert np.array_equal(orc.RefHost().resize(ref_img, opt.width, opt.height, opt.n_super_samples),
                          oracle.resize(img, opt.width, opt.height, opt.n_super_samples))
