"""The HIP path against THE REFERENCE ITSELF on the MI355X, bit for bit, with no stand-in anywhere.

What OpenCL's dot / cross / normalize / length / sin / cos / cospi / sinpi round to is the implementation's business
(OpenCL 1.2 section 7.4), so "the reference on this GPU" is src/intersect_kernel.cl compiled for gfx950 against ROCm's own
builtin library: oracle/_ref/ref_kernel_<tag>_strict.co (the clang driver links opencl.bc / ocml.bc itself;
-ffp-contract=off and correctly rounded / and sqrt for the kernel's own code).  The product fixes those builtins to their
IEEE definitions, which is why tests/test_refkernel_gpu.py MEASURES a distance between the two (570 of
2 073 600 pixels at 1080p with ambient occlusion) instead of asserting zero.

Here the distance is taken out of the comparison instead: a TEST-ONLY build of the HIP kernels (-DOCRT_OCML_BUILTINS,
opencl_raytracer_amd/lib_ocml) calls the very same library functions -- oracle/ocl_builtins.cl, compiled by the same clang
with the strict build's options to bitcode that carries the library's code, linked into the device side of kernels.hip;
the triangle records' products are made with the same fused chains on the host (scene_pack.cc) and the direction table by
the device's own trigonometry.  Everything ELSE is the product's: the rebuilt walk tree, the shared walks over padded
boxes, the exact leaf gate, the triangle predicate, the short reciprocal, the batching, the tile order, the finishing
kernel.  Zero differing float words between that build and the strict code object says: the HIP path's control flow and
every formula of its own are the reference's -- checked against the reference, not against a restatement of it.
"""
import numpy as np
import pytest

from conftest import bits, options_for
from orc import REFKERNEL_ALL_CASES as ALL_CASES, REFKERNEL_CASES_OTHER as CASES_OTHER

pytestmark = pytest.mark.gpu

_SCENES = {}


@pytest.mark.parametrize("frames", [None, 1000], ids=["one_shot", "stream"])
@pytest.mark.parametrize("name", ALL_CASES)
def test_hip_path_with_the_library_builtins_equals_the_reference_kernel(rt_ocml, golden, name, frames):
    import os

    import orc
    from conftest import mesh_file

    if not os.path.exists(os.path.join(orc.ORACLE_DIR, "libref_launch.so")):
        pytest.skip("oracle/libref_launch.so not built")
    c = golden["renders"][name]
    opt = options_for(rt_ocml, c)
    p = orc.params_from_options(opt)
    co = orc.ref_kernel_gfx950(p, c["ss"], "strict", build=False)
    if co is None:
        pytest.skip("no strict gfx950 code object of the reference kernel for this case under oracle/_ref/")
    key = (c["mesh"], c["bvh"])
    if key not in _SCENES:  # (handles are not shared between two copies of the library)
        sc = rt_ocml.Scene.load_off(mesh_file(c["mesh"])).build_bvh(0 if c["bvh"] == "longest" else 1)
        _SCENES[key] = (sc, orc.SceneArrays.from_scene(sc))
    scene, arrays = _SCENES[key]
    ref_img, _ = orc.RefGpu().render(co, p, arrays, block=CASES_OTHER.get(name, (16, 16)), repeats=0)
    host = rt_ocml.Host(opt, 0)
    if frames:
        host.expect_frames(frames)
    host.upload_scene(scene)
    host.render()
    img = host.download()
    differ = int(np.count_nonzero(bits(img) != bits(ref_img)))
    assert differ == 0, f"{differ} of {img.size} float words differ from the reference kernel's (strict build, ROCm's builtin library)"
    # (not a vacuous zero: this build is as far from the product's IEEE builtins as the strict reference build is from the
    # oracle -- the committed distance of tests/test_refkernel_gpu.py, float word for float word)
    import hashlib
    import json

    from conftest import ROOT

    with open(os.path.join(ROOT, "tests", "golden", "refkernel_gfx950_distance.json")) as f:
        committed = json.load(f)[name]["strict"]["float_words_differ"]
    assert (hashlib.sha256(img.tobytes()).hexdigest() != c["float_sha256"]) == (committed != 0)
    host.close()


@pytest.mark.parametrize("name", __import__("orc").REFKERNEL_RANDOM_CASES)
def test_random_sampling_equals_the_reference_kernel_too(rt_ocml, golden, name):
    """`-m random` (reference src/intersect_kernel.cl:128-183, 257-276: xorshift128 per sub-pixel, acos / sin / cos / cospi /
    sinpi of the draws) is outside the bit-exact contract with a CPU -- device and host libm round differently, the product
    is only checked statistically (tests/test_hip_parity.py).  Against the reference kernel ON THIS GPU there is no such
    excuse: with the library's own trigonometry (OCRT_SIN ..., kernels/common.hip.h) the frame must be the reference's bit
    for bit -- generator, draw order, the extra ray along the normal, the divisor AO_NUM_SAMPLES + 1 and all."""
    import os

    import orc
    from conftest import mesh_file

    if not os.path.exists(os.path.join(orc.ORACLE_DIR, "libref_launch.so")):
        pytest.skip("oracle/libref_launch.so not built")
    c = dict(golden["renders"][name])
    opt = options_for(rt_ocml, c)
    opt.ao_method = 1
    p = orc.params_from_options(opt)
    co = orc.ref_kernel_gfx950(p, c["ss"], "strict_static", build=False)  # (strict + `inline` read as `static`: oracle/Makefile)
    if co is None:
        pytest.skip("no strict gfx950 code object of the reference kernel with AO_METHOD=1 for this case under oracle/_ref/")
    key = (c["mesh"], c["bvh"])
    if key not in _SCENES:
        sc = rt_ocml.Scene.load_off(mesh_file(c["mesh"])).build_bvh(0 if c["bvh"] == "longest" else 1)
        _SCENES[key] = (sc, orc.SceneArrays.from_scene(sc))
    scene, arrays = _SCENES[key]
    ref_img, _ = orc.RefGpu().render(co, p, arrays, block=(16, 16), repeats=0)
    host = rt_ocml.Host(opt, 0)
    host.upload_scene(scene)
    host.render()
    img = host.download()
    differ = int(np.count_nonzero(bits(img) != bits(ref_img)))
    assert differ == 0, f"{differ} of {img.size} float words differ from the reference kernel's `-m random` frame"
    assert np.count_nonzero(img) > 0
    host.close()
