import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    import orc

    return orc.Oracle()


@pytest.fixture(scope="session")
def rt():
    """The product binding; building is the job of __graft_entry__.build()."""
    import opencl_raytracer_amd as rt_mod

    rt_mod.load_library()
    return rt_mod


@pytest.fixture(scope="session")
def rt_knobs():
    """A second, independent binding on the A/B build of the library (make EXTRA_DEFS=-DOCRT_DEBUG_KNOBS ->
    opencl_raytracer_amd/lib_knobs): the only build that reads the OCRT_* scheduling knobs from the environment and
    that still holds the first-generation kernels.  The product library contains neither."""
    import importlib.util

    path = os.path.join(ROOT, "opencl_raytracer_amd", "lib_knobs", "libocrt_hip.so")
    if not os.path.exists(path):
        pytest.fail("opencl_raytracer_amd/lib_knobs/libocrt_hip.so is missing: run __graft_entry__.build()")
    spec = importlib.util.spec_from_file_location("ocrt_api_knobs", os.path.join(ROOT, "opencl_raytracer_amd", "api.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    old = os.environ.get("OCRT_LIB_DIR")
    os.environ["OCRT_LIB_DIR"] = "lib_knobs"
    try:
        spec.loader.exec_module(mod)
        mod.load_library()
        assert mod.lib_path() == path
    finally:
        if old is None:
            del os.environ["OCRT_LIB_DIR"]
        else:
            os.environ["OCRT_LIB_DIR"] = old
    return mod


def second_binding(lib_dir: str, module_name: str):
    """An independent ctypes binding of the same api.py on another build of the library (opencl_raytracer_amd/<lib_dir>)."""
    import importlib.util

    path = os.path.join(ROOT, "opencl_raytracer_amd", lib_dir, "libocrt_hip.so")
    if not os.path.exists(path):
        pytest.fail(f"opencl_raytracer_amd/{lib_dir}/libocrt_hip.so is missing: run __graft_entry__.build()")
    spec = importlib.util.spec_from_file_location(module_name, os.path.join(ROOT, "opencl_raytracer_amd", "api.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    old = os.environ.get("OCRT_LIB_DIR")
    os.environ["OCRT_LIB_DIR"] = lib_dir
    try:
        spec.loader.exec_module(mod)
        mod.load_library()
        assert mod.lib_path() == path
    finally:
        if old is None:
            del os.environ["OCRT_LIB_DIR"]
        else:
            os.environ["OCRT_LIB_DIR"] = old
    return mod


@pytest.fixture(scope="session")
def rt_ocml():
    """A binding on the TEST-ONLY build whose dot / cross / normalize / length and ring-angle trigonometry are ROCm's own
    OpenCL library functions (make EXTRA_DEFS=-DOCRT_OCML_BUILTINS DEVICE_BITCODE=oracle/_ref/ocl_builtins_gfx950.bc ->
    opencl_raytracer_amd/lib_ocml; kernels/common.hip.h): the build that must equal the reference kernel compiled
    against that library bit for bit (tests/test_ocml_pin.py)."""
    return second_binding("lib_ocml", "ocrt_api_ocml")


def mesh_file(name: str) -> str:
    from tools.meshes import bunny_path, interior_hard_path, interior_path

    if name == "bunny":
        return bunny_path()
    if name == "interior":  # the generated, labelled stand-in for the missing sibenik.off
        return interior_path()
    if name == "interior_hard":  # ... and its harder variant (huge triangles beside fine ornament, slivers)
        return interior_hard_path()
    return os.path.join(GOLDEN_DIR, "meshes", name + ".off")


_SCENES = {}


@pytest.fixture(scope="session")
def scene_for(rt):
    """(mesh, bvh) -> (product Scene with BVH, SceneArrays for the oracle); cached."""
    import orc

    def get(mesh: str, bvh: str):
        key = (mesh, bvh)
        if key not in _SCENES:
            sc = rt.Scene.load_off(mesh_file(mesh)).build_bvh(0 if bvh == "longest" else 1)
            _SCENES[key] = (sc, orc.SceneArrays.from_scene(sc))
        return _SCENES[key]

    return get


_KNOB_SCENES = {}


@pytest.fixture(scope="session")
def scene_for_knobs(rt_knobs):
    """scene_for, through the A/B build's own binding (handles are not shared between two copies of the library)."""
    import orc

    def get(mesh: str, bvh: str):
        key = (mesh, bvh)
        if key not in _KNOB_SCENES:
            sc = rt_knobs.Scene.load_off(mesh_file(mesh)).build_bvh(0 if bvh == "longest" else 1)
            _KNOB_SCENES[key] = (sc, orc.SceneArrays.from_scene(sc))
        return _KNOB_SCENES[key]

    return get


def options_for(rt, c: dict):
    """golden.json render entry -> rt_options."""
    return rt.Options.defaults(width=c["width"], height=c["height"], n_super_samples=c["ss"], ao_num_samples=c["ao"],
                               enable_ao=int(c["ao"] != 0), ao_max_distance=c["aod"], focal_length=c["focal"],
                               enable_shading=c["shading"], ao_alpha_min=c["amin"], ao_alpha_max=c["amax"],
                               bvh_method=0 if c["bvh"] == "longest" else 1)


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
