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


def mesh_file(name: str) -> str:
    from tools.meshes import bunny_path, interior_path

    if name == "bunny":
        return bunny_path()
    if name == "interior":  # the generated, labelled stand-in for the missing sibenik.off
        return interior_path()
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


def options_for(rt, c: dict):
    """golden.json render entry -> rt_options."""
    return rt.Options.defaults(width=c["width"], height=c["height"], n_super_samples=c["ss"], ao_num_samples=c["ao"],
                               enable_ao=int(c["ao"] != 0), ao_max_distance=c["aod"], focal_length=c["focal"],
                               enable_shading=c["shading"], ao_alpha_min=c["amin"], ao_alpha_max=c["amax"],
                               bvh_method=0 if c["bvh"] == "longest" else 1)


def bits(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
