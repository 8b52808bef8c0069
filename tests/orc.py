"""ctypes views of the CHECKERS (test infrastructure only).

* :class:`Oracle`     -- oracle/liborc.so, this repo's C restatement of the
  reference kernel (oracle/rt_oracle.c).
* :class:`RefHost`    -- oracle/_ref/libref_host.so, the reference's own host
  objects (mesh / BVH / resize), only where it was built (this container).
* :func:`ref_kernel`  -- oracle/_ref/libref_kernel_<tag>.so, the reference's own
  kernel source compiled for x86-64 for one -D configuration.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
from typing import Optional

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
REFERENCE_TREE = "/root/reference"


class OrcParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("focal_length", C.c_float),
        ("shading_enable", C.c_int32),
        ("ao_enable", C.c_int32),
        ("ao_max_distance", C.c_float),
        ("ao_num_samples", C.c_uint32),
        ("ao_method", C.c_int32),
        ("ao_alpha_min", C.c_int32),
        ("ao_alpha_max", C.c_int32),
    ]


class OrcScene(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("faces", "nodes", "aabbs", "vertices", "normals")]


class OrcCounters(C.Structure):
    _fields_ = [
        (n, C.c_uint64)
        for n in (
            "primary_rays",
            "primary_hits",
            "primary_node_visits",
            "primary_tri_tests",
            "ao_rays",
            "ao_occluded",
            "ao_node_visits",
            "ao_tri_tests",
        )
    ]


def build_oracle() -> str:
    path = os.path.join(ORACLE_DIR, "liborc.so")
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    return path


def kernel_float(v: float) -> float:
    """Float option after the reference's -D round trip (include/compiler_options.h:13-19)."""
    return float(np.float32(float("%g" % np.float32(v))))


def params_from_options(opt, kernel_constants: bool = True) -> OrcParams:
    """rt_options (opencl_raytracer_amd.Options) -> the kernel's macro set
    (reference src/opencl_host.cc:42-53)."""
    n = int(np.sqrt(float(opt.n_super_samples)))
    p = OrcParams()
    p.width = opt.width * n
    p.height = opt.height * n
    p.focal_length = kernel_float(opt.focal_length) if kernel_constants else opt.focal_length
    p.shading_enable = int(bool(opt.enable_shading))
    p.ao_enable = int(bool(opt.enable_ao))
    p.ao_max_distance = kernel_float(opt.ao_max_distance) if kernel_constants else opt.ao_max_distance
    p.ao_num_samples = opt.ao_num_samples
    p.ao_method = opt.ao_method
    p.ao_alpha_min = opt.ao_alpha_min
    p.ao_alpha_max = opt.ao_alpha_max
    return p


class SceneArrays:
    """The five arrays of OpenCLHost::upload, kept alive for the C side."""

    def __init__(self, faces, nodes, aabbs, vertices, normals):
        self.faces = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1)
        self.nodes = np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1)
        self.aabbs = np.ascontiguousarray(aabbs, dtype=np.float32).reshape(-1, 4)
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 4)
        self.normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 4)

    @classmethod
    def from_scene(cls, scene) -> "SceneArrays":
        """From an opencl_raytracer_amd.Scene with a built BVH."""
        return cls(scene.sorted_faces, scene.nodes, scene.aabbs, scene.vertices, scene.vnormals)

    def c_struct(self) -> OrcScene:
        s = OrcScene()
        s.faces = self.faces.ctypes.data
        s.nodes = self.nodes.ctypes.data
        s.aabbs = self.aabbs.ctypes.data
        s.vertices = self.vertices.ctypes.data
        s.normals = self.normals.ctypes.data
        return s


class Oracle:
    def __init__(self):
        path = os.path.join(ORACLE_DIR, "liborc.so")
        if not os.path.exists(path):
            build_oracle()
        self.lib = C.CDLL(path)
        self.lib.orc_render_rows.restype = C.c_int
        self.lib.orc_render_rows.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcScene), C.c_void_p, C.c_uint32,
                                             C.c_uint32, C.POINTER(OrcCounters), C.c_int]
        self.lib.orc_resize.restype = None
        self.lib.orc_resize.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        self.lib.orc_ao_table.restype = C.c_uint32
        self.lib.orc_ao_table.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_uint32]
        self.lib.orc_ss_factor.restype = C.c_uint32
        self.lib.orc_ss_factor.argtypes = [C.c_uint32]

    def render(self, params: OrcParams, scene: SceneArrays, rows=None, nthreads: int = 0, image=None):
        """Returns (float image HxW, counters dict, threads used)."""
        if image is None:
            image = np.zeros((params.height, params.width), dtype=np.float32)
        y0, y1 = (0, params.height) if rows is None else rows
        counters = OrcCounters()
        cs = scene.c_struct()
        used = self.lib.orc_render_rows(C.byref(params), C.byref(cs), image.ctypes.data, y0, y1, C.byref(counters),
                                        nthreads)
        if used < 0:
            raise RuntimeError("oracle: AO direction table too large")
        return image, {k: int(getattr(counters, k)) for k, _ in OrcCounters._fields_}, used

    def resize(self, tmp: np.ndarray, width: int, height: int, n_super_samples: int) -> np.ndarray:
        t = np.ascontiguousarray(tmp, dtype=np.float32)
        out = np.empty((height, width), dtype=np.uint8)
        self.lib.orc_resize(t.ctypes.data, out.ctypes.data, width, height, n_super_samples)
        return out

    def ao_table(self, params: OrcParams) -> np.ndarray:
        n = self.lib.orc_ao_table(C.byref(params), None, 0)
        out = np.zeros((n, 3), dtype=np.float32)
        self.lib.orc_ao_table(C.byref(params), out.ctypes.data, n)
        return out


# ---------------------------------------------------------------------------
# reference-derived checkers (only where /root/reference or prebuilt _ref exist)
# ---------------------------------------------------------------------------

def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_TREE, "src"))


def kernel_defs(p: OrcParams, n_super_samples: int) -> str:
    """The -D string the reference would emit (src/opencl_host.cc:42-53)."""

    def fl(v):
        s = "%g" % np.float32(v)
        if float(np.float32(v)) == np.floor(float(np.float32(v))):
            s += "."
        return s + "f"

    defs = [f"-DWIDTH={p.width}", f"-DHEIGHT={p.height}", f"-DFOCAL_LENGTH={fl(p.focal_length)}",
            f"-DNSUPERSAMPLES={n_super_samples}"]
    if p.shading_enable:
        defs.append("-DSHADING_ENABLE")
    if p.ao_enable:
        defs.append("-DAO_ENABLE")
    defs += [f"-DAO_MAX_DISTANCE={fl(p.ao_max_distance)}", f"-DAO_NUM_SAMPLES={p.ao_num_samples}",
             f"-DAO_METHOD={p.ao_method}", f"-DAO_ALPHA_MIN={p.ao_alpha_min}", f"-DAO_ALPHA_MAX={p.ao_alpha_max}"]
    return " ".join(defs)


def ref_kernel_path(p: OrcParams, n_super_samples: int) -> str:
    tag = hashlib.sha1(kernel_defs(p, n_super_samples).encode()).hexdigest()[:16]
    return os.path.join(REF_DIR, f"libref_kernel_{tag}.so")


def ref_kernel(p: OrcParams, n_super_samples: int, build: bool = True) -> Optional[C.CDLL]:
    """The reference kernel for this macro set; built on demand when the
    reference tree is present, else only loaded if prebuilt."""
    path = ref_kernel_path(p, n_super_samples)
    if not os.path.exists(path):
        if not (build and reference_available()):
            return None
        tag = os.path.basename(path)[len("libref_kernel_"):-3]
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "ref-kernel", f"TAG={tag}",
                        f"DEFS={kernel_defs(p, n_super_samples)}"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(path)
    lib.ref_render_rows.restype = C.c_int
    lib.ref_render_rows.argtypes = [C.c_void_p] * 6 + [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    return lib


# ---- the reference kernel compiled for the MI355X itself (oracle/Makefile: ref-kernel-gfx950) ----
# Golden cases the reference's own kernel is compiled for gfx950 and launched for (tests/test_refkernel_gpu.py;
# __graft_entry__.build() makes the code objects): those whose NDRange the reference's 16 x 16 work-groups divide
# (SURVEY fact 0.8), and the headline frame, whose height 1080 they do not divide: launched with 16 x 8 work-groups.
REFKERNEL_CASES_16 = ["bunny_256_s1_a0", "bunny_256_s1_a3", "blob_128x96_s4_a3", "ties_64_s4_a3", "bunny_600_defaults"]
REFKERNEL_CASES_OTHER = {"bunny_1080p_s1_a0": (16, 8), "bunny_1080p_s1_a3": (16, 8)}
REFKERNEL_ALL_CASES = REFKERNEL_CASES_16 + sorted(REFKERNEL_CASES_OTHER)
# ... and golden cases rendered once more with `-m random` (AO_METHOD=1: the one sampling mode the reference compiles with
# on ROCm as it is): the strict gfx950 build only, for tests/test_ocml_pin.py
REFKERNEL_RANDOM_CASES = ["bunny_256_s1_a3", "blob_128x96_s4_a3", "ties_64_s4_a3"]

GFX950_MODES = ("default", "strict", "ieee_dot", "ieee_cross", "ieee_normalize", "ieee_length", "ieee_geom", "ieee_all")


def ref_kernel_gfx950_path(p: OrcParams, n_super_samples: int, mode: str) -> str:
    tag = hashlib.sha1(kernel_defs(p, n_super_samples).encode()).hexdigest()[:16]
    return os.path.join(REF_DIR, f"ref_kernel_{tag}_{mode}.co")


def ref_kernel_gfx950(p: OrcParams, n_super_samples: int, mode: str, build: bool = True) -> Optional[str]:
    """Path of the gfx950 code object of the reference kernel for this macro set and
    floating-point mode; built on demand where the reference tree is present."""
    path = ref_kernel_gfx950_path(p, n_super_samples, mode)
    if not os.path.exists(path):
        if not (build and reference_available()):
            return None
        tag = os.path.basename(path)[len("ref_kernel_"):-len(f"_{mode}.co")]
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "ref-kernel-gfx950", f"TAG={tag}", f"MODE={mode}",
                        f"DEFS={kernel_defs(p, n_super_samples)}"], check=True, stdout=subprocess.DEVNULL)
    return path


class RefGpu:
    """oracle/libref_launch.so: loads such a code object and launches `intersect` on cuda:0."""

    def __init__(self):
        path = os.path.join(ORACLE_DIR, "libref_launch.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-s", "-C", ORACLE_DIR, "ref-launch"], check=True)
        lib = C.CDLL(path)
        lib.refgpu_last_error.restype = C.c_char_p
        lib.refgpu_run.restype = C.c_int
        lib.refgpu_run.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.c_int, C.POINTER(C.c_float)]
        self.lib = lib

    def render(self, code_object: str, p: OrcParams, scene: SceneArrays, block=(16, 16), repeats: int = 0):
        """Returns (float image, average kernel ms over `repeats` launches or None)."""
        image = np.zeros((p.height, p.width), dtype=np.float32)
        ms = C.c_float(0.0)
        rc = self.lib.refgpu_run(os.fsencode(code_object), scene.faces.ctypes.data, scene.faces.size,
                                 scene.nodes.ctypes.data, scene.nodes.size, scene.aabbs.ctypes.data,
                                 scene.vertices.ctypes.data, scene.vertices.shape[0], scene.normals.ctypes.data,
                                 image.ctypes.data, p.width, p.height, block[0], block[1], repeats, C.byref(ms))
        if rc != 0:
            raise RuntimeError("reference kernel on the GPU: " + self.lib.refgpu_last_error().decode())
        return image, (float(ms.value) if repeats > 0 else None)


def ref_render(lib: C.CDLL, p: OrcParams, scene: SceneArrays, rows=None, nthreads: int = 0):
    image = np.zeros((p.height, p.width), dtype=np.float32)
    y0, y1 = (0, p.height) if rows is None else rows
    used = lib.ref_render_rows(scene.faces.ctypes.data, scene.nodes.ctypes.data, scene.aabbs.ctypes.data,
                               scene.vertices.ctypes.data, scene.normals.ctypes.data, image.ctypes.data, p.width, y0,
                               y1, nthreads)
    return image, used


class RefHost:
    """Reference mesh loader / BVH builder / resize (oracle/_ref/libref_host.so)."""

    def __init__(self):
        path = os.path.join(REF_DIR, "libref_host.so")
        if not os.path.exists(path):
            if not reference_available():
                raise FileNotFoundError(path)
            subprocess.run(["make", "-s", "-C", ORACLE_DIR, "ref-host"], check=True, stdout=subprocess.DEVNULL)
        lib = C.CDLL(path)
        lib.ref_scene_load.restype = C.c_void_p
        lib.ref_scene_load.argtypes = [C.c_char_p]
        lib.ref_scene_free.argtypes = [C.c_void_p]
        lib.ref_scene_num_vertices.restype = C.c_uint32
        lib.ref_scene_num_vertices.argtypes = [C.c_void_p]
        lib.ref_scene_num_faces.restype = C.c_uint32
        lib.ref_scene_num_faces.argtypes = [C.c_void_p]
        lib.ref_scene_get_mesh.argtypes = [C.c_void_p] * 4
        lib.ref_scene_build_bvh.restype = C.c_uint32
        lib.ref_scene_build_bvh.argtypes = [C.c_void_p, C.c_int]
        lib.ref_scene_get_bvh.argtypes = [C.c_void_p] * 5
        lib.ref_resize.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        lib.ref_total_dims.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        self.lib = lib

    def load(self, path: str):
        h = self.lib.ref_scene_load(os.fsencode(path))
        if not h:
            raise RuntimeError("reference loader failed")
        nv, nf = self.lib.ref_scene_num_vertices(h), self.lib.ref_scene_num_faces(h)
        v = np.zeros((nv, 4), np.float32)
        n = np.zeros((nv, 4), np.float32)
        f = np.zeros(3 * nf, np.uint32)
        self.lib.ref_scene_get_mesh(h, v.ctypes.data, n.ctypes.data, f.ctypes.data)
        return h, v, n, f

    def build_bvh(self, h, method: int):
        """Returns nodes, aabbs, triangles, sorted_faces.  The reference's SAH
        builder prints a progress line per candidate: silence fd 1 meanwhile."""
        nf = self.lib.ref_scene_num_faces(h)
        saved = os.dup(1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        try:
            os.dup2(devnull, 1)
            count = self.lib.ref_scene_build_bvh(h, method)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
            os.close(devnull)
        nodes = np.zeros(count, np.uint32)
        aabbs = np.zeros((2 * count, 4), np.float32)
        tris = np.zeros(nf, np.uint32)
        sorted_faces = np.zeros(3 * nf, np.uint32)
        self.lib.ref_scene_get_bvh(h, nodes.ctypes.data, aabbs.ctypes.data, tris.ctypes.data, sorted_faces.ctypes.data)
        return nodes, aabbs, tris, sorted_faces

    def free(self, h):
        self.lib.ref_scene_free(h)

    def resize(self, tmp: np.ndarray, width: int, height: int, n_super_samples: int) -> np.ndarray:
        t = np.ascontiguousarray(tmp, dtype=np.float32)
        out = np.empty((height, width), dtype=np.uint8)
        self.lib.ref_resize(t.ctypes.data, out.ctypes.data, width, height, n_super_samples)
        return out
