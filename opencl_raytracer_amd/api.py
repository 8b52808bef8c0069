"""ctypes binding of include/rt_hip.h, rt_hip_ring.h and rt_hip_debug.h (see those headers for the contract)."""
from __future__ import annotations

import ctypes as C
import weakref
import os
import sys
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path() -> str:
    """lib/libocrt_hip.so; OCRT_LIB_DIR (debug knob) names another directory of this package holding a build of
    the same library, e.g. lib_stamps for the instrumented one (make EXTRA_DEFS=-DOCRT_STAMPS LIBDIR=...)."""
    return os.path.join(_HERE, os.environ.get("OCRT_LIB_DIR", "lib"), "libocrt_hip.so")


class RtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"[rt {code}] {message}")
        self.code = code
        self.message = message


RT_E_INVALID, RT_E_NO_DEVICE, RT_E_DEVICE, RT_E_STATE, RT_E_IO = -1, -2, -3, -4, -5


class Options(C.Structure):
    """rt_options == RayTracer::Options (reference include/ray_tracer.h:17-30)."""

    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("focal_length", C.c_float),
        ("n_super_samples", C.c_uint32),
        ("enable_shading", C.c_int32),
        ("enable_ao", C.c_int32),
        ("ao_max_distance", C.c_float),
        ("ao_num_samples", C.c_uint32),
        ("ao_method", C.c_int32),
        ("ao_alpha_min", C.c_int32),
        ("ao_alpha_max", C.c_int32),
        ("bvh_method", C.c_int32),
    ]

    @classmethod
    def defaults(cls, **overrides) -> "Options":
        o = cls()
        load_library().rt_options_default(C.byref(o))
        for k, v in overrides.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        # the CLI's rule: AO is on iff the sample count is non-zero (reference src/render.cc:42)
        if "ao_num_samples" in overrides and "enable_ao" not in overrides:
            o.enable_ao = int(o.ao_num_samples != 0)
        return o

    @property
    def total_width(self) -> int:
        return load_library().rt_total_width(C.byref(self))

    @property
    def total_height(self) -> int:
        return load_library().rt_total_height(C.byref(self))


class _Stats(C.Structure):
    _fields_ = [
        ("primary_rays", C.c_uint64),
        ("primary_hits", C.c_uint64),
        ("ao_rays", C.c_uint64),
        ("ao_occluded", C.c_uint64),
    ]


_LIB: Optional[C.CDLL] = None

# name -> (restype, argtypes); also the list tests check against the header.
_SIGNATURES = {
    "rt_last_error": (C.c_char_p, []),
    "rt_last_error_code": (C.c_int, []),
    "rt_options_default": (None, [C.POINTER(Options)]),
    "rt_total_width": (C.c_uint32, [C.POINTER(Options)]),
    "rt_total_height": (C.c_uint32, [C.POINTER(Options)]),
    "rt_resize_cpu": (C.c_int, [C.POINTER(Options), C.c_void_p, C.c_void_p]),
    "rt_scene_load_off": (C.c_void_p, [C.c_char_p]),
    "rt_scene_from_arrays": (C.c_void_p, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "rt_scene_free": (None, [C.c_void_p]),
    "rt_scene_num_vertices": (C.c_uint32, [C.c_void_p]),
    "rt_scene_num_faces": (C.c_uint32, [C.c_void_p]),
    "rt_scene_build_bvh": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_scene_num_nodes": (C.c_uint32, [C.c_void_p]),
    "rt_scene_vertices": (C.c_void_p, [C.c_void_p]),
    "rt_scene_vnormals": (C.c_void_p, [C.c_void_p]),
    "rt_scene_faces": (C.c_void_p, [C.c_void_p]),
    "rt_scene_nodes": (C.c_void_p, [C.c_void_p]),
    "rt_scene_aabbs": (C.c_void_p, [C.c_void_p]),
    "rt_scene_triangles": (C.c_void_p, [C.c_void_p]),
    "rt_scene_sorted_faces": (C.c_void_p, [C.c_void_p]),
    "rt_create": (C.c_void_p, [C.POINTER(Options)]),
    "rt_create_on": (C.c_void_p, [C.POINTER(Options), C.c_int, C.c_uint32, C.c_uint32]),
    "rt_destroy": (None, [C.c_void_p]),
    "rt_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                            C.c_uint32, C.c_void_p]),
    "rt_upload_scene": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p]),
    "rt_render_async": (C.c_int, [C.c_void_p]),
    "rt_sync": (C.c_int, [C.c_void_p]),
    "rt_download": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_download_u8": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_local_rows": (C.c_uint32, [C.c_void_p]),
    "rt_download_u8_local": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_local_to_global_row": (C.c_uint32, [C.c_void_p, C.c_uint32]),
    "rt_partition_local_rows": (C.c_uint32, [C.POINTER(Options), C.c_uint32, C.c_uint32]),
    "rt_partition_global_row": (C.c_uint32, [C.POINTER(Options), C.c_uint32, C.c_uint32, C.c_uint32]),
    "rt_resize_into_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_use_private_stream": (C.c_int, [C.c_void_p]),
    "rt_get_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "rt_set_device_share": (C.c_int, [C.c_void_p, C.c_uint]),
    "rt_expect_frames": (C.c_int, [C.c_void_p, C.c_uint64]),
    "rt_get_stats": (C.c_int, [C.c_void_p, C.POINTER(_Stats)]),
    "rt_last_kernel_ms": (C.c_float, [C.c_void_p]),
    "rt_total_kernel_ms": (C.c_double, [C.c_void_p]),
    "rt_last_ao_ms": (C.c_float, [C.c_void_p]),
    "rt_total_ao_ms": (C.c_double, [C.c_void_p]),
    "rt_kernel_launches": (C.c_uint64, [C.c_void_p]),
    "rt_reset_timers": (None, [C.c_void_p]),
    "rt_ring_create": (C.c_void_p, [C.POINTER(Options), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rt_ring_destroy": (None, [C.c_void_p]),
    "rt_ring_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                 C.c_uint32, C.c_void_p]),
    "rt_ring_upload_scene": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_ring_device_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "rt_ring_set_gather_timeout": (C.c_int, [C.c_void_p, C.c_double]),
    "rt_ring_rccl_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rt_ring_set_calibration": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_ring_calibration": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "rt_set_ao_prefetch": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_walk_entries": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_double),
                        C.POINTER(C.c_double)]),
    "rt_ring_size": (C.c_uint32, [C.c_void_p]),
    "rt_ring_slots": (C.c_uint32, [C.c_void_p]),
    "rt_ring_local_rows": (C.c_uint32, [C.c_void_p]),
    "rt_ring_in_flight": (C.c_uint32, [C.c_void_p]),
    "rt_ring_host": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "rt_ring_set_graph_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_ring_set_pacing": (C.c_int, [C.c_void_p, C.c_float]),
    "rt_ring_bind_output": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "rt_ring_submit": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "rt_ring_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_void_p)]),
    "rt_ring_collect_into_device": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_ring_step": (C.c_int, [C.c_void_p]),
    "rt_ring_run": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rt_ring_drain": (C.c_int, [C.c_void_p]),
    "rt_ring_last_image_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "rt_ring_download_last": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_ring_reset_clock": (C.c_int, [C.c_void_p]),
    "rt_ring_keep_frame_times": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_ring_frame_times": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_float)]),
    "rt_ring_timers": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                                 C.POINTER(C.c_uint64)]),
    "rt_ring_reset_timers": (None, [C.c_void_p]),
    "rt_ring_cpu_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_uint64)]),
    "rt_rccl_available": (C.c_int, []),
    "rt_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "rt_ring_attach_rccl": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_ring_rccl_self_test": (C.c_int, [C.c_void_p]),
    "rt_print_info": (None, []),
    "rt_device_count": (C.c_int, []),
    # include/rt_hip_debug.h
    "rt_debug_measure_tile_costs": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int]),
    "rt_debug_set_order_policy": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float]),
    "rt_debug_set_primary_split": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rt_debug_tile_order_slots": (C.c_uint32, [C.c_void_p]),
    "rt_debug_tiles": (C.c_uint32, [C.c_void_p]),
    "rt_debug_tile_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_debug_set_tile_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    "rt_debug_set_frame_form": (C.c_int, [C.c_void_p, C.c_int]),
    "rt_debug_frame_is_fused": (C.c_int, [C.c_void_p]),
    "rt_debug_poison_hit_list": (C.c_int, [C.c_void_p]),
}


def load_library() -> C.CDLL:
    """Loads lib/libocrt_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C opencl_raytracer_amd/csrc`). There is no CPU fallback."
            )
        lib = C.CDLL(path)
        for name, (restype, argtypes) in _SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                # an OLDER build of the library loaded on purpose for an A/B run (OCRT_LIB_DIR + OCRT_ALLOW_OLD_LIB=1, set by
                # the A/B tools): it may lack the newest entry points, which such a run does not call.  Every other library
                # -- the product one, and one that OCRT_LIB_DIR merely locates -- must export every one of them.
                if os.environ.get("OCRT_LIB_DIR") and os.environ.get("OCRT_ALLOW_OLD_LIB") == "1":
                    print(f"opencl_raytracer_amd: {path} lacks {name} (OCRT_ALLOW_OLD_LIB=1: skipped)", file=sys.stderr)
                    continue
                raise
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = lib
    return _LIB


def _check(rc: int) -> None:
    if rc != 0:
        lib = load_library()
        raise RtError(rc, lib.rt_last_error().decode("utf-8", "replace"))


def _raise_last() -> None:
    lib = load_library()
    raise RtError(lib.rt_last_error_code(), lib.rt_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    return load_library().rt_device_count()


def _view(ptr: int, count: int, dtype) -> np.ndarray:
    if count == 0:
        return np.zeros(0, dtype=dtype)
    ctype = {np.uint32: C.c_uint32, np.float32: C.c_float}[dtype]
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,))
    return arr.copy()


class Scene:
    """CPU-side mesh + BVH: load_off_mesh / compute_vertex_normals / BVH::buildBVH."""

    def __init__(self, handle: int):
        self._h = handle

    @classmethod
    def load_off(cls, path: str) -> "Scene":
        h = load_library().rt_scene_load_off(os.fsencode(path))
        if not h:
            _raise_last()
        return cls(h)

    @classmethod
    def from_arrays(cls, vertices: np.ndarray, faces: np.ndarray) -> "Scene":
        v = np.ascontiguousarray(vertices, dtype=np.float32)
        if v.ndim != 2 or v.shape[1] not in (3, 4):
            raise ValueError("vertices must be (V,3) or (V,4)")
        if v.shape[1] == 3:
            v = np.concatenate([v, np.zeros((v.shape[0], 1), np.float32)], axis=1)
        v = np.ascontiguousarray(v)
        f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1)
        h = load_library().rt_scene_from_arrays(v.ctypes.data, v.shape[0], f.ctypes.data, f.size // 3)
        if not h:
            _raise_last()
        return cls(h)

    def close(self) -> None:
        if self._h:
            load_library().rt_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_bvh(self, method: int = 0) -> "Scene":
        _check(load_library().rt_scene_build_bvh(self._h, int(method)))
        return self

    @property
    def num_vertices(self) -> int:
        return load_library().rt_scene_num_vertices(self._h)

    @property
    def num_faces(self) -> int:
        return load_library().rt_scene_num_faces(self._h)

    @property
    def num_nodes(self) -> int:
        return load_library().rt_scene_num_nodes(self._h)

    def _arr(self, getter: str, count: int, dtype):
        return _view(getattr(load_library(), getter)(self._h), count, dtype)

    @property
    def vertices(self) -> np.ndarray:
        return self._arr("rt_scene_vertices", 4 * self.num_vertices, np.float32).reshape(-1, 4)

    @property
    def vnormals(self) -> np.ndarray:
        return self._arr("rt_scene_vnormals", 4 * self.num_vertices, np.float32).reshape(-1, 4)

    @property
    def faces(self) -> np.ndarray:
        return self._arr("rt_scene_faces", 3 * self.num_faces, np.uint32)

    @property
    def nodes(self) -> np.ndarray:
        return self._arr("rt_scene_nodes", self.num_nodes, np.uint32)

    @property
    def aabbs(self) -> np.ndarray:
        return self._arr("rt_scene_aabbs", 8 * self.num_nodes, np.float32).reshape(-1, 4)

    @property
    def triangles(self) -> np.ndarray:
        return self._arr("rt_scene_triangles", self.num_faces if self.num_nodes else 0, np.uint32)

    @property
    def sorted_faces(self) -> np.ndarray:
        return self._arr("rt_scene_sorted_faces", 3 * self.num_faces if self.num_nodes else 0, np.uint32)


class Host:
    """One render host == one OpenCLHost of the reference (ctor/upload/()/download)."""

    def __init__(self, options: Options, device: int = -1, rank: int = 0, nranks: int = 1):
        self.options = options
        self._h = load_library().rt_create_on(C.byref(options), device, rank, nranks)
        if not self._h:
            _raise_last()

    @classmethod
    def _borrowed(cls, options: Options, handle: int, owner=None) -> "Host":
        """A view of a host that `owner` (a FrameRing) owns: it keeps the owner alive, and the owner's close() takes the
        handle away (`_h = None`: every later call then fails with RtError / returns 0 instead of touching freed memory)."""
        h = cls.__new__(cls)
        h.options, h._h, h._owned, h._owner = options, handle, False, owner
        return h

    def close(self) -> None:
        if getattr(self, "_h", None):
            if getattr(self, "_owned", True):
                load_library().rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, faces, nodes, aabbs, vertices, vnormals) -> None:
        f = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1)
        n = np.ascontiguousarray(nodes, dtype=np.uint32).reshape(-1)
        a = np.ascontiguousarray(aabbs, dtype=np.float32).reshape(-1, 4)
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 4)
        vn = np.ascontiguousarray(vnormals, dtype=np.float32).reshape(-1, 4)
        if a.shape[0] != 2 * n.size:
            raise RtError(RT_E_INVALID, "aabbs must hold a (min,max) pair per node")
        if vn.shape[0] != v.shape[0]:
            raise RtError(RT_E_INVALID, "one normal per vertex expected")
        _check(load_library().rt_upload(self._h, f.ctypes.data, f.size // 3, n.ctypes.data, n.size, a.ctypes.data,
                                        v.ctypes.data, v.shape[0], vn.ctypes.data))

    def upload_scene(self, scene: Scene) -> None:
        _check(load_library().rt_upload_scene(self._h, scene._h))

    def render(self) -> None:
        _check(load_library().rt_render(self._h))

    __call__ = render

    def render_async(self) -> None:
        _check(load_library().rt_render_async(self._h))

    def sync(self) -> None:
        _check(load_library().rt_sync(self._h))

    def download(self) -> np.ndarray:
        img = np.empty((self.options.total_height, self.options.total_width), dtype=np.float32)
        _check(load_library().rt_download(self._h, img.ctypes.data))
        return img

    def download_u8(self) -> np.ndarray:
        img = np.empty((self.options.height, self.options.width), dtype=np.uint8)
        _check(load_library().rt_download_u8(self._h, img.ctypes.data))
        return img

    @property
    def local_rows(self) -> int:
        return load_library().rt_local_rows(self._h)

    def local_to_global_rows(self) -> np.ndarray:
        lib = load_library()
        return np.array([lib.rt_local_to_global_row(self._h, j) for j in range(self.local_rows)], dtype=np.int64)

    def download_u8_local(self) -> np.ndarray:
        rows = np.empty((self.local_rows, self.options.width), dtype=np.uint8)
        _check(load_library().rt_download_u8_local(self._h, rows.ctypes.data))
        return rows

    def resize_into_device(self, device_ptr: int) -> None:
        _check(load_library().rt_resize_into_device(self._h, device_ptr))

    def set_stream(self, hip_stream: int) -> None:
        _check(load_library().rt_set_stream(self._h, hip_stream))

    def use_private_stream(self) -> None:
        _check(load_library().rt_use_private_stream(self._h))

    def set_device_share(self, hosts: int) -> None:
        """`hosts` of them (this one included) take frames in turn on this GPU."""
        _check(load_library().rt_set_device_share(self._h, int(hosts)))

    @property
    def stream_handle(self) -> int:
        """The hipStream_t (as an integer) this host enqueues on."""
        p = C.c_void_p()
        _check(load_library().rt_get_stream(self._h, C.byref(p)))
        return int(p.value or 0)

    def set_ao_prefetch(self, on: bool) -> None:
        """Which form of the AO pass's node loop this host launches (include/rt_hip_debug.h, rt_set_ao_prefetch); same results."""
        _check(load_library().rt_set_ao_prefetch(self._h, int(on)))

    def set_frame_form(self, form: str) -> None:
        """"auto" (the library's rule), "fused" (both ray passes in one persistent launch) or "separate" (two kernels)."""
        _check(load_library().rt_debug_set_frame_form(self._h, {"auto": 0, "fused": 1, "separate": 2}[form]))

    @property
    def frame_is_fused(self) -> bool:
        return bool(load_library().rt_debug_frame_is_fused(self._h))

    def poison_hit_list(self) -> None:
        _check(load_library().rt_debug_poison_hit_list(self._h))

    def expect_frames(self, frames: int) -> None:
        """Announces a stream of frames (include/rt_hip.h, rt_expect_frames): uploads then prepare the walk intervals."""
        _check(load_library().rt_expect_frames(self._h, int(frames)))

    def measure_tile_costs(self, frames: int = 2, reorder: bool = True) -> None:
        """Measures what the tiles' AO packets cost (include/rt_hip_debug.h) and, with `reorder`, claims them by that."""
        _check(load_library().rt_debug_measure_tile_costs(self._h, int(frames), int(reorder)))

    def set_primary_split(self, above: int) -> None:
        """Primary pass: tiles of cost class `above` or more are cast in quarters (DeviceRenderer::setPrimarySplit; 0: none)."""
        _check(load_library().rt_debug_set_primary_split(self._h, int(above)))

    def set_order_policy(self, heavy: float, runway: float, split_above: float = -1.0) -> None:
        """How orders are made from measured costs (DeviceRenderer::orderByMeasuredCost); re-orders if costs have been measured."""
        _check(load_library().rt_debug_set_order_policy(self._h, float(heavy), float(runway), float(split_above)))

    def tile_order(self) -> dict:
        """The AO pass's claim order: list (eight segments), 8 x 3 constants, tile words, measured costs per tile."""
        lib = load_library()
        slots, tiles = lib.rt_debug_tile_order_slots(self._h), lib.rt_debug_tiles(self._h)
        order, constants = np.zeros(slots, np.uint32), np.zeros(24, np.uint32)
        words, costs = np.zeros(tiles, np.uint32), np.zeros(tiles, np.float32)
        _check(lib.rt_debug_tile_order(self._h, order.ctypes.data, constants.ctypes.data, words.ctypes.data, costs.ctypes.data))
        return {"order": order, "constants": constants.reshape(8, 3), "words": words, "costs": costs}

    def set_tile_order(self, order, constants) -> None:
        order = np.ascontiguousarray(order, np.uint32)
        constants = np.ascontiguousarray(constants, np.uint32).reshape(24)
        _check(load_library().rt_debug_set_tile_order(self._h, order.ctypes.data, len(order), constants.ctypes.data))

    def walk_entries(self) -> dict:
        """The intervals of the node array the tiles' any-hit packets walk (include/rt_hip_debug.h, rt_walk_entries)."""
        hit, narrowed, share, packet_share = C.c_uint32(), C.c_uint32(), C.c_double(), C.c_double()
        _check(load_library().rt_walk_entries(self._h, C.byref(hit), C.byref(narrowed), C.byref(share), C.byref(packet_share)))
        return {"tiles_hit": hit.value, "tiles_narrowed": narrowed.value, "mean_share": share.value,
                "mean_packet_share": packet_share.value}

    def stats(self) -> dict:
        s = _Stats()
        _check(load_library().rt_get_stats(self._h, C.byref(s)))
        return {k: int(getattr(s, k)) for k, _ in _Stats._fields_}

    @property
    def last_kernel_ms(self) -> float:
        return float(load_library().rt_last_kernel_ms(self._h))

    @property
    def total_kernel_ms(self) -> float:
        return float(load_library().rt_total_kernel_ms(self._h))

    @property
    def last_ao_ms(self) -> float:
        return float(load_library().rt_last_ao_ms(self._h))

    @property
    def total_ao_ms(self) -> float:
        return float(load_library().rt_total_ao_ms(self._h))

    @property
    def kernel_launches(self) -> int:
        return int(load_library().rt_kernel_launches(self._h))

    def reset_timers(self) -> None:
        load_library().rt_reset_timers(self._h)


class FrameRing:
    """rt_ring: several render hosts of one scene on one GPU that take frames in turn (include/rt_hip_ring.h, "frame
    ring").  The library owns the hosts, their streams and captured graphs, the frame bookkeeping and -- with a
    communicator attached -- the band gather; this class only forwards.  `submit()` enqueues a frame and returns at
    once, `collect()` waits for the oldest one, `run(k)` is k steps of a steady stream in ONE call into the library."""

    def __init__(self, options: Options, scene: Optional["Scene"] = None, device: int = 0, rank: int = 0, nranks: int = 1,
                 hosts: int = 3):
        self.options = options
        self._r = load_library().rt_ring_create(C.byref(options), device, rank, nranks, hosts)
        if not self._r:
            _raise_last()
        if scene is not None:
            self.upload_scene(scene)

    def close(self) -> None:
        if getattr(self, "_r", None):
            # the hosts handed out by host() point into the ring: they die with it
            for ref in getattr(self, "_lent", []):
                h = ref()
                if h is not None:
                    h._h = None
                    h._owner = None
            self._lent = []
            load_library().rt_ring_destroy(self._r)
            self._r = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_scene(self, scene: "Scene") -> None:
        _check(load_library().rt_ring_upload_scene(self._r, scene._h))

    def set_gather_timeout(self, seconds: float) -> None:
        """The exchange step waits at most this long (default 30 s) for a frame's gather, then fails with RtError."""
        _check(load_library().rt_ring_set_gather_timeout(self._r, float(seconds)))

    def rccl_info(self):
        """(ranks of the attached communicator by ncclCommCount, RCCL version code by ncclGetVersion); -1 = unknown."""
        a, b = C.c_int(), C.c_int()
        _check(load_library().rt_ring_rccl_info(self._r, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def set_calibration(self, on: bool) -> None:
        """Before an upload: whether the upload measures which form of the AO pass suits the scene (default: yes)."""
        _check(load_library().rt_ring_set_calibration(self._r, int(on)))

    def calibration(self):
        """(ms per ao_kernel without the look-ahead loads, with them -- 0.0: not measured --, whether they are in use)."""
        a, b, c = C.c_float(), C.c_float(), C.c_int()
        _check(load_library().rt_ring_calibration(self._r, C.byref(a), C.byref(b), C.byref(c)))
        return float(a.value), float(b.value), bool(c.value)

    def device_bytes(self):
        """(bytes of the scene's arrays on the device, copies of them among the hosts -- one --, bytes of everything requested)."""
        a, c, t = C.c_uint64(), C.c_uint32(), C.c_uint64()
        _check(load_library().rt_ring_device_bytes(self._r, C.byref(a), C.byref(c), C.byref(t)))
        return int(a.value), int(c.value), int(t.value)

    @property
    def size(self) -> int:
        return load_library().rt_ring_size(self._r)

    @property
    def slots(self) -> int:
        """Band buffers (2 x size): frame f is rendered into buffer f % slots."""
        return load_library().rt_ring_slots(self._r)

    @property
    def local_rows(self) -> int:
        return load_library().rt_ring_local_rows(self._r)

    @property
    def in_flight(self) -> int:
        return load_library().rt_ring_in_flight(self._r)

    def host(self, slot: int) -> "Host":
        """Host `slot` as a borrowed Host (statistics, timers, downloads of its last frame)."""
        h = load_library().rt_ring_host(self._r, slot)
        if not h:
            raise IndexError(slot)
        host = Host._borrowed(self.options, h, owner=self)  # (the view keeps the ring alive)
        if not hasattr(self, "_lent"):
            self._lent = []
        self._lent = [r for r in self._lent if r() is not None]
        self._lent.append(weakref.ref(host))
        return host

    @property
    def hosts(self):
        return [self.host(k) for k in range(self.size)]

    def set_graph_mode(self, on: bool) -> None:
        _check(load_library().rt_ring_set_graph_mode(self._r, int(on)))

    def set_pacing(self, beta: float) -> None:
        """0: submit as soon as a host is free; default 0.5 (include/rt_hip_ring.h, rt_ring_set_pacing)."""
        _check(load_library().rt_ring_set_pacing(self._r, float(beta)))

    def bind_output(self, slot: int, device_ptr: int) -> None:
        _check(load_library().rt_ring_bind_output(self._r, slot, device_ptr))

    def submit(self) -> int:
        f = C.c_uint64()
        _check(load_library().rt_ring_submit(self._r, C.byref(f)))
        return int(f.value)

    def collect_info(self):
        """Waits for the oldest frame; returns (frame number, slot, device address of its bands)."""
        f, s, p = C.c_uint64(), C.c_uint32(), C.c_void_p()
        _check(load_library().rt_ring_collect(self._r, C.byref(f), C.byref(s), C.byref(p)))
        return int(f.value), int(s.value), int(p.value or 0)

    def collect(self) -> np.ndarray:
        """Waits for the oldest frame and returns its 8-bit image (an unpartitioned ring, or rank 0 of a gathering one)."""
        self.collect_info()
        return self.download_last()

    def step(self) -> None:
        _check(load_library().rt_ring_step(self._r))

    def run(self, frames: int) -> None:
        _check(load_library().rt_ring_run(self._r, int(frames)))

    def drain(self) -> None:
        _check(load_library().rt_ring_drain(self._r))

    def last_image_device(self) -> int:
        p = C.c_void_p()
        _check(load_library().rt_ring_last_image_device(self._r, C.byref(p)))
        return int(p.value or 0)

    def download_last(self) -> np.ndarray:
        img = np.empty((self.options.height, self.options.width), dtype=np.uint8)
        _check(load_library().rt_ring_download_last(self._r, img.ctypes.data))
        return img

    def reset_clock(self) -> None:
        _check(load_library().rt_ring_reset_clock(self._r))

    def keep_frame_times(self, on: bool = True) -> None:
        _check(load_library().rt_ring_keep_frame_times(self._r, int(on)))

    def frame_times(self, frame: int):
        """(begin, ao begin, ao end, end) of a collected frame in ms since reset_clock()."""
        t = (C.c_float * 4)()
        _check(load_library().rt_ring_frame_times(self._r, frame, t))
        return tuple(float(x) for x in t)

    def timers(self) -> dict:
        k, a, nk, na = C.c_double(), C.c_double(), C.c_uint64(), C.c_uint64()
        _check(load_library().rt_ring_timers(self._r, C.byref(k), C.byref(nk), C.byref(a), C.byref(na)))
        return {"kernel_ms": k.value, "frames": int(nk.value), "ao_ms": a.value, "ao_frames": int(na.value)}

    def reset_timers(self) -> None:
        load_library().rt_ring_reset_timers(self._r)

    def cpu_times(self) -> dict:
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_uint64()
        _check(load_library().rt_ring_cpu_times(self._r, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return {"submit_s": a.value, "wait_s": b.value, "collect_s": c.value, "frames": int(n.value)}

    def attach_rccl(self, unique_id: bytes) -> None:
        """Collective over the job (ncclCommInitRank): every rank passes the 128 bytes rank 0 got from rccl_unique_id()."""
        if len(unique_id) != 128:
            raise ValueError("an RCCL unique id is 128 bytes")
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(load_library().rt_ring_attach_rccl(self._r, buf))

    def rccl_self_test(self) -> None:
        _check(load_library().rt_ring_rccl_self_test(self._r))


def rccl_available() -> bool:
    return bool(load_library().rt_rccl_available())


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _check(load_library().rt_rccl_unique_id(buf))
    return buf.raw


def resize_cpu(options: Options, tmp: np.ndarray) -> np.ndarray:
    """RayTracer::resize on the host (reference src/ray_tracer.cc:3-16)."""
    t = np.ascontiguousarray(tmp, dtype=np.float32)
    if t.size != options.total_width * options.total_height:
        raise RtError(RT_E_INVALID, "tmp must hold total_width*total_height floats")
    out = np.empty((options.height, options.width), dtype=np.uint8)
    _check(load_library().rt_resize_cpu(C.byref(options), t.ctypes.data, out.ctypes.data))
    return out


def partition_rows(options: Options, rank: int, nranks: int) -> np.ndarray:
    """Global output row of every local row of `rank` (rows >= height are padding)."""
    lib = load_library()
    count = lib.rt_partition_local_rows(C.byref(options), rank, nranks)
    return np.array([lib.rt_partition_global_row(C.byref(options), rank, nranks, j) for j in range(count)],
                    dtype=np.int64)


def pgm_bytes(image_u8: np.ndarray) -> bytes:
    """The file `render` writes: 'P5 W H 255\\n' + raw bytes (reference src/render.cc:135-136)."""
    h, w = image_u8.shape
    return f"P5 {w} {h} 255\n".encode() + np.ascontiguousarray(image_u8, dtype=np.uint8).tobytes()
