"""MI355X-native ray-casting path behind the reference's render-host API.

The product is native code: ``csrc/`` builds ``lib/libocrt_hip.so`` (C ABI of
``include/rt_hip.h`` + the C++ ``HipHost`` class + gfx950 kernels) and the
``bin/render`` CLI.  This package is only the thin ctypes view of that C ABI
used by the tests and ``bench.py``; it contains no compute and no fallback --
if the shared library is missing, importing :mod:`opencl_raytracer_amd.api`
raises.
"""
from .api import (  # noqa: F401
    FrameRing,
    Host,
    Options,
    RtError,
    Scene,
    device_count,
    lib_path,
    load_library,
    partition_rows,
    pgm_bytes,
    rccl_available,
    rccl_unique_id,
    resize_cpu,
)
