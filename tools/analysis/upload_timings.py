"""Where an upload's time goes (A/B build of the library: OCRT_LIB_DIR=.../lib_knobs OCRT_UPLOAD_TIMINGS=1), per workload:
    OCRT_LIB_DIR=$PWD/opencl_raytracer_amd/lib_knobs OCRT_UPLOAD_TIMINGS=1 python3 tools/analysis/upload_timings.py bunny_1080p_ao ..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, load_scene, workload_options  # noqa: E402

for name in sys.argv[1:]:
    w = WORKLOADS[name]
    opt = workload_options(rt, w)
    scene = load_scene(rt, w).build_bvh(opt.bvh_method)
    for attempt in range(2):  # (the first upload of a process also pays for the device's start-up)
        host = rt.Host(opt, 0)
        t0 = time.perf_counter()
        host.upload_scene(scene)
        t1 = time.perf_counter()
        print(f"== {name} upload #{attempt}: {1e3 * (t1 - t0):.2f} ms; intervals {host.walk_entries()}", flush=True)
        host.close()
