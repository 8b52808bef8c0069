"""Driver for rocprofv3 --pmc runs WITH FRAMES IN FLIGHT (GPU box): a ring of `hosts` render hosts of a bench workload
replays `frames` frames as a steady stream -- the mode the headline is measured in.  Uses only entry points that every
build of the library since round 3 has, so that OCRT_LIB_DIR can point it at an older build for a before / after pair.

    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d out -- python3 tools/analysis/inflight_pmc.py bunny_1080p_ao 3 12
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, mesh_path, workload_options  # noqa: E402

w = WORKLOADS[sys.argv[1]]
hosts, frames = int(sys.argv[2]), int(sys.argv[3])
opt = workload_options(rt, w)
scene = rt.Scene.load_off(mesh_path(w["mesh"])).build_bvh(opt.bvh_method)
ring = rt.FrameRing(opt, None, hosts=hosts)
if hasattr(ring, "set_calibration"):  # (round 4 on: no measuring frames among the counted ones; the hosts keep the default form)
    ring.set_calibration(False)
ring.upload_scene(scene)
ring.run(frames)
ring.drain()
print(f"{sys.argv[1]}: {frames} frames through a ring of {hosts} ({os.environ.get('OCRT_LIB_DIR', 'lib')})")
ring.close()
