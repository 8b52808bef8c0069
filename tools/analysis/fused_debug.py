"""Debug aid (GPU box): the fused frame against the two-kernel frame of the same host, with and without a poisoned hit
list; counts and places the sub-pixels that differ.   python3 tools/analysis/fused_debug.py [workload]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, load_scene, workload_options  # noqa: E402

w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "bunny_600_defaults"]
opt = workload_options(rt, w)
scene = load_scene(rt, w).build_bvh(opt.bvh_method)
host = rt.Host(opt, 0)
host.expect_frames(1000)
host.upload_scene(scene)
host.set_frame_form("separate")
host.render()
ref = host.download().view(np.uint32)
print("separate:", host.stats(), flush=True)
for poison in (False, True, True, False):
    host.set_frame_form("fused")
    if poison:
        host.poison_hit_list()
    host.render()
    try:
        img = host.download().view(np.uint32)
    except rt.RtError as e:
        print("poison", poison, "error:", e)
        continue
    bad = np.argwhere(img != ref)
    print("  ", host.stats())
    print(f"poison {poison}: {len(bad)} sub-pixels differ; kernel {host.last_kernel_ms:.3f} ms, fused pass {host.last_ao_ms:.3f} ms", flush=True)
    if len(bad):
        tiles = {(int(y) // 8, int(x) // 8) for y, x in bad}
        print("   tiles:", len(tiles), sorted(tiles)[:12], "NaN:", int(np.isnan(img.view(np.float32)).sum()))
        y, x = bad[0]
        print("   first:", (int(y), int(x)), hex(int(img[y, x])), "want", hex(int(ref[y, x])))
host.close()
