"""Probe (GPU box): the passes of ONE GPU's share of a frame split N ways, one frame at a time, with the primary pass's
quarters from different cost classes on (rt_debug_set_primary_split): ms of the primary pass, the any-hit pass, the frame.
    python3 tools/analysis/share_passes_probe.py [WORKLOAD]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, load_scene, workload_options  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p_ao"
w = WORKLOADS[name]
opt = workload_options(rt, w)
scene = load_scene(rt, w).build_bvh(opt.bvh_method)
for n in (8, 4, 2):
    for rank in sorted({0, n // 2, n - 1}):
        ring = rt.FrameRing(opt, scene, 0, rank, n, hosts=1)
        ring.set_graph_mode(False)
        host = ring.host(0)
        ring.run(10)
        ring.drain()
        for above in (64, 48, 32, 16, 8, 2):
            host.set_primary_split(above)
            ring.run(5)
            ring.drain()
            ring.reset_clock()
            first = ring.submit()
            ring.collect_info()
            ring.run(40)
            ring.drain()
            t = [ring.frame_times(f) for f in range(first + 1, first + 41)]
            total = statistics.median(x[3] - x[0] for x in t)
            primary = statistics.median((x[1] - x[0]) if x[1] else (x[3] - x[0]) for x in t)
            ao = statistics.median(x[2] - x[1] for x in t)
            print(f"{name} share {rank} of {n}, quarters from class {above:2d}: primary {primary:.4f} ms, any-hit {ao:.4f}, frame's kernels {total:.4f}", flush=True)
        ring.close()
