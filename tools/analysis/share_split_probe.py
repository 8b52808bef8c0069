"""Probe (GPU box): one GPU's share of a frame split N ways, one frame at a time, with the heaviest tiles claimed in halves
from different thresholds on (rt_debug_set_order_policy's split_above): ms per frame of the slowest of three shares.
    python3 tools/analysis/share_split_probe.py [WORKLOAD]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, load_scene, workload_options  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p_ao"
w = WORKLOADS[name]
opt = workload_options(rt, w)
scene = load_scene(rt, w).build_bvh(opt.bvh_method)
for n in (8, 4, 2):
    for split_above in (0.0, 1.0, 0.5, 0.25, 0.12, 0.0):
        worst = 0.0
        for rank in range(min(n, 3)):
            ring = rt.FrameRing(opt, scene, 0, rank, n, hosts=1)
            ring.host(0).set_order_policy(2.0, 2.0, split_above)
            ring.run(10)
            ring.drain()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                ring.run(60)
                ring.drain()
                best = min(best, (time.perf_counter() - t0) / 60 * 1e3)
            worst = max(worst, best)
            ring.close()
        print(f"{name} 1/{n} share, tiles above {split_above:4.2f} of the pass's ideal length in halves: {worst:.4f} ms per frame", flush=True)
