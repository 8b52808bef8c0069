"""Probe (GPU box): the primary pass with its costliest tiles cast in quarters (rt_debug_set_primary_split), thresholds
interleaved: ms of the primary pass and of the blocking frame.
    python3 tools/analysis/primary_split_probe.py [WORKLOAD ...]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, load_scene, workload_options  # noqa: E402

ABOVE = [0, 64, 56, 48, 40, 32, 24, 16, 8] if "OCRT_SWEEP_ALL" in os.environ else [0, 64, 48]
for name in sys.argv[1:] or ["bunny_1080p_ao"]:
    w = WORKLOADS[name]
    opt = workload_options(rt, w)
    scene = load_scene(rt, w).build_bvh(opt.bvh_method)
    ring = rt.FrameRing(opt, scene, 0, 0, 1, hosts=1)
    host = ring.host(0)
    ring.run(20)
    ring.drain()
    words = host.tile_order()["words"]
    print(f"{name}: {int((words & 0xFF != 0).sum())} tiles with hits of {len(words)}, cost class 64: {int(((words >> 8) >= 64).sum())}, >= 48: {int(((words >> 8) >= 48).sum())}", flush=True)
    frame, primary = {a: [] for a in ABOVE}, {a: [] for a in ABOVE}
    for rep in range(5):
        for a in ABOVE:
            host.set_primary_split(a)
            ring.run(5)
            ring.drain()
            t0 = time.perf_counter()
            ring.run(100)
            ring.drain()
            frame[a].append((time.perf_counter() - t0) / 100 * 1e3)
            ring.set_graph_mode(False)
            ring.reset_clock()
            first = ring.submit()
            ring.collect_info()
            ring.run(20)
            ring.drain()
            t = [ring.frame_times(f) for f in range(first + 1, first + 21)]
            primary[a].append(statistics.median((x[1] - x[0]) if x[1] else (x[3] - x[0]) for x in t))
            ring.set_graph_mode(True)
    base = statistics.median(frame[0])
    for a in ABOVE:
        m = statistics.median(frame[a])
        print(f"{name}: tiles of cost class >= {a:2d} in quarters: primary pass {statistics.median(primary[a]):.4f} ms, blocking frame {m:.4f} ms ({100 * (m / base - 1):+.1f} %)", flush=True)
    ring.close()
