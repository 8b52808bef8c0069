"""Probe (GPU box): what ONE blocking frame would take if it were rendered as K band partitions of the same frame on K
streams of the one GPU, staggered -- part k + 1 submitted `delay` microseconds after part k, so that its primary pass runs
beside the earlier parts' ambient-occlusion passes -- against the frame in one piece (a ring of one host).

    python3 tools/analysis/split_frame_probe.py WORKLOAD [K,K,...] [delay_us,...]
"""
import os, statistics, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import opencl_raytracer_amd as rt
from bench import WORKLOADS, load_scene, workload_options
w = WORKLOADS[sys.argv[1]]
parts_list = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4]
delays = [float(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 40, 80, 120]
opt = workload_options(rt, w)
scene = load_scene(rt, w).build_bvh(opt.bvh_method)


def spin(us):
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t:
        pass


for parts in parts_list:
    rings = [rt.FrameRing(opt, scene, 0, k, parts, hosts=1) for k in range(parts)]
    for r in rings:
        r.set_pacing(0.0)
        r.run(5)
        r.drain()
    for delay in (delays if parts > 1 else [0]):
        times = []
        for _ in range(60):
            t0 = time.perf_counter()
            for k, r in enumerate(rings):
                if k and delay:
                    spin(delay)
                r.submit()
            for r in rings:
                r.collect_info()
            times.append((time.perf_counter() - t0) * 1e3)
        print(f"{sys.argv[1]}: {parts} part(s), {delay:.0f} us apart: median {statistics.median(times):.4f} ms, min {min(times):.4f}", flush=True)
    for r in rings:
        r.close()
