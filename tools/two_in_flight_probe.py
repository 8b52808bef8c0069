"""Probe (GPU box): frames per second of ONE renderer against TWO renderers of the same scene on the same GPU, each on
its own stream, frames enqueued alternately -- i.e. what overlapping the end of one frame's ambient-occlusion pass
(falling occupancy) with the next frame's primary pass would buy.

    python3 tools/two_in_flight_probe.py [workload] [frames]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, load_scene, mesh_path, workload_options  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p_ao"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    w = WORKLOADS[name]
    opt = workload_options(rt, w)
    scene = load_scene(rt, w).build_bvh(opt.bvh_method)
    hosts = [rt.Host(opt, 0) for _ in range(2)]
    for h in hosts:
        h.upload_scene(scene)
        h.use_private_stream()
    for h in hosts:  # warm-up
        for _ in range(3):
            h.render_async()
        h.sync()
    for label, users in (("one renderer", hosts[:1]), ("two renderers, alternating", hosts), ("one renderer", hosts[:1]),
                         ("two renderers, alternating", hosts)):
        t0 = time.perf_counter()
        for k in range(frames):
            users[k % len(users)].render_async()
        for h in users:
            h.sync()
        dt = time.perf_counter() - t0
        print(f"{name}: {label}: {dt / frames * 1e3:.4f} ms per frame")


if __name__ == "__main__":
    main()
