"""Per-rank kernel time of the band partition, measured on one GPU: renders every rank's
share of the frame for 1, 2, 4 and 8 ranks (what each GPU of an N-GPU run would do), one frame at a time and with
three frames in flight (rt.FrameRing, wall clock per frame).

    python3 tools/partition_probe.py [workload] [ranks,ranks,...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, load_scene, mesh_path, workload_options  # noqa: E402

w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p_ao"]
opt = workload_options(rt, w)
scene = load_scene(rt, w).build_bvh(opt.bvh_method)
for n in ([int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8)):
    worst, worst_ao = 0.0, 0.0
    for rank in range(n):
        host = rt.Host(opt, 0, rank, n)
        host.upload_scene(scene)
        for _ in range(3):
            host.render()
        host.reset_timers()
        for _ in range(10):
            host.render_async()
        host.sync()
        worst = max(worst, host.total_kernel_ms / host.kernel_launches)
        worst_ao = max(worst_ao, host.total_ao_ms / host.kernel_launches)
        host.close()
    worst_ring = 0.0
    for rank in range(n):
        ring = rt.FrameRing(opt, scene, 0, rank, n, hosts=3)
        for phase, frames in (("warm", 9), ("timed", 60)):
            t0 = time.perf_counter()
            ring.run(frames)
            ring.drain()
            dt = (time.perf_counter() - t0) / frames * 1e3
        worst_ring = max(worst_ring, dt)
        ring.close()
    print(f"{n} ranks: slowest rank {worst:.3f} ms per frame (AO passes {worst_ao:.3f} ms); three frames in flight {worst_ring:.3f} ms per frame")
