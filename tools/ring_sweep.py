"""Probe (GPU box): ms per frame of a 1/8, 1/4 and whole share of a frame with 1, 2, 3, 4 and 6 renderers taking frames in
turn on the one GPU (rt.FrameRing) -- how many frames to keep in flight per GPU for a given number of ranks.

    python3 tools/ring_sweep.py WORKLOAD
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import opencl_raytracer_amd as rt
from bench import WORKLOADS, load_scene, mesh_path, workload_options
w = WORKLOADS[sys.argv[1]]
opt = workload_options(rt, w)
scene = load_scene(rt, w).build_bvh(opt.bvh_method)
for n in (8, 4, 1):
    for hosts in (1, 2, 3, 4, 6):
        worst = 0.0
        for rank in range(n if n < 8 else 3):   # (three of the eight shares are enough to see the trend)
            ring = rt.FrameRing(opt, scene, 0, rank, n, hosts=hosts)
            for frames in (3 * hosts, 90):
                t0 = time.perf_counter()
                ring.run(frames)  # (one call into the library: submit, collect the oldest beyond hosts - 1 in flight)
                ring.drain()
                dt = (time.perf_counter() - t0) / frames * 1e3
            worst = max(worst, dt)
            ring.close()
        print(f"{sys.argv[1]} 1/{n} share, {hosts} hosts: {worst:.3f} ms per frame", flush=True)
