"""Probe (GPU box): ms per frame of a 1/N share of a frame with H renderers taking frames in turn on the one GPU, for the
HIP runtime's hardware-queue limit the process was started with (GPU_MAX_HW_QUEUES: read by the runtime when it starts,
so one process per setting -- the caller's loop):

    for Q in "" 8 16; do GPU_MAX_HW_QUEUES=$Q python3 tools/analysis/queues_sweep.py bunny_1080p_ao 8 6,8,12; done
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if not os.environ.get("GPU_MAX_HW_QUEUES"):
    os.environ.pop("GPU_MAX_HW_QUEUES", None)
import opencl_raytracer_amd as rt
from bench import WORKLOADS, load_scene, workload_options
w = WORKLOADS[sys.argv[1]]
n = int(sys.argv[2])
opt = workload_options(rt, w)
scene = load_scene(rt, w).build_bvh(opt.bvh_method)
for hosts in [int(h) for h in sys.argv[3].split(",")]:
    worst = 0.0
    for rank in range(min(n, 3)):   # (three of the shares are enough to see the trend)
        ring = rt.FrameRing(opt, scene, 0, rank, n, hosts=hosts)
        dt = 1e9
        for frames in (3 * hosts, 600, 600):  # (warm-up, then the better of two runs of 600 frames)
            t0 = time.perf_counter()
            ring.run(frames)
            ring.drain()
            if frames > 3 * hosts:
                dt = min(dt, (time.perf_counter() - t0) / frames * 1e3)
        worst = max(worst, dt)
        ring.close()
    print(f"{sys.argv[1]} queues={os.environ.get('GPU_MAX_HW_QUEUES', 'default')} 1/{n} share, {hosts} hosts: {worst:.3f} ms per frame", flush=True)
