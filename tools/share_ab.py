"""Probe (GPU box): ms per frame of one rank's share of a frame (rank 0..3 of `ranks`) with `hosts` renderers taking
frames in turn on the one GPU (rt.FrameRing); reads OCRT_* knobs from the environment, for A/B runs.

    python3 tools/share_ab.py WORKLOAD RANKS HOSTS
"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import opencl_raytracer_amd as rt
from bench import WORKLOADS, load_scene, mesh_path, workload_options
w = WORKLOADS[sys.argv[1]]
n, hosts = int(sys.argv[2]), int(sys.argv[3])
opt = workload_options(rt, w)
scene = load_scene(rt, w).build_bvh(opt.bvh_method)
worst = 0.0
for rank in range(min(n, 4)):
    ring = rt.FrameRing(opt, scene, 0, rank, n, hosts=hosts)
    for frames in (3 * hosts, 120):
        t0 = time.perf_counter()
        ring.run(frames)  # (one call into the library: submit, collect the oldest beyond hosts - 1 in flight)
        ring.drain()
        dt = (time.perf_counter() - t0) / frames * 1e3
    worst = max(worst, dt)
    ring.close()
print(f"{sys.argv[1]} 1/{n} share, {hosts} hosts, {os.environ.get('OCRT_AO_CLAIM_MAX', 'rule')}: {worst:.3f} ms per frame", flush=True)
