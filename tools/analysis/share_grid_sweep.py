"""Probe (GPU box, A/B build: OCRT_LIB_DIR=.../lib_knobs): ms per frame of a 1/8 and a 1/4 share of the headline frame with
the ring's usual number of hosts, for several sizes of the persistent AO grid (OCRT_AO_BLOCKS; 0 = the library's rule)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import opencl_raytracer_amd as rt
from bench import WORKLOADS, load_scene, workload_options
w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p_ao"]
opt = workload_options(rt, w)
scene = load_scene(rt, w).build_bvh(opt.bvh_method)
for n, hosts in ((8, 6), (4, 3), (2, 3)):
    for blocks in (0, 128, 256, 512, 768, 1024):
        if blocks:
            os.environ["OCRT_AO_BLOCKS"] = str(blocks)
        else:
            os.environ.pop("OCRT_AO_BLOCKS", None)
        worst = 0.0
        for rank in range(min(n, 3)):
            ring = rt.FrameRing(opt, scene, 0, rank, n, hosts=hosts)
            for frames in (3 * hosts, 120):
                t0 = time.perf_counter()
                ring.run(frames)
                ring.drain()
                dt = (time.perf_counter() - t0) / frames * 1e3
            worst = max(worst, dt)
            ring.close()
        print(f"1/{n} share, {hosts} hosts, AO grid {blocks or 'default'}: {worst:.3f} ms per frame", flush=True)
