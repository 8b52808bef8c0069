"""Probe (GPU box, A/B build: OCRT_LIB_DIR=.../lib_knobs): ms per frame of a 1/8 and a 1/4 share of the headline frame with
the ring's usual number of hosts, for several sizes of the persistent AO grid (OCRT_AO_BLOCKS; 0 = the library's rule).

    OCRT_LIB_DIR=lib_knobs python3 tools/analysis/share_grid_sweep.py [WORKLOAD [SHARES:HOSTS,... [GRID,...]]]     e.g. 8:12,8:6 0,256,512"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import opencl_raytracer_amd as rt
from bench import WORKLOADS, load_scene, workload_options
w = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "bunny_1080p_ao"]
opt = workload_options(rt, w)
scene = load_scene(rt, w).build_bvh(opt.bvh_method)
cases = [tuple(int(v) for v in c.split(":")) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [(8, 6), (4, 3), (2, 3)]
grids = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 128, 256, 512, 768, 1024]
for n, hosts in cases:
    for blocks in grids:
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
