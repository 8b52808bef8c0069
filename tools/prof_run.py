"""Minimal driver for rocprofv3 runs: renders N frames of a bench workload through
the C ABI (no torch), e.g.

  rocprofv3 --kernel-trace --stats -d out -- python3 tools/prof_run.py --frames 5
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU -d out -- python3 tools/prof_run.py --frames 2
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, load_scene, mesh_path, workload_options  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="bunny_1080p_ao", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--bvh", default=None, choices=[None, "longest", "sah"])
    ap.add_argument("--share", type=int, default=1, help="tell the host that this many hosts share its GPU (the grid of a ring's hosts)")
    args = ap.parse_args()
    w = dict(WORKLOADS[args.workload])
    if args.bvh:
        w["bvh"] = args.bvh
    opt = workload_options(rt, w)
    scene = load_scene(rt, w).build_bvh(opt.bvh_method)
    host = rt.Host(opt, 0)
    host.upload_scene(scene)
    host.set_device_share(args.share)
    for _ in range(args.frames):
        host.render()
    st = host.stats()
    rays = st["primary_rays"] + st["ao_rays"]
    ms = host.total_kernel_ms / host.kernel_launches
    print(f"{args.workload}: {rays} rays, kernel {ms:.4f} ms avg over {host.kernel_launches}, {rays / ms / 1e3:.1f} Mrays/s")


if __name__ == "__main__":
    main()
