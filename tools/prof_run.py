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
    # a ring of ONE host = the blocking frame of bench.py's `value`: its upload prepares what a stream of frames gets (walk
    # intervals, the tiles' measured costs and the claim order made from them, the form of the node loop); plain launches,
    # so that every kernel of a frame is a dispatch of its own in the profiler's output.  (The upload's own measuring and
    # calibrating frames are dispatches of the same kernels: ~20 more launches with the same instruction counts to within
    # the claim order's effect.)
    ring = rt.FrameRing(opt, scene, hosts=1)
    ring.set_graph_mode(False)
    host = ring.host(0)
    if args.share != 1:
        host.set_device_share(args.share)
    ring.run(args.frames)
    ring.drain()
    st = host.stats()
    rays = st["primary_rays"] + st["ao_rays"]
    timers = ring.timers()
    ms = timers["kernel_ms"] / max(1, timers["frames"])
    print(f"{args.workload}: {rays} rays, kernel {ms:.4f} ms avg over {timers['frames']} frames, {rays / ms / 1e3:.1f} Mrays/s")
    ring.close()


if __name__ == "__main__":
    main()
