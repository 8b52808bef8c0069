"""Probe (GPU box): what one frame of a steady stream costs the CPU, read from the ring's own accounting
(rt_ring_cpu_times: seconds inside submit -- the launches --, inside collect waiting for the device, inside collect
otherwise), for a captured hipGraph per host against plain launches; on the headline frame (the GPU is the bottleneck:
the CPU numbers are what matters for N ranks sharing a host) and on a frame so small that the GPU is idle (64 x 64, no
AO: wall clock per step = the CPU side).  With `rccl` the ring also runs its RCCL exchange step in a world of one.

    python3 tools/step_overhead_probe.py [hosts] [rccl]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, mesh_path, workload_options  # noqa: E402

n_hosts = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with_rccl = "rccl" in sys.argv[2:]
scene = rt.Scene.load_off(mesh_path("bunny")).build_bvh(0)
tiny = rt.Options.defaults(width=64, height=64, n_super_samples=1, ao_num_samples=0, enable_ao=0)
cases = (("64x64, no AO (idle GPU)", tiny, 2000), ("headline frame", workload_options(rt, WORKLOADS["bunny_1080p_ao"]), 300))
for label, opt, steps in cases:
    for graph in (True, False):
        ring = rt.FrameRing(opt, scene, hosts=n_hosts)
        ring.set_graph_mode(graph)
        if with_rccl:
            ring.attach_rccl(rt.rccl_unique_id())
        ring.run(50)
        ring.drain()
        ring.reset_clock()
        t0 = time.perf_counter()
        ring.run(steps)
        ring.drain()
        wall = (time.perf_counter() - t0) / steps * 1e6
        c = ring.cpu_times()
        n = max(1, c["frames"])
        print(f"{label}, {n_hosts} hosts, {'graph replay' if graph else 'plain launches'}{', RCCL step' if with_rccl else ''}: "
              f"wall {wall:.1f} us per step; CPU submit {c['submit_s'] / n * 1e6:.1f} us, collect (not waiting) "
              f"{c['collect_s'] / n * 1e6:.1f} us, waiting {c['wait_s'] / n * 1e6:.1f} us", flush=True)
        ring.close()
