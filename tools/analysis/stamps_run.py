"""Wave-time per phase of the ambient-occlusion pass (GPU box; a library built with -DOCRT_STAMPS:
    make -C opencl_raytracer_amd/csrc EXTRA_DEFS=-DOCRT_STAMPS OBJDIR=.../build_stamps LIBDIR=.../lib_stamps BINDIR=.../bin_stamps
    OCRT_LIB_DIR=lib_stamps python3 tools/analysis/stamps_run.py WORKLOAD ...
A ring of one host (its upload measures the tiles' costs and claims them by that); the library prints the table when the
statistics are asked for."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, load_scene, workload_options  # noqa: E402

for name in sys.argv[1:]:
    w = WORKLOADS[name]
    opt = workload_options(rt, w)
    scene = load_scene(rt, w).build_bvh(opt.bvh_method)  # (an OFF file or a generated height field)
    ring = rt.FrameRing(opt, scene, hosts=1)
    ring.set_graph_mode(False)
    ring.run(5)
    ring.drain()
    host = ring.host(0)
    print(name, "kernel ms", host.last_kernel_ms, "ao ms", host.last_ao_ms, flush=True)
    host.stats()
    ring.close()
