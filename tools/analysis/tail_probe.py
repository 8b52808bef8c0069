"""When the waves of the ambient-occlusion pass end (GPU box; a library built with -DOCRT_TAIL:
    make -C opencl_raytracer_amd/csrc EXTRA_DEFS=-DOCRT_TAIL OBJDIR=.../build_tail LIBDIR=.../lib_tail BINDIR=.../bin_tail
    OCRT_LIB_DIR=lib_tail OCRT_ALLOW_OLD_LIB=1 python3 tools/analysis/tail_probe.py [workload ...]
One frame at a time (a ring of one host, plain launches); the library prints the histogram of the waves' end times,
counted from the moment the pass could begin, when the statistics are asked for."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import opencl_raytracer_amd as rt
    from bench import WORKLOADS, load_scene, workload_options

    for name in sys.argv[1:] or ["bunny_1080p_ao"]:
        w = WORKLOADS[name]
        opt = workload_options(rt, w)
        scene = load_scene(rt, w).build_bvh(opt.bvh_method)
        ring = rt.FrameRing(opt, scene, hosts=1)  # (its upload measures the tiles' costs and orders them by that)
        ring.set_graph_mode(False)
        ring.run(10)
        ring.drain()
        for _ in range(3):
            ring.run(1)
            ring.drain()
            print(name, flush=True)
            sys.stderr.flush()
            ring.host(0).stats()
        ring.close()


if __name__ == "__main__":
    main()
