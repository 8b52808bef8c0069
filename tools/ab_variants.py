"""A/B runs of library builds (GPU box): for each build directory of the package (lib, lib_knobs, lib_<tag> made with
`make EXTRA_DEFS=... OBJDIR=... LIBDIR=... BINDIR=...`) and each workload, in fresh processes and interleaved twice so
that box drift hits every variant alike:

    one frame at a time (a ring of one host, plain launches: HIP events around the passes)
        ms per frame, of which primary pass + ordering step, ao_kernel
    a steady stream (a ring of four hosts -- OCRT_AB_HOSTS --, graph replay)      ms per frame by the wall clock

    python3 tools/ab_variants.py lib lib_x lib_knobs:OCRT_COST_SHIFT=3 -- bunny_1080p_ao interior_1080p_ao [--frames 120]
"""
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(workload, frames):
    import opencl_raytracer_amd as rt
    from bench import WORKLOADS, load_scene, mesh_path, workload_options

    w = WORKLOADS[workload]
    opt = workload_options(rt, w)
    scene = load_scene(rt, w).build_bvh(opt.bvh_method)
    ring = rt.FrameRing(opt, scene, hosts=1)
    ring.set_graph_mode(False)
    if "OCRT_AB_FRAME_FORM" in os.environ and hasattr(ring.host(0), "set_frame_form"):  # (a tool's own variable: "fused" / "separate")
        ring.host(0).set_frame_form(os.environ["OCRT_AB_FRAME_FORM"])
    ring.run(10)
    ring.drain()
    ring.reset_clock()
    first = ring.submit()
    ring.collect_info()
    ring.run(40)
    ring.drain()
    t = [ring.frame_times(f) for f in range(first + 1, first + 41)]
    total = statistics.median(x[3] - x[0] for x in t)
    primary = statistics.median((x[1] - x[0]) if x[1] else (x[3] - x[0]) for x in t)
    ao = statistics.median(x[2] - x[1] for x in t)
    st = ring.host(0).stats()
    rays = st["primary_rays"] + st["ao_rays"]
    ring.close()
    ring = rt.FrameRing(opt, scene, hosts=int(os.environ.get("OCRT_AB_HOSTS", "4")))  # (a tool's own variable: hosts of the stream's ring)
    if "OCRT_AB_PACING" in os.environ:  # (a tool's own variable: the ring's pacing factor, 0 = off)
        ring.set_pacing(float(os.environ["OCRT_AB_PACING"]))
    ring.run(30)
    ring.drain()
    walls = []
    for _ in range(3):
        t0 = time.perf_counter()
        ring.run(frames)
        ring.drain()
        walls.append((time.perf_counter() - t0) / frames * 1e3)
    ring.close()
    pipe = min(walls)
    print(f"{os.environ.get('OCRT_AB_LABEL', os.environ.get('OCRT_LIB_DIR', 'lib')):34s} {workload:20s} blocking {total:7.4f} ms (primary+order {primary:7.4f}, ao {ao:7.4f})  "
          f"stream {pipe:7.4f} ms = {rays / pipe / 1e3:8.1f} Mrays/s", flush=True)


def main():
    args = sys.argv[1:]
    if args and args[0] == "--one":
        return one(args[1], int(args[2]))
    frames = 120
    if "--frames" in args:
        frames = int(args[args.index("--frames") + 1])
        del args[args.index("--frames"):args.index("--frames") + 2]
    if "--reps" in args:
        reps_value = int(args[args.index("--reps") + 1])
        del args[args.index("--reps"):args.index("--reps") + 2]
        args += ["--reps", str(reps_value)]
    split = args.index("--")
    variants, workloads = args[:split], [a for a in args[split + 1:] if not a.startswith("--") and not a.isdigit()]
    reps = 2
    if "--reps" in args:
        reps = int(args[args.index("--reps") + 1])
    for rep in range(reps):
        for w in workloads:
            for v in variants:  # "lib_dir" or "lib_dir:KNOB=value,KNOB=value" (knobs: the A/B build lib_knobs reads them)
                lib_dir, _, knobs = v.partition(":")
                env = dict(os.environ, OCRT_LIB_DIR=lib_dir, OCRT_AB_LABEL=v, OCRT_ALLOW_OLD_LIB="1")
                for item in filter(None, knobs.split(",")):
                    key, _, value = item.partition("=")
                    env[key] = value
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", w, str(frames)], env=env, capture_output=True, text=True)
                out = [ln for ln in r.stdout.splitlines() if "blocking" in ln]
                print(out[0] if out else f"{v} {w}: FAILED {r.stderr[-300:]}", flush=True)


if __name__ == "__main__":
    main()
