"""Probe (GPU box): does pacing the submissions of a ring help?  A frame is submitted no sooner than `beta` x the running
mean of the time per frame after the previous submission (frames that complete together otherwise start their successors
together, and the ring runs in lockstep bursts).   python3 tools/analysis/pacing_probe.py WORKLOAD [hosts]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opencl_raytracer_amd as rt
from bench import WORKLOADS, mesh_path, workload_options
name = sys.argv[1]; hosts = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = WORKLOADS[name]; opt = workload_options(rt, w)
scene = rt.Scene.load_off(mesh_path(w["mesh"])).build_bvh(opt.bvh_method)
ring = rt.FrameRing(opt, scene, hosts=hosts)
ring.run(30); ring.drain()
def stream(frames, beta):
    period, last_submit, done = None, 0.0, 0
    t0 = time.perf_counter()
    submitted = 0
    while done < frames:
        if submitted < frames and ring.in_flight < hosts:
            now = time.perf_counter()
            if period is not None and beta > 0:
                while now < last_submit + beta * period:
                    now = time.perf_counter()
            ring.submit(); submitted += 1; last_submit = now
        if ring.in_flight > max(1, hosts - 1) - 0 or submitted == frames:
            ring.collect_info(); done += 1
            elapsed = time.perf_counter() - t0
            if done >= 8:
                period = elapsed / done
    return (time.perf_counter() - t0) / frames * 1e3
for rep in range(2):
    t0 = time.perf_counter(); ring.run(200); ring.drain(); base = (time.perf_counter() - t0) / 200 * 1e3
    line = [f"ring.run {base:.4f}"]
    for beta in (0.0, 0.3, 0.5, 0.7, 0.85, 0.95):
        line.append(f"beta {beta}: {stream(200, beta):.4f}")
    print(name, hosts, "hosts:", "  ".join(line), flush=True)
ring.close()
