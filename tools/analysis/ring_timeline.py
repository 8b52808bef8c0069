import sys, os, hashlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import opencl_raytracer_amd as rt
from bench import WORKLOADS, mesh_path, workload_options
w = WORKLOADS["bunny_1080p_ao"]; opt = workload_options(rt, w)
scene = rt.Scene.load_off(mesh_path(w["mesh"])).build_bvh(0)
ring = rt.FrameRing(opt, scene, hosts=3)
if len(sys.argv) > 1: ring.set_pacing(float(sys.argv[1]))
frames = []
for frame in range(9):
    if frame >= 3: frames.append(ring.collect())
    ring.submit()
while len(frames) < 9: frames.append(ring.collect())
for h in ring.hosts: h.download(); h.stats()
ring.reset_clock(); ring.keep_frame_times(True)
first = ring.submit(); ring.collect_info(); ring.run(30); ring.drain()
for f in range(first, first + 30):
    t = ring.frame_times(f)
    print(f, " ".join(f"{x:8.3f}" for x in t), f"  latency {t[3]-t[0]:.3f}")
