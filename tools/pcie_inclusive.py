import sys, time
sys.path.insert(0, '.')
import opencl_raytracer_amd as rt
from bench import WORKLOADS, mesh_path, workload_options
w = WORKLOADS["bunny_1080p_ao"]; opt = workload_options(rt, w)
t0 = time.perf_counter(); scene = rt.Scene.load_off(mesh_path("bunny")); t1 = time.perf_counter(); scene.build_bvh(0); t2 = time.perf_counter()
host = rt.Host(opt, 0); t3 = time.perf_counter(); host.upload_scene(scene); t4 = time.perf_counter()
for _ in range(3): host.render(); host.download_u8()
n = 20; t = time.perf_counter()
for _ in range(n): host.render(); img = host.download_u8()
dt = (time.perf_counter() - t) / n
st = host.stats(); rays = st["primary_rays"] + st["ao_rays"]
print(f"load {1e3*(t1-t0):.1f} ms, bvh {1e3*(t2-t1):.1f} ms, create {1e3*(t3-t2):.1f} ms, upload {1e3*(t4-t3):.1f} ms; render+resize+D2H(2 MB) {dt*1e3:.3f} ms/frame = {rays/dt/1e6:.0f} Mrays/s; kernels {host.last_kernel_ms:.3f} ms")
