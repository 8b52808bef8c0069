import os, sys
sys.path.insert(0, os.getcwd())
import opencl_raytracer_amd as rt
from bench import WORKLOADS, mesh_path, workload_options
w = WORKLOADS[sys.argv[1]]
opt = workload_options(rt, w)
scene = rt.Scene.load_off(mesh_path(w["mesh"])).build_bvh(opt.bvh_method)
host = rt.Host(opt, 0)
host.upload_scene(scene)
for _ in range(3):
    host.render()
host.reset_timers()
host.render()
print(sys.argv[1], "kernel ms", host.total_kernel_ms, "ao ms", host.total_ao_ms)
host.stats()
