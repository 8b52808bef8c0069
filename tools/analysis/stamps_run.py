import os, sys
sys.path.insert(0, "/root/repo")
import opencl_raytracer_amd as rt
from bench import WORKLOADS, mesh_path, workload_options
for name in sys.argv[1:]:
    w = WORKLOADS[name]; opt = workload_options(rt, w)
    scene = rt.Scene.load_off(mesh_path(w["mesh"])).build_bvh(opt.bvh_method)
    host = rt.Host(opt, 0); host.upload_scene(scene)
    for _ in range(3): host.render()
    print(name, "kernel ms", host.last_kernel_ms, "ao ms", host.last_ao_ms, flush=True)
    host.stats(); host.close()
