import os, sys
sys.path.insert(0, "/root/repo")
import opencl_raytracer_amd as rt
from bench import WORKLOADS, load_scene, workload_options
for name in sys.argv[1:]:
    w = WORKLOADS[name]; opt = workload_options(rt, w)
    scene = load_scene(rt, w).build_bvh(opt.bvh_method)  # (an OFF file or a generated height field)
    host = rt.Host(opt, 0); host.upload_scene(scene)
    for _ in range(3): host.render()
    print(name, "kernel ms", host.last_kernel_ms, "ao ms", host.last_ao_ms, flush=True)
    host.stats(); host.close()
