"""Probe (GPU box, A/B build): the AO pass alone with 3 ... 8 workgroups per CU (= waves per SIMD) -- how much of its time
is latency that more chains in flight would hide.   OCRT_LIB_DIR=lib_knobs python3 tools/analysis/occupancy_probe.py"""
import os, sys, statistics, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2:
    import opencl_raytracer_amd as rt
    from bench import WORKLOADS, mesh_path, workload_options
    w = WORKLOADS[sys.argv[1]]; opt = workload_options(rt, w)
    scene = rt.Scene.load_off(mesh_path(w["mesh"])).build_bvh(opt.bvh_method)
    host = rt.Host(opt, 0); host.upload_scene(scene)
    for _ in range(5): host.render()
    host.reset_timers()
    for _ in range(20): host.render()
    print(f"{sys.argv[1]} {sys.argv[2]} workgroups per CU: ao {host.total_ao_ms / host.kernel_launches:.4f} ms, frame {host.total_kernel_ms / host.kernel_launches:.4f} ms", flush=True)
else:
    for w in ("bunny_1080p_ao", "interior_1080p_ao"):
        for per_cu in (3, 4, 5, 6, 7, 8):
            env = dict(os.environ, OCRT_LIB_DIR="lib_knobs", OCRT_AO_BLOCKS=str(256 * per_cu))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), w, str(per_cu)], env=env, capture_output=True, text=True)
            print(r.stdout.strip() or r.stderr[-300:], flush=True)
