"""Probe (GPU box, A/B build): the AO pass alone with other claim sizes (directions per wave and claim; default rule: 7 =
a quarter of a tile) and with guided self-scheduling.   python3 tools/analysis/claim_probe.py"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import opencl_raytracer_amd as rt
    from bench import WORKLOADS, mesh_path, workload_options
    w = WORKLOADS[sys.argv[1]]; opt = workload_options(rt, w)
    scene = rt.Scene.load_off(mesh_path(w["mesh"])).build_bvh(opt.bvh_method)
    host = rt.Host(opt, 0); host.upload_scene(scene)
    for _ in range(5): host.render()
    host.reset_timers()
    for _ in range(20): host.render()
    print(f"{sys.argv[1]} {sys.argv[2]}: ao {host.total_ao_ms / host.kernel_launches:.4f} ms, frame {host.total_kernel_ms / host.kernel_launches:.4f} ms", flush=True)
else:
    settings = [{}] + [{"OCRT_COST_SHIFT": str(k)} for k in (1, 2, 3, 4, 6)]
    for rep in range(2):
        for w in ("bunny_1080p_ao", "bunny_600_defaults", "interior_1080p_ao", "interior_4k_ao", "bunny_1080p_s16"):
            for setting in settings:
                env = dict(os.environ, OCRT_LIB_DIR="lib_knobs", **setting)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), w, str(setting or "default")], env=env, capture_output=True, text=True)
                print(r.stdout.strip() or r.stderr[-300:], flush=True)
