"""GPU box: measures the tiles' ambient-occlusion costs of bench workloads (rt_debug_measure_tile_costs, spatial-order
conditions: the cost-class order) and saves them with the claim order and the tile words to gpurun_out/tile_costs_<workload>.npz
for offline scheduling models (tools/analysis/schedule_model.py).   python3 tools/analysis/dump_tile_costs.py WORKLOAD ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, load_scene, workload_options  # noqa: E402

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for name in sys.argv[1:]:
    w = WORKLOADS[name]
    opt = workload_options(rt, w)
    scene = load_scene(rt, w).build_bvh(opt.bvh_method)
    host = rt.Host(opt, 0)
    host.expect_frames(1000)
    host.upload_scene(scene)
    for _ in range(3):
        host.render()
    host.measure_tile_costs(4, reorder=False)
    info = host.tile_order()
    n = int(np.sqrt(opt.n_super_samples))
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"tile_costs_{name}.npz"), order=info["order"], constants=info["constants"],
                        words=info["words"], costs=info["costs"] / 4.0, tiles_x=(opt.width * n + 7) // 8, rows=(opt.height * n + 7) // 8)
    print(name, "saved", int((info["words"] >> 8 != 0).sum()), "tiles with work", flush=True)
    host.close()
