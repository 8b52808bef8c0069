"""Probe (GPU box, -DOCRT_PRIMARY_TICKS build in lib_pt): how long the primary pass's waves live, tile by tile.
    OCRT_LIB_DIR=lib_pt python3 tools/analysis/primary_ticks_probe.py [WORKLOAD ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import opencl_raytracer_amd as rt  # noqa: E402
from bench import WORKLOADS, load_scene, workload_options  # noqa: E402

for name in sys.argv[1:] or ["bunny_1080p_ao"]:
    w = WORKLOADS[name]
    opt = workload_options(rt, w)
    scene = load_scene(rt, w).build_bvh(opt.bvh_method)
    host = rt.Host(opt, 0, 0, 1)
    host.expect_frames(1000)
    host.upload_scene(scene)
    for _ in range(5):
        host.render()
    host.measure_tile_costs(1)
    t = host.tile_order()
    us = t["costs"].astype(np.float64) * 0.01  # 100 MHz ticks
    words = t["words"]
    stops = (words >> 8) & 0xFF
    host.render()
    print(f"{name}: {len(us)} tiles, the frame kernels {host.last_kernel_ms:.4f} ms of which any-hit pass {host.last_ao_ms:.4f}")
    print(f"  wave life, us: sum {us.sum():.0f} (/ 8192 slots = {us.sum() / 8192:.1f}), max {us.max():.1f}, "
          f"percentiles 50/90/99/99.9: {np.percentile(us, 50):.1f} / {np.percentile(us, 90):.1f} / {np.percentile(us, 99):.1f} / {np.percentile(us, 99.9):.1f}")
    for lo, hi in ((0, 1), (1, 2), (2, 8), (8, 16), (16, 32), (32, 65)):
        m = (stops >= lo) & (stops < hi)
        if m.any():
            print(f"  tiles with {lo}..{hi - 1} leaf stops (cost class): {m.sum():6d}, wave life mean {us[m].mean():6.1f} us, max {us[m].max():6.1f}")
    top = np.argsort(-us)[:10]
    print("  longest:", ", ".join(f"{us[i]:.0f} us ({stops[i]} stops)" for i in top))
    host.close()
