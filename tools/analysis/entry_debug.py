"""Debug aid: renders one workload with the library in OCRT_LIB_DIR and writes the float image to argv[2] (.npy);
`compare a.npy b.npy` lists where two such images differ (sub-pixel, tile, values)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if sys.argv[1] == "compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = a.view(np.uint32) != b.view(np.uint32)
    ys, xs = np.nonzero(d)
    print("shape", a.shape, "differing sub-pixels", ys.size)
    tiles = sorted({(int(y) // 8, int(x) // 8) for y, x in zip(ys[:100000], xs[:100000])})
    print("tiles touched (first 40 of %d):" % len(tiles), tiles[:40])
    for y, x in list(zip(ys, xs))[:40]:
        print(int(y), int(x), float(a[y, x]), float(b[y, x]))
else:
    import opencl_raytracer_amd as rt
    from bench import WORKLOADS, load_scene, workload_options

    w = dict(WORKLOADS[sys.argv[1]])
    for kv in sys.argv[3:]:
        k, v = kv.split("=")
        w[k] = int(v)
    opt = workload_options(rt, w)
    scene = load_scene(rt, w).build_bvh(opt.bvh_method)
    host = rt.Host(opt, 0)
    host.upload_scene(scene)
    host.render()
    np.save(sys.argv[2], host.download())
    print(sys.argv[1], host.stats())
