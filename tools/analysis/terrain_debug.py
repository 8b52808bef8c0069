"""Debug helper (GPU box): terrain scenes of several sizes against the oracle, with and without AO, for one library build
(OCRT_LIB_DIR).  Prints mismatch counts."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import opencl_raytracer_amd as rt
import orc
from tools.big_meshes import terrain

o = orc.Oracle()
for n in [int(a) for a in sys.argv[1:]] or [100, 300, 1000]:
    v, f = terrain(n)
    scene = rt.Scene.from_arrays(v, f).build_bvh(0)
    arrays = orc.SceneArrays.from_scene(scene)
    for ao in (0, 3):
        opt = rt.Options.defaults(width=192, height=108, n_super_samples=1, ao_num_samples=ao)
        ref, c, _ = o.render(orc.params_from_options(opt), arrays)
        host = rt.Host(opt, 0)
        host.upload_scene(scene)
        host.render()
        img = host.download()
        st = host.stats()
        bad = np.flatnonzero(img.view(np.uint32).ravel() != ref.view(np.uint32).ravel())
        print(f"n={n} tris={f.shape[0]} ao={ao}: {bad.size} words differ; hits {st['primary_hits']} vs {c['primary_hits']}, occluded {st['ao_occluded']} vs {c['ao_occluded']}", flush=True)
        if bad.size:
            k = bad[:5]
            print("   first:", [(int(i), float(img.ravel()[i]), float(ref.ravel()[i])) for i in k])
        host.close()
