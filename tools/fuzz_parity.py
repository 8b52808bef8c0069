"""Randomised parity sweep (GPU box): renders random small configurations -- image size, supersampling,
AO rings / distance / angles, focal length, mesh, BVH strategy, scheduling knobs -- with the HIP path and with the
oracle and compares float images and statistics bit for bit.

    python tools/fuzz_parity.py [n_cases] [seed]
"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import opencl_raytracer_amd as rt  # noqa: E402
import orc  # noqa: E402

KNOBS = [{}, {}, {}, {"OCRT_BATCH_BELOW": "0"}, {"OCRT_BATCH_BELOW": "65"}, {"OCRT_AO_CLAIM_MAX": "1"}, {"OCRT_AO_CLAIM_MAX": "28"},
         {"OCRT_KEEP_TREE": "1"}, {"OCRT_FORCE_EXACT_WALK": "1"}, {"OCRT_AO_BLOCKS": "2"}, {"OCRT_NO_SHARED_WALK": "1"},
         {"OCRT_CONTRACT": "0.3"}, {"OCRT_CONTRACT": "2.0"}, {"OCRT_AO_GUIDE": "3"}, {"OCRT_NO_SCALED_WALK": "1"}]


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    oracle = orc.Oracle()
    meshes = {}
    for name in ("blob", "ties", "single"):
        for bvh in (0, 1):
            scene = rt.Scene.load_off(os.path.join(ROOT, "tests", "golden", "meshes", name + ".off")).build_bvh(bvh)
            meshes[name, bvh] = (scene, orc.SceneArrays.from_scene(scene))
    from tools.meshes import bunny_path, interior_path  # (the big scenes: longest-axis trees only)

    for name, path in (("bunny", bunny_path()), ("interior", interior_path())):
        scene = rt.Scene.load_off(path).build_bvh(0)
        meshes[name, 0] = (scene, orc.SceneArrays.from_scene(scene))
    bad = 0
    for case in range(n_cases):
        name, bvh = rng.choice(list(meshes))
        scene, arrays = meshes[name, bvh]
        knobs = rng.choice(KNOBS)
        for key in [k for k in os.environ if k.startswith("OCRT_")]:
            del os.environ[key]
        os.environ.update(knobs)
        opt = rt.Options.defaults(width=rng.randint(1, 150), height=rng.randint(1, 110),
                                  n_super_samples=rng.choice([1, 1, 2, 4, 5, 9, 16]), ao_num_samples=rng.choice([0, 1, 2, 3, 3, 4, 6]),
                                  ao_max_distance=rng.choice([0.05, 0.2, 0.2, 0.5, 3.0, 0.25, 1.0, 0.0371, 7.3e-7, 2.5e6, 17.0]), focal_length=rng.choice([0.7, 1.0, 1.0, 1.6]),
                                  enable_shading=rng.choice([1, 1, 0]))
        opt.ao_alpha_min = rng.choice([4, 4, 10, 0])
        opt.ao_alpha_max = rng.choice([90, 90, 60])
        host = rt.Host(opt, 0)
        host.upload_scene(scene)
        share = rng.choice([1, 1, 3, 6])  # (a host that is told it shares its GPU launches a smaller AO grid and keeps its claim size)
        host.set_device_share(share)
        host.render()
        got = host.download()
        st = host.stats()
        host.close()
        ref, counters, _ = oracle.render(orc.params_from_options(opt), arrays)
        same = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        stats_ok = all(st[k] == counters[k] for k in ("primary_rays", "primary_hits", "ao_rays", "ao_occluded"))
        if not (same and stats_ok):
            bad += 1
            print(f"MISMATCH case {case}: {name} bvh={bvh} {opt.width}x{opt.height} s{opt.n_super_samples} a{opt.ao_num_samples} "
                  f"d{opt.ao_max_distance} f{opt.focal_length} knobs={knobs} share={share}: image {same}, stats {stats_ok}", flush=True)
        elif case % 20 == 0:
            print(f"case {case} ok", flush=True)
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
