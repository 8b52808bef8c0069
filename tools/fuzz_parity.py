"""Randomised parity sweep (GPU box): renders random small configurations -- image size, supersampling,
AO rings / distance / angles, focal length, mesh, BVH strategy, hosts that share their GPU, scheduling knobs -- with the
HIP path and with the oracle and compares float images, 8-bit images and statistics bit for bit.

    python tools/fuzz_parity.py [n_cases] [seed]

The scheduling knobs are environment variables that only the A/B build of the library reads
(make EXTRA_DEFS=-DOCRT_DEBUG_KNOBS, opencl_raytracer_amd/lib_knobs): run as a tool, this file drives that build;
tests/test_hip_parity.py::test_fuzz_slice runs a seeded slice of the same cases, those without knobs on the product
library."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

KNOBS = [{}, {}, {}, {"OCRT_BATCH_BELOW": "0"}, {"OCRT_BATCH_BELOW": "65"}, {"OCRT_AO_CLAIM_MAX": "1"}, {"OCRT_AO_CLAIM_MAX": "28"},
         {"OCRT_KEEP_TREE": "1"}, {"OCRT_FORCE_EXACT_WALK": "1"}, {"OCRT_AO_BLOCKS": "2"}, {"OCRT_NO_SHARED_WALK": "1"},
         {"OCRT_CONTRACT": "0.3"}, {"OCRT_CONTRACT": "2.0"}, {"OCRT_AO_GUIDE": "3"}, {"OCRT_NO_SCALED_WALK": "1"},
         {"OCRT_STRIP_TILES": "4"}, {"OCRT_STRIP_TILES": "16"}, {"OCRT_STRIP_TILES": "32"}, {"OCRT_ENTRY_PER_TILE": "1"}]
SMALL_MESHES = [(name, bvh) for name in ("blob", "ties", "single") for bvh in (0, 1)]
# (the big scenes: longest-axis trees only; "terrain:<n>": a height field generated in memory, tools/big_meshes.py -- full
# tiles whose ambient-occlusion packets descend into dense geometry: what showed round 4's look-ahead race)
MESHES = SMALL_MESHES + [("bunny", 0), ("interior", 0), ("terrain:60", 0), ("terrain:150", 0), ("slivers", 0)]


def mesh_path(name):
    from tools.meshes import bunny_path, interior_path

    if name == "bunny":
        return bunny_path()
    if name == "interior":
        return interior_path()
    return os.path.join(ROOT, "tests", "golden", "meshes", name + ".off")


def draw_case(rng):
    """One random configuration (a plain dict; the same seed always draws the same sequence)."""
    name, bvh = rng.choice(MESHES)
    return dict(
        mesh=name, bvh=bvh, knobs=rng.choice(KNOBS),
        width=rng.randint(1, 150), height=rng.randint(1, 110), ss=rng.choice([1, 1, 2, 4, 5, 9, 16, 25, 36, 64]),
        ao=rng.choice([0, 1, 2, 3, 3, 4, 6]),
        aod=rng.choice([0.05, 0.2, 0.2, 0.5, 3.0, 0.25, 1.0, 0.0371, 7.3e-7, 2.5e6, 17.0]),
        focal=rng.choice([0.7, 1.0, 1.0, 1.6]), shading=rng.choice([1, 1, 0]), amin=rng.choice([4, 4, 10, 0]),
        amax=rng.choice([90, 90, 60]),
        share=rng.choice([1, 1, 3, 6]),  # (a host that is told it shares its GPU launches a smaller AO grid and keeps its claim size)
        ring=rng.choice([0, 0, 0, 2]),   # (0: one blocking host; n: a ring of n hosts, graph replay, three frames)
        lookahead=rng.choice([0, 1, 2]),  # (the form of the AO pass: without / with look-ahead loads / as calibrated or by default)
        announce=rng.choice([0, 1, 1]),  # (a blocking host: one-shot, or with a stream of frames announced -- its upload then prepares the walk intervals)
        form=rng.choice(["auto", "auto", "fused"]),  # (a blocking host's frame: two kernels, or both ray passes in one persistent launch -- kernels/frame.hip.h)
        measure=rng.choice([0, 0, 1]),  # (a blocking host: its tiles claimed by measured cost -- rt_debug_measure_tile_costs)
        quarters=rng.choice([None, None, 1, 4, 16, 40, 0]))  # (the primary pass casts tiles of this cost class or more in quarters -- rt_debug_set_primary_split; None: the default)


def run_case(rt, orc, oracle, scenes, case):
    """Renders `case` through the binding `rt` and through the oracle; returns (floats identical, statistics identical).
    `scenes`: cache {(mesh, bvh): (Scene, SceneArrays)} owned by the caller (per binding)."""
    key = (case["mesh"], case["bvh"])
    if key not in scenes:
        if case["mesh"].startswith("terrain:") or case["mesh"] == "slivers":
            from tools import big_meshes

            arrays = big_meshes.slivers() if case["mesh"] == "slivers" else big_meshes.terrain(int(case["mesh"].split(":")[1]))
            scene = rt.Scene.from_arrays(*arrays).build_bvh(case["bvh"])
        else:
            scene = rt.Scene.load_off(mesh_path(case["mesh"])).build_bvh(case["bvh"])
        scenes[key] = (scene, orc.SceneArrays.from_scene(scene))
    scene, arrays = scenes[key]
    for k in [k for k in os.environ if k.startswith("OCRT_") and k not in ("OCRT_LIB_DIR", "OCRT_DEVICE")]:
        del os.environ[k]
    os.environ.update(case["knobs"])
    try:
        opt = rt.Options.defaults(width=case["width"], height=case["height"], n_super_samples=case["ss"],
                                  ao_num_samples=case["ao"], ao_max_distance=case["aod"], focal_length=case["focal"],
                                  enable_shading=case["shading"])
        opt.ao_alpha_min, opt.ao_alpha_max = case["amin"], case["amax"]
        if case["ring"]:
            ring = rt.FrameRing(opt, None, hosts=case["ring"])
            if case["lookahead"] < 2 and hasattr(ring, "set_calibration"):
                ring.set_calibration(False)
                for slot in range(case["ring"]):
                    ring.host(slot).set_ao_prefetch(bool(case["lookahead"]))
            ring.upload_scene(scene)
            ring.run(3)
            ring.drain()
            host = ring.host(2 % case["ring"])
            got, got_u8, st = host.download(), ring.download_last(), host.stats()  # (the ring's frames are filtered into ITS band buffers)
            ring.close()
        else:
            host = rt.Host(opt, 0)
            if case.get("announce") and hasattr(host, "expect_frames"):
                host.expect_frames(1000)  # (an upload then prepares the walk intervals; without: the one-shot host)
            if case["lookahead"] < 2 and hasattr(host, "set_ao_prefetch"):
                host.set_ao_prefetch(bool(case["lookahead"]))
            host.upload_scene(scene)
            host.set_device_share(case["share"])
            if case.get("measure") and hasattr(host, "measure_tile_costs"):
                host.measure_tile_costs(1)
            if case.get("quarters") is not None and hasattr(host, "set_primary_split"):
                host.set_primary_split(case["quarters"])
            if case.get("form", "auto") != "auto" and hasattr(host, "set_frame_form"):
                host.set_frame_form(case["form"])
                if hasattr(host, "poison_hit_list"):
                    host.render()
                    host.poison_hit_list()  # (a fused frame must not read a record before it is handed over)
            host.render()
            got, got_u8, st = host.download(), host.download_u8(), host.stats()
            host.close()
    finally:
        for k in case["knobs"]:
            os.environ.pop(k, None)
    ref, counters, _ = oracle.render(orc.params_from_options(opt), arrays)
    # floats (incl. the ambient-occlusion factors the frame's last kernel wrote back) and the 8-bit image it box-filtered
    same = np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and \
        np.array_equal(got_u8, oracle.resize(ref, opt.width, opt.height, opt.n_super_samples))
    stats_ok = all(st[k] == counters[k] for k in ("primary_rays", "primary_hits", "ao_rays", "ao_occluded"))
    return same, stats_ok


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    os.environ.setdefault("OCRT_LIB_DIR", "lib_knobs")  # the build that reads the knobs
    import opencl_raytracer_amd as rt
    import orc

    oracle, scenes, bad = orc.Oracle(), {}, 0
    for index in range(n_cases):
        case = draw_case(rng)
        same, stats_ok = run_case(rt, orc, oracle, scenes, case)
        if not (same and stats_ok):
            bad += 1
            print(f"MISMATCH case {index}: {case}: image {same}, stats {stats_ok}", flush=True)
        elif index % 20 == 0:
            print(f"case {index} ok", flush=True)
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
