"""Generates the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py [--full-sah]

What it does, per case of CASES below:
  1. loads the mesh and builds the BVH with the reference's own host objects
     (oracle/_ref/libref_host.so = reference src/{mesh,bvh,aabb,triangle}.cc);
  2. renders the float image with the reference's own kernel source compiled
     for x86-64 with that case's -D macro set (oracle/_ref/libref_kernel_*.so);
  3. box-filters with the reference's RayTracer::resize and forms the PGM file
     bytes exactly as reference src/render.cc:135-136 writes them;
  4. renders the same case with this repo's oracle (oracle/rt_oracle.c) and
     REFUSES to write anything unless float image and PGM are bit-identical;
  5. records md5(PGM file), sha256(float image), sha256 of every scene array and
     the oracle's traversal counters in tests/golden/golden.json.
Small inputs are additionally dumped whole (tests/golden/*.npz) so the tests can
compare arrays, not just digests.  Only data is written: no reference source.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orc  # noqa: E402
from tools.meshes import bunny_path, interior_hard_path, interior_path  # noqa: E402


def mesh_file(name: str) -> str:
    if name == "bunny":
        return bunny_path()
    if name == "interior":  # generated, labelled stand-in for the missing sibenik.off; its arrays are pinned under "scenes"
        return interior_path()
    if name == "interior_hard":  # ... and its harder variant (huge triangles beside fine ornament, slivers), pinned the same way
        return interior_hard_path()
    return os.path.join(HERE, "meshes", name + ".off")


def case(name, mesh, bvh="longest", width=64, height=64, ss=1, ao=3, aod=0.2, focal=1.0, shading=1, amin=4, amax=90,
         dump=False):
    return dict(name=name, mesh=mesh, bvh=bvh, width=width, height=height, ss=ss, ao=ao, aod=aod, focal=focal,
                shading=shading, amin=amin, amax=amax, dump=dump)


CASES = [
    # SURVEY.md 8c table (bunny, longest-axis tree)
    case("bunny_256_s1_a0", "bunny", width=256, height=256, ao=0),
    case("bunny_256_s1_a3", "bunny", width=256, height=256, ao=3),
    case("bunny_1080p_s1_a0", "bunny", width=1920, height=1080, ao=0),
    case("bunny_1080p_s1_a3", "bunny", width=1920, height=1080, ao=3),
    case("bunny_600_defaults", "bunny", width=600, height=600, ss=4, ao=3),
    # small float dump, odd sizes, non-square supersample counts, other constants
    case("bunny_64_s1_a3", "bunny", width=64, height=64, ao=3, dump=True),
    case("bunny_101x77_s9_a2", "bunny", width=101, height=77, ss=9, ao=2),
    case("bunny_50x40_s5_a1_f15", "bunny", width=50, height=40, ss=5, ao=1, aod=0.35, focal=1.5),
    case("bunny_96x54_s1_a4_alpha", "bunny", width=96, height=54, ao=4, amin=10, amax=60, aod=0.1),
    # small meshes, both trees
    case("blob_128x96_s4_a3", "blob", width=128, height=96, ss=4, ao=3, dump=True),
    case("blob_128x96_s4_a3_sah", "blob", bvh="sah", width=128, height=96, ss=4, ao=3),
    case("blob_80_s1_a5_noshade", "blob", width=80, height=80, ao=5, shading=0, aod=0.5),
    case("blob_33x17_s1_a0", "blob", width=33, height=17, ao=0),
    case("ties_33_s1_a3", "ties", width=33, height=33, ao=3, aod=1.0, dump=True),
    case("ties_33_s1_a3_sah", "ties", bvh="sah", width=33, height=33, ao=3, aod=1.0),
    case("ties_64_s4_a3", "ties", width=64, height=64, ss=4, ao=3, aod=1.0),
    case("ties_5x3_s1_a1", "ties", width=5, height=3, ao=1, aod=2.0, dump=True),
    case("single_32_s1_a3", "single", width=32, height=32, ao=3, dump=True),
    # BASELINE.json configs 3-5 at full size (digests only): interior stand-in 1080p and 4K, bunny with 64 samples
    # per pixel (regular 8x8 grid: 15360x8640 sub-pixels, 2.2 G rays)
    case("interior_1080p_s1_a3", "interior", width=1920, height=1080, ao=3),
    case("interior_4k_s1_a3", "interior", width=3840, height=2160, ao=3),
    case("bunny_1080p_s64_a3", "bunny", width=1920, height=1080, ss=64, ao=3),
    # the HARDER interior stand-in (tools/make_interior_mesh.py --hard: the shell as a handful of huge triangles, slivers,
    # ornament 100 x denser): a small frame the oracle re-renders in the tests, and configs 3 / 4 at full size
    case("interior_hard_160x90_s4_a3", "interior_hard", width=160, height=90, ss=4, ao=3),
    case("interior_hard_1080p_s1_a3", "interior_hard", width=1920, height=1080, ao=3),
    case("interior_hard_4k_s1_a3", "interior_hard", width=3840, height=2160, ao=3),
]


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class Opt:
    """Minimal stand-in for rt_options so orc.params_from_options works without the product."""

    def __init__(self, c):
        self.width, self.height = c["width"], c["height"]
        self.focal_length = c["focal"]
        self.n_super_samples = c["ss"]
        self.enable_shading = c["shading"]
        self.enable_ao = int(c["ao"] != 0)
        self.ao_max_distance = c["aod"]
        self.ao_num_samples = c["ao"]
        self.ao_method = 0
        self.ao_alpha_min, self.ao_alpha_max = c["amin"], c["amax"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full-sah", action="store_true", help="also run the reference's O(n^2) SAH on the full bunny (~6 min)")
    ap.add_argument("--only", default=None, help="comma-separated substrings of the case names to (re)generate")
    args = ap.parse_args()
    if not orc.reference_available():
        sys.exit("reference tree not present: goldens can only be regenerated in the build container")

    ref = orc.RefHost()
    oracle = orc.Oracle()
    out_path = os.path.join(HERE, "golden.json")
    golden = {"scenes": {}, "renders": {}, "ao_table_default_hex": None}
    if os.path.exists(out_path):
        golden.update(json.load(open(out_path)))

    # ---- scenes: mesh + BVH arrays from the reference's host code ----
    scenes = {}
    selected = [c for c in CASES if not args.only or any(part in c["name"] for part in args.only.split(","))]
    wanted = sorted({(c["mesh"], c["bvh"]) for c in selected} | ({("bunny", "sah")} if args.full_sah else set()))
    for mesh, bvh in wanted:
        t0 = time.time()
        h, v, n, f = ref.load(mesh_file(mesh))
        nodes, aabbs, tris, sorted_faces = ref.build_bvh(h, 0 if bvh == "longest" else 1)
        ref.free(h)
        scenes[(mesh, bvh)] = orc.SceneArrays(sorted_faces, nodes, aabbs, v, n)
        golden["scenes"][f"{mesh}/{bvh}"] = {
            "num_vertices": int(v.shape[0]), "num_faces": int(f.size // 3), "num_nodes": int(nodes.size),
            "vertices": sha(v), "vnormals": sha(n), "faces": sha(f), "nodes": sha(nodes), "aabbs": sha(aabbs),
            "triangles": sha(tris), "sorted_faces": sha(sorted_faces),
        }
        if mesh not in ("bunny", "interior", "interior_hard"):
            np.savez_compressed(os.path.join(HERE, f"scene_{mesh}_{bvh}.npz"), vertices=v, vnormals=n, faces=f,
                                nodes=nodes, aabbs=aabbs, triangles=tris, sorted_faces=sorted_faces)
        print(f"scene {mesh}/{bvh}: {nodes.size} nodes ({time.time() - t0:.1f}s)", flush=True)

    # ---- renders ----
    for c in CASES:
        if args.only and not any(part in c["name"] for part in args.only.split(",")):
            continue
        t0 = time.time()
        opt = Opt(c)
        p = orc.params_from_options(opt)
        scene = scenes[(c["mesh"], c["bvh"])]
        lib = orc.ref_kernel(p, c["ss"])
        ref_img, _ = orc.ref_render(lib, p, scene)
        ref_u8 = ref.resize(ref_img, c["width"], c["height"], c["ss"])
        pgm = f"P5 {c['width']} {c['height']} 255\n".encode() + ref_u8.tobytes()
        orc_img, counters, _ = oracle.render(p, scene)
        orc_u8 = oracle.resize(orc_img, c["width"], c["height"], c["ss"])
        same_f = np.array_equal(ref_img.view(np.uint32), orc_img.view(np.uint32))
        same_u = np.array_equal(ref_u8, orc_u8)
        if not (same_f and same_u):
            diff = int(np.count_nonzero(ref_img.view(np.uint32) != orc_img.view(np.uint32)))
            sys.exit(f"{c['name']}: oracle differs from the reference kernel ({diff} float words, u8 equal={same_u})")
        entry = dict(c)
        entry.update(total_width=int(p.width), total_height=int(p.height), pgm_md5=hashlib.md5(pgm).hexdigest(),
                     float_sha256=sha(ref_img), u8_sha256=sha(ref_u8), counters=counters)
        golden["renders"][c["name"]] = entry
        if c["dump"]:
            np.savez_compressed(os.path.join(HERE, f"render_{c['name']}.npz"), image=ref_img, u8=ref_u8)
        print(f"{c['name']}: md5 {entry['pgm_md5']} ({time.time() - t0:.1f}s)", flush=True)

    # ---- the default 28-direction UNIFORM table as hex floats (libm canary) ----
    p = orc.params_from_options(Opt(case("t", "bunny")))
    table = oracle.ao_table(p)
    golden["ao_table_default_hex"] = [[float(x).hex() for x in row] for row in table]

    with open(out_path, "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
