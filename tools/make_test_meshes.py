"""Deterministic small OFF meshes for the parity tests (tests/golden/meshes/).

They are inputs, not reference material: each one is built to exercise an edge
of the reference's arithmetic:

* blob.off   perturbed sphere + ground quad: generic closest-hit / AO / normals.
* ties.off   coincident triangles with opposite winding and separate vertices
             (closest-hit ties must resolve to the lowest leaf index), geometry
             whose box bounds are exactly 0 on x and y (rays with a zero
             direction component hit the inf*0 = NaN slab path on odd image
             widths), slivers, a zero-area face, an unreferenced vertex.
* single.off one triangle (a one-node BVH).

Run:  python tools/make_test_meshes.py
"""
from __future__ import annotations

import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "meshes")


def write_off(path, verts, faces):
    with open(path, "w") as f:
        f.write("OFF\n%d %d 0\n" % (len(verts), len(faces)))
        for v in verts:
            f.write("%.7f %.7f %.7f\n" % tuple(v))
        for t in faces:
            f.write("3 %d %d %d\n" % tuple(t))


def blob(rings=20, segs=24, seed=7):
    rng = np.random.default_rng(seed)
    verts, faces = [], []
    for i in range(rings + 1):
        th = np.pi * i / rings
        for j in range(segs):
            ph = 2 * np.pi * j / segs
            r = 0.55 * (1.0 + 0.12 * np.sin(3 * th) * np.cos(2 * ph)) + 0.01 * rng.standard_normal()
            verts.append((r * np.sin(th) * np.cos(ph), r * np.cos(th) + 0.05, r * np.sin(th) * np.sin(ph) - 0.2))
    for i in range(rings):
        for j in range(segs):
            a = i * segs + j
            b = i * segs + (j + 1) % segs
            c = (i + 1) * segs + j
            d = (i + 1) * segs + (j + 1) % segs
            faces.append((a, c, b))
            faces.append((b, c, d))
    base = len(verts)
    verts += [(-3, -0.62, -3), (3, -0.62, -3), (3, -0.62, 3), (-3, -0.62, 3)]
    faces += [(base, base + 2, base + 1), (base, base + 3, base + 2)]
    return np.array(verts), np.array(faces)


def ties():
    verts, faces = [], []

    def tri(a, b, c):
        base = len(verts)
        verts.extend([a, b, c])
        faces.append((base, base + 1, base + 2))

    # coincident pair, opposite winding, separate vertices -> different normals
    tri((-0.6, -0.4, -1.0), (0.6, -0.4, -1.0), (0.0, 0.7, -1.0))
    tri((-0.6, -0.4, -1.0), (0.0, 0.7, -1.0), (0.6, -0.4, -1.0))
    # the same again shifted, in the other file order
    tri((0.2, -0.9, -0.5), (0.2, 0.1, -0.5), (1.0, -0.9, -0.5))
    tri((0.2, -0.9, -0.5), (1.0, -0.9, -0.5), (0.2, 0.1, -0.5))
    # box bounds exactly 0 in x / y: quads touching the x = 0 and y = 0 planes
    tri((0.0, -1.0, -2.0), (0.0, 1.0, -2.0), (-1.5, 0.0, -2.0))
    tri((0.0, 0.0, -3.0), (1.5, 0.0, -3.0), (0.0, 1.2, -3.0))
    tri((-1.0, 0.0, -1.5), (0.0, 0.0, -1.5), (-0.5, -0.8, -1.5))
    # slivers and near-degenerate faces
    tri((-1.2, 0.9, -1.2), (1.2, 0.9000001, -1.2), (0.0, 0.9000002, -1.2))
    tri((0.9, -0.2, -0.8), (0.9000001, 0.6, -0.8), (0.9000002, 0.2, -0.80001))
    # zero-area face (normal length 0)
    tri((0.3, 0.3, -0.7), (0.3, 0.3, -0.7), (0.5, 0.5, -0.7))
    # plane seen edge-on (parallel to the central ray)
    tri((0.0, -0.5, 0.5), (0.0, 0.5, 0.5), (0.0, 0.0, -4.0))
    # backdrop so AO rays find occluders
    tri((-2.0, -1.0, -4.0), (2.0, -1.0, -4.0), (0.0, 2.0, -4.0))
    tri((-2.0, -1.0, -3.9), (0.0, 2.0, -3.9), (2.0, -1.0, -3.9))
    verts.append((5.0, 5.0, 5.0))  # unreferenced vertex -> zero normal
    return np.array(verts), np.array(faces)


def single():
    return np.array([(-0.5, -0.5, 0.0), (0.5, -0.5, 0.0), (0.0, 0.5, 0.0)]), np.array([(0, 1, 2)])


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fn in (("blob", blob), ("ties", ties), ("single", single)):
        v, f = fn()
        write_off(os.path.join(OUT, name + ".off"), v, f)
        print(name, len(v), "vertices", len(f), "faces")


if __name__ == "__main__":
    main()
