"""Deterministic closed-interior OFF scene: the labelled STAND-IN for the
reference's missing `meshes/sibenik.off` (the asset is not in
the reference checkout and there is no network).

A nave seen from inside: floor, walls, barrel vault, two rows of fluted columns
with bases and capitals, transverse arches and an apse of spheres, tessellated
to ~75 k triangles.  The camera of the renderer sits at (0, 0, 2) looking down
-z, i.e. inside the room, so every primary ray hits something (unlike bunny,
where 44 % miss) and AO rays see close occluders everywhere.

    python tools/make_interior_mesh.py [out.off]
    python tools/make_interior_mesh.py --hard [out.off]     the harder variant (build_hard): huge triangles beside fine
                                                            ornament, slivers, density varying 100 x
"""
from __future__ import annotations

import math
import os
import sys

import numpy as np


class Builder:
    def __init__(self):
        self.verts = []
        self.faces = []

    def grid(self, origin, du, dv, nu, nv, flip=False):
        """Tessellated parallelogram origin + s*du + t*dv, s,t in [0,1]."""
        base = len(self.verts)
        o, du, dv = (np.asarray(a, dtype=np.float64) for a in (origin, du, dv))
        for j in range(nv + 1):
            for i in range(nu + 1):
                self.verts.append(o + du * (i / nu) + dv * (j / nv))
        for j in range(nv):
            for i in range(nu):
                a = base + j * (nu + 1) + i
                b, c, d = a + 1, a + nu + 1, a + nu + 2
                if flip:
                    self.faces += [(a, c, b), (b, c, d)]
                else:
                    self.faces += [(a, b, c), (b, d, c)]

    def revolve(self, center, profile, segments, flute=0.0, flutes=0):
        """Surface of revolution around the y axis through `center`; profile =
        [(radius, y)], optional sinusoidal fluting of the radius."""
        base = len(self.verts)
        cx, cy, cz = center
        for (r, y) in profile:
            for s in range(segments):
                ang = 2 * math.pi * s / segments
                rr = r * (1.0 + flute * math.cos(flutes * ang)) if flutes else r
                self.verts.append(np.array([cx + rr * math.cos(ang), cy + y, cz + rr * math.sin(ang)]))
        for k in range(len(profile) - 1):
            for s in range(segments):
                a = base + k * segments + s
                b = base + k * segments + (s + 1) % segments
                c, d = a + segments, b + segments
                self.faces += [(a, c, b), (b, c, d)]

    def arch(self, z, radius, y0, thickness, depth, segments):
        """Half-ring (transverse arch) in the plane z, spanning x = -radius..radius."""
        for (r, flip) in ((radius, False), (radius - thickness, True)):
            base = len(self.verts)
            for k in range(segments + 1):
                ang = math.pi * k / segments
                for dz in (0.0, depth):
                    self.verts.append(np.array([r * math.cos(ang), y0 + r * math.sin(ang), z - dz]))
            for k in range(segments):
                a = base + 2 * k
                b, c, d = a + 1, a + 2, a + 3
                self.faces += ([(a, c, b), (b, c, d)] if flip else [(a, b, c), (b, d, c)])
        # front face of the ring
        base = len(self.verts)
        for k in range(segments + 1):
            ang = math.pi * k / segments
            for r in (radius, radius - thickness):
                self.verts.append(np.array([r * math.cos(ang), y0 + r * math.sin(ang), z]))
        for k in range(segments):
            a = base + 2 * k
            self.faces += [(a, a + 1, a + 2), (a + 1, a + 3, a + 2)]

    def sphere(self, center, radius, rings, segments):
        profile = [(radius * math.sin(math.pi * k / rings), -radius * math.cos(math.pi * k / rings)) for k in range(rings + 1)]
        profile[0] = (1e-4 * radius, profile[0][1])
        profile[-1] = (1e-4 * radius, profile[-1][1])
        self.revolve(center, profile, segments)


def build():
    b = Builder()
    x0, x1, y0, y1, z0, z1 = -4.0, 4.0, -2.0, 2.2, -14.0, 3.0
    # floor, walls, back/front
    b.grid((x0, y0, z1), (x1 - x0, 0, 0), (0, 0, z0 - z1), 40, 80)
    b.grid((x0, y0, z0), (0, 0, z1 - z0), (0, y1 - y0, 0), 80, 24)
    b.grid((x1, y0, z1), (0, 0, z0 - z1), (0, y1 - y0, 0), 80, 24)
    b.grid((x0, y0, z0), (x1 - x0, 0, 0), (0, y1 - y0 + 4.0, 0), 48, 40, flip=True)
    b.grid((x1, y0, z1), (x0 - x1, 0, 0), (0, y1 - y0 + 4.0, 0), 24, 20, flip=True)
    # barrel vault: half cylinder of radius 4 on top of the walls
    base = len(b.verts)
    segs, slices = 36, 80
    for j in range(slices + 1):
        z = z1 + (z0 - z1) * j / slices
        for k in range(segs + 1):
            ang = math.pi * k / segs
            b.verts.append(np.array([4.0 * math.cos(ang), y1 + 4.0 * math.sin(ang), z]))
    for j in range(slices):
        for k in range(segs):
            a = base + j * (segs + 1) + k
            bb, c, d = a + 1, a + segs + 1, a + segs + 2
            b.faces += [(a, c, bb), (bb, c, d)]
    # two rows of fluted columns with bases and capitals
    shaft = [(0.34, 0.0), (0.34, 0.12), (0.27, 0.2)] + [(0.25 - 0.0012 * k, 0.2 + 0.1333 * k) for k in range(1, 25)] + \
            [(0.3, 3.45), (0.36, 3.6), (0.36, 3.75)]
    for zc in np.linspace(0.5, -12.5, 9):
        for xc in (-2.3, 2.3):
            b.revolve((xc, y0, float(zc)), shaft, 32, flute=0.04, flutes=8)
    # transverse arches between the columns
    for zc in np.linspace(0.5, -12.5, 9):
        b.arch(float(zc) + 0.15, 3.95, y1, 0.3, 0.3, 48)
    # apse: a few spheres at the far end and a font near the camera axis
    for (c, r) in (((0.0, -0.9, -12.6), 1.0), ((-1.6, -1.4, -11.8), 0.55), ((1.6, -1.4, -11.8), 0.55), ((0.0, -1.55, -3.0), 0.45)):
        b.sphere(c, r, 40, 48)
    return np.array(b.verts), np.array(b.faces)


def box(b, lo, hi):
    """Axis-aligned box as 12 triangles (two per face, outward)."""
    (x0, y0, z0), (x1, y1, z1) = lo, hi
    b.grid((x0, y0, z1), (x1 - x0, 0, 0), (0, y1 - y0, 0), 1, 1)             # front  (+z)
    b.grid((x1, y0, z0), (x0 - x1, 0, 0), (0, y1 - y0, 0), 1, 1)             # back   (-z)
    b.grid((x0, y0, z0), (0, 0, z1 - z0), (0, y1 - y0, 0), 1, 1)             # left   (-x)
    b.grid((x1, y0, z1), (0, 0, z0 - z1), (0, y1 - y0, 0), 1, 1)             # right  (+x)
    b.grid((x0, y1, z1), (x1 - x0, 0, 0), (0, 0, z0 - z1), 1, 1)             # top    (+y)
    b.grid((x0, y0, z0), (x1 - x0, 0, 0), (0, 0, z1 - z0), 1, 1)             # bottom (-y)


def build_hard():
    """The same nave as build(), made the way a modelled interior (Sibenik) is rather than the way a generator likes it:
    * floor, walls, end walls and the vault are A HANDFUL OF HUGE TRIANGLES (2 per wall, 8 x 17 units each; a vault of 12
      flat panels) -- their one-triangle leaves span the whole room, which is what a midpoint-split BVH
      (reference src/bvh.cc:59-94) copes worst with;
    * long thin triangles: window mullions and transoms on both walls (bars 3 cm wide, 3 m tall), the steps of a staircase
      before the apse (8 m wide, 6 cm deep), ribs along the vault;
    * the ornament is tessellated 100 x more densely than the shell it stands in: finely fluted columns with stacked
      capitals, chandeliers of small spheres, the apse's spheres -- ~97 % of the triangles in ~5 % of the surface.
    ~75 k triangles like the other stand-in; every primary ray hits something."""
    b = Builder()
    x0, x1, y0, y1, z0, z1 = -4.0, 4.0, -2.0, 2.2, -14.0, 3.0
    # the shell: two triangles per surface
    b.grid((x0, y0, z1), (x1 - x0, 0, 0), (0, 0, z0 - z1), 1, 1)
    b.grid((x0, y0, z0), (0, 0, z1 - z0), (0, y1 - y0, 0), 1, 1)
    b.grid((x1, y0, z1), (0, 0, z0 - z1), (0, y1 - y0, 0), 1, 1)
    b.grid((x0, y0, z0), (x1 - x0, 0, 0), (0, y1 - y0 + 4.0, 0), 1, 1, flip=True)
    b.grid((x1, y0, z1), (x0 - x1, 0, 0), (0, y1 - y0 + 4.0, 0), 1, 1, flip=True)
    # the vault: 12 flat panels running the whole length
    panels = 12
    for k in range(panels):
        a0, a1 = math.pi * k / panels, math.pi * (k + 1) / panels
        p0 = (4.0 * math.cos(a0), y1 + 4.0 * math.sin(a0), z1)
        p1 = (4.0 * math.cos(a1), y1 + 4.0 * math.sin(a1), z1)
        b.grid(p0, (p1[0] - p0[0], p1[1] - p0[1], 0.0), (0.0, 0.0, z0 - z1), 1, 1, flip=True)
    # ribs along the vault: thin boxes, 17 m long
    for k in range(1, panels):
        a = math.pi * k / panels
        cx, cy = 3.93 * math.cos(a), y1 + 3.93 * math.sin(a)
        box(b, (cx - 0.03, cy - 0.03, z0), (cx + 0.03, cy + 0.03, z1))
    # windows: a lattice of mullions (vertical) and transoms (horizontal) 3 cm wide, standing 2 cm off both walls
    for side, xw in ((-1, x0 + 0.02), (1, x1 - 0.05)):
        for zc in np.linspace(0.0, -12.0, 7):
            for m in range(6):
                zm = float(zc) - 0.6 + 0.24 * m
                box(b, (xw, -0.6, zm - 0.015), (xw + 0.03, 2.0, zm + 0.015))
            for t in range(5):
                yt = -0.6 + 0.65 * t
                box(b, (xw, yt - 0.015, float(zc) - 0.62), (xw + 0.03, yt + 0.015, float(zc) + 0.62))
    # a staircase before the apse: twelve steps the whole width of the nave, 6 cm deep and 4 cm high each
    for k in range(12):
        box(b, (x0 + 0.3, y0 + 0.04 * k, -10.6 - 0.06 * k - 0.06), (x1 - 0.3, y0 + 0.04 * (k + 1), -10.6 - 0.06 * k))
    # two rows of finely fluted columns with stacked capitals (the dense part)
    shaft = [(0.36, 0.0), (0.36, 0.1), (0.3, 0.14), (0.3, 0.2), (0.27, 0.24)] + \
            [(0.25 - 0.0006 * k, 0.24 + 0.0643 * k) for k in range(1, 50)] + \
            [(0.27, 3.42), (0.31, 3.46), (0.29, 3.52), (0.35, 3.58), (0.33, 3.64), (0.38, 3.7), (0.38, 3.78)]
    for zc in np.linspace(0.5, -12.5, 9):
        for xc in (-2.3, 2.3):
            b.revolve((xc, y0, float(zc)), shaft, 24, flute=0.05, flutes=12)
    # chandeliers: rings of small spheres hanging over the nave's axis
    for zc in (-2.0, -6.0, -10.0):
        for ring, (radius, count, drop) in enumerate(((0.5, 10, 0.0), (0.3, 6, -0.25))):
            for i in range(count):
                ang = 2.0 * math.pi * i / count + 0.3 * ring
                b.sphere((radius * math.cos(ang), 1.6 + drop, zc + radius * math.sin(ang)), 0.07, 8, 10)
        box(b, (-0.01, 1.7, zc - 0.01), (0.01, y1 + 4.0, zc + 0.01))  # the chain: 4.5 m long, 2 cm thick
    # apse: spheres at the far end and a font near the camera axis
    for (c, r) in (((0.0, -0.9, -12.6), 1.0), ((-1.6, -1.4, -11.8), 0.55), ((1.6, -1.4, -11.8), 0.55), ((0.0, -1.55, -3.0), 0.45)):
        b.sphere(c, r, 36, 44)
    return np.array(b.verts), np.array(b.faces)


def write_interior_mesh(path: str, hard: bool = False) -> None:
    v, f = build_hard() if hard else build()
    with open(path, "w") as out:
        out.write("OFF\n%d %d 0\n" % (len(v), len(f)))
        for p in v:
            out.write("%.7f %.7f %.7f\n" % (p[0], p[1], p[2]))
        for t in f:
            out.write("3 %d %d %d\n" % (t[0], t[1], t[2]))


if __name__ == "__main__":
    hard = "--hard" in sys.argv
    args = [a for a in sys.argv[1:] if a != "--hard"]
    name = "interior_hard.off" if hard else "interior_standin.off"
    target = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "meshes", name)
    write_interior_mesh(target, hard)
    v, f = build_hard() if hard else build()
    print(target, len(v), "vertices", len(f), "triangles")
