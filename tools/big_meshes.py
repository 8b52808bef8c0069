"""Deterministic generators for scenes the two test meshes cannot stand for (all in memory: arrays for
``rt.Scene.from_arrays``; nothing of this size is written to disk or committed):

* ``terrain(n)``      -- an n x n height field that fills the view of the fixed camera at (0, 0, 2): 2 n^2 triangles.
                         n = 1000: 2 M triangles (packed scene ~0.6 GB >> the 32 MB of L2), n = 3200 at scale 4: 20.5 M
                         triangles (~6 GB >> the 256 MB Infinity Cache).  Every primary ray hits, every hit casts its AO rays into
                         neighbouring bumps.  This is where the memory roofline of north_star means something.
* ``slivers()``       -- needles (aspect 1 : 10^6), a triangle that spans 2 * 10^5 units, points at +-10^5 ... 10^6: the
                         ``origin_limit`` / ``RECIPROCAL_LIMIT`` / ``walk_scale_usable`` paths of the walk.
* ``coplanar_stack(n, copies)`` -- a small terrain whose every triangle exists ``copies`` times (shuffled): closest hits
                         tie in distance everywhere and must go to the lowest reference leaf, at scale.
"""
from __future__ import annotations

import numpy as np


def _height(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Rolling bumps in five octaves, amplitude ~0.25, pushed back so the surface stays in front of the camera."""
    z = np.zeros_like(x)
    amp, freq = 0.11, 2.3
    for k in range(5):
        z += amp * np.sin(freq * x + 1.7 * k) * np.cos(freq * 1.13 * y - 0.9 * k)
        amp *= 0.5
        freq *= 2.17
    return z - 0.35


def terrain(n: int, half_width: float = 1.25, half_height: float = 0.75, scale: float = 1.0):
    """(vertices float32 [(n+1)^2, 3], faces uint32 [2 n^2, 3]); rows of quads, each split along alternating diagonals.

    `scale` blows the field up about the camera at (0, 0, 2): the picture's outline stays, the triangles grow.  It has to
    for the largest fields: the reference's triangle test rejects a hit when |dot(cross(u, v), direction)| < 1e-6
    (src/intersect_kernel.cl:80) -- an UNNORMALISED normal, so triangles of less than ~5e-7 units of area are invisible to
    it whatever the implementation (a 3200 x 3200 field at scale 1: not one hit, on the oracle and on the GPU alike)."""
    xs = np.linspace(-half_width, half_width, n + 1, dtype=np.float64)
    ys = np.linspace(-half_height, half_height, n + 1, dtype=np.float64)
    x, y = np.meshgrid(xs, ys)
    v = np.stack([x, y, _height(x, y)], axis=-1).reshape(-1, 3)
    camera = np.array([0.0, 0.0, 2.0])
    v = (camera + scale * (v - camera)).astype(np.float32)
    i, j = np.meshgrid(np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32))
    a = (j * (n + 1) + i).reshape(-1)
    b, c, d = a + 1, a + (n + 1), a + (n + 2)
    flip = (((i + j) & 1) == 1).reshape(-1)
    t1 = np.where(flip[:, None], np.stack([a, b, d], 1), np.stack([a, b, c], 1))
    t2 = np.where(flip[:, None], np.stack([a, d, c], 1), np.stack([b, d, c], 1))
    f = np.empty((2 * n * n, 3), dtype=np.uint32)
    f[0::2] = t1
    f[1::2] = t2
    return v, f


def slivers(seed: int = 20261004):
    """~6 k triangles: a 40 x 40 terrain patch in view, needles across it, huge and far-away triangles around it."""
    rng = np.random.default_rng(seed)
    v, f = terrain(40, 0.9, 0.5)
    verts, faces = [v], [f]
    base = v.shape[0]

    def add(tri):
        nonlocal base
        verts.append(np.asarray(tri, dtype=np.float32).reshape(3, 3))
        faces.append(np.array([[base, base + 1, base + 2]], dtype=np.uint32))
        base += 3

    for _ in range(400):  # needles: two vertices 1e-6 apart, the third up to 4 units away, hovering over the patch
        p = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.45, 0.45), rng.uniform(-0.2, 0.3)])
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        w = np.cross(d, rng.normal(size=3))
        w /= np.linalg.norm(w)
        add([p, p + w * 10.0 ** rng.uniform(-6, -3), p + d * 10.0 ** rng.uniform(-1, 0.6)])
    add([[-1e5, -0.8, -1e5], [1e5, -0.8, -1e5], [0.0, -0.8, 1e5]])  # a floor 2e5 units across
    for e in (1e3, 1e4, 1e5, 1e6):  # small triangles far out on every axis: the root box reaches 1e6
        for axis in range(3):
            for sign in (-1.0, 1.0):
                c = np.zeros(3)
                c[axis] = sign * e
                add([c, c + np.array([1.0, 0.0, 0.5]), c + np.array([0.0, 1.0, 0.5])])
    for _ in range(200):  # flat triangles exactly in axis planes (zero-thickness boxes), inside the view
        c = np.array([rng.uniform(-0.8, 0.8), rng.uniform(-0.45, 0.45), rng.uniform(-0.1, 0.4)])
        axis = int(rng.integers(3))
        a, b = rng.normal(size=3) * 0.05, rng.normal(size=3) * 0.05
        a[axis] = b[axis] = 0.0
        add([c, c + a, c + b])
    return np.concatenate(verts), np.concatenate(faces)


def coplanar_stack(n: int = 160, copies: int = 3, seed: int = 7):
    """Every triangle of an n x n terrain `copies` times over the same vertices, the face list shuffled."""
    v, f = terrain(n, 1.1, 0.65)
    f = np.concatenate([f] * copies)
    np.random.default_rng(seed).shuffle(f, axis=0)
    return v, np.ascontiguousarray(f)
