"""Test/bench scene files.

* ``bunny_path()``   -- meshes/bunny.off, decompressed on first use from
  meshes/bunny.off.gz (the reference's scene asset, a DATA file: Stanford bunny
  + 2-triangle ground plane, 35 290 vertices / 70 570 triangles).
* ``interior_path()`` -- meshes/interior_standin.off, the labelled stand-in for
  the reference's missing ``sibenik.off``: generated
  deterministically by tools/make_interior_mesh.py.
"""
from __future__ import annotations

import gzip
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MESH_DIR = os.path.join(ROOT, "meshes")


def bunny_path() -> str:
    dst = os.path.join(MESH_DIR, "bunny.off")
    if not os.path.exists(dst):
        tmp = dst + f".tmp{os.getpid()}"
        with gzip.open(dst + ".gz", "rb") as src, open(tmp, "wb") as out:
            shutil.copyfileobj(src, out)
        os.replace(tmp, dst)
    return dst


def interior_path() -> str:
    dst = os.path.join(MESH_DIR, "interior_standin.off")
    if not os.path.exists(dst):
        from tools.make_interior_mesh import write_interior_mesh

        tmp = dst + f".tmp{os.getpid()}"
        write_interior_mesh(tmp)
        os.replace(tmp, dst)
    return dst


def interior_hard_path() -> str:
    """meshes/interior_hard.off: the HARDER labelled stand-in (tools/make_interior_mesh.py, build_hard): the same nave with
    its shell as a handful of huge triangles, long thin ones (mullions, steps, ribs) and ornament 100 x denser."""
    dst = os.path.join(MESH_DIR, "interior_hard.off")
    if not os.path.exists(dst):
        from tools.make_interior_mesh import write_interior_mesh

        tmp = dst + f".tmp{os.getpid()}"
        write_interior_mesh(tmp, hard=True)
        os.replace(tmp, dst)
    return dst
