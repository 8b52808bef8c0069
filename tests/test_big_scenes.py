"""Scenes the two test meshes cannot stand for (tools/big_meshes.py), against the oracle at small resolutions, bit for bit:

* a 2 M-triangle height field -- the packed scene (~0.6 GB) is far beyond the 32 MB of L2, every primary ray hits, every
  AO ray walks a dense neighbourhood;
* needles, a triangle 2e5 units across, points out to +-1e6 and flat axis-aligned triangles: the `origin_limit`,
  `RECIPROCAL_LIMIT` and `walk_scale_usable` corners of the walk (with a tiny AO_MAX_DISTANCE the scaled node test must be
  refused for such an extent and the any-hit rays take the exact form);
* every triangle three times over: closest hits tie in distance everywhere, the lowest reference leaf must win.

(The reference's tests hold nothing of the kind -- it has none; these are this repo's own edge cases.)
"""
import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu


def _check(rt, oracle, vertices, faces, **options):
    import orc

    scene = rt.Scene.from_arrays(vertices, faces).build_bvh(options.pop("bvh_method", 0))
    arrays = orc.SceneArrays.from_scene(scene)
    opt = rt.Options.defaults(**options)
    ref_img, counters, _ = oracle.render(orc.params_from_options(opt), arrays)
    for hosts in (0, -1, 3):  # a host on its own (plain launches) -- one-shot, then with a stream announced --, a ring of three replaying its graphs
        if hosts <= 0:
            host = rt.Host(opt, 0)
            if hosts < 0:
                host.expect_frames(1000)  # (the walk intervals: entry_kernel)
            host.upload_scene(scene)
            host.render()
            img, u8, st = host.download(), host.download_u8(), host.stats()
            host.close()
        else:
            ring = rt.FrameRing(opt, scene, hosts=hosts)
            ring.run(4)
            ring.drain()
            u8 = ring.download_last()
            view = ring.host(0)
            img, st = view.download(), view.stats()
            ring.close()
        mism = int(np.count_nonzero(bits(img) != bits(ref_img)))
        assert mism == 0, f"{mism} float words differ from the oracle (hosts={hosts})"
        assert np.array_equal(u8, oracle.resize(ref_img, opt.width, opt.height, opt.n_super_samples))
        for key in ("primary_rays", "primary_hits", "ao_rays", "ao_occluded"):
            assert st[key] == counters[key], (key, hosts)
    return counters


@pytest.mark.parametrize("n", [40, 100, 300])
def test_height_fields(rt, oracle, n):
    """Small height fields: every tile is full and every ambient-occlusion packet descends into a dense neighbourhood --
    the walk's node loop runs long stretches without leaving.  (A look-ahead load still on its way when the loop was left
    corrupted a register here long before it showed on the bunny: profiles/r04_notes.md.)"""
    from tools.big_meshes import terrain

    v, f = terrain(n)
    _check(rt, oracle, v, f, width=192, height=108, n_super_samples=1, ao_num_samples=3)


def test_two_million_triangles(rt, oracle):
    from tools.big_meshes import terrain

    v, f = terrain(1000)
    assert f.shape[0] == 2_000_000
    c = _check(rt, oracle, v, f, width=192, height=108, n_super_samples=1, ao_num_samples=3)
    assert c["primary_hits"] == 192 * 108 and c["ao_occluded"] > 0  # the field fills the view


def test_wide_strips_of_a_scene_beyond_the_caches(rt, oracle):
    """A scene of several times the L2s in an image wide enough (>= 64 tiles across) gets strips of FOUR tiles instead of
    two in the PRODUCT build (DeviceRenderer::adopt: scene_beyond_caches) -- the tile <-> workgroup mapping, the claim
    orders and the block lists all depend on the strip width; the knob build's OCRT_STRIP_TILES is not what runs here."""
    from tools.big_meshes import terrain

    v, f = terrain(420)  # 352 800 triangles: ~120 MB on the device, beyond BIG_SCENE_BYTES (96 MB)
    c = _check(rt, oracle, v, f, width=640, height=104, n_super_samples=1, ao_num_samples=2)
    assert c["primary_hits"] > 0 and c["ao_occluded"] > 0


def test_twenty_million_triangles(rt, oracle):
    """6 GB of scene: beyond every cache of the device.  (Building it takes the CPU some tens of seconds.)"""
    from tools.big_meshes import terrain

    v, f = terrain(3200, scale=4.0)  # (at scale 1 its triangles are too small for the reference's test to see: tools/big_meshes.py)
    assert f.shape[0] == 20_480_000
    c = _check(rt, oracle, v, f, width=160, height=90, n_super_samples=1, ao_num_samples=3)
    assert c["primary_hits"] > 0.95 * 160 * 90 and c["ao_occluded"] > 0


@pytest.mark.parametrize("ao_distance", [0.2, 0.004, 30.0])
def test_slivers_and_huge_extents(rt, oracle, ao_distance):
    from tools.big_meshes import slivers

    v, f = slivers()
    _check(rt, oracle, v, f, width=160, height=90, n_super_samples=4, ao_num_samples=2, ao_max_distance=ao_distance)


@pytest.mark.parametrize("bvh_method", [0, 1])
def test_coplanar_duplicates_at_scale(rt, oracle, bvh_method):
    from tools.big_meshes import coplanar_stack

    v, f = coplanar_stack(160, 3)
    _check(rt, oracle, v, f, width=128, height=72, n_super_samples=4, ao_num_samples=3, bvh_method=bvh_method)
