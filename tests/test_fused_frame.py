"""GPU tests of the fused frame kernel (kernels/frame.hip.h): primary rays and ambient occlusion in one persistent
launch, the hit records handed from workgroup to workgroup inside it.  Bar: the bits of the two-kernel frame, which are
the oracle's (tests/test_hip_parity.py)."""
import hashlib

import numpy as np
import pytest

from conftest import bits, options_for

pytestmark = pytest.mark.gpu

CASES = ["bunny_256_s1_a3", "bunny_101x77_s9_a2", "blob_128x96_s4_a3", "ties_64_s4_a3", "ties_5x3_s1_a1", "single_32_s1_a3",
         "bunny_600_defaults", "bunny_1080p_s1_a3", "interior_1080p_s1_a3"]


@pytest.mark.parametrize("name", CASES)
def test_fused_frame_equals_the_two_kernel_frame(rt, golden, scene_for, name):
    """Every frame starts from a POISONED hit list: the records of an earlier frame of the same scene are the very bits
    this frame writes, so a workgroup that read a tile's records before they were handed over could not be told from one
    that waited -- unless what it finds there is garbage."""
    c = golden["renders"][name]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    host = rt.Host(opt, 0)
    host.expect_frames(1000)
    host.upload_scene(scene)
    host.set_frame_form("separate")
    assert not host.frame_is_fused
    host.render()
    assert hashlib.sha256(host.download().tobytes()).hexdigest() == c["float_sha256"]
    host.set_frame_form("fused")
    assert host.frame_is_fused
    for _ in range(12):
        host.poison_hit_list()
        host.render()
        assert hashlib.sha256(host.download().tobytes()).hexdigest() == c["float_sha256"]
        assert hashlib.md5(rt.pgm_bytes(host.download_u8())).hexdigest() == c["pgm_md5"]
    st = host.stats()
    assert st["primary_hits"] == c["counters"]["primary_hits"] and st["ao_occluded"] == c["counters"]["ao_occluded"]
    # the forms alternate on one host (the claim cursors and the frame count are the finishing kernel's to put back)
    for form in ("separate", "fused", "auto", "separate", "fused"):
        host.set_frame_form(form)
        host.poison_hit_list()
        host.render()
        assert hashlib.sha256(host.download().tobytes()).hexdigest() == c["float_sha256"], form
    host.close()


def test_the_fused_frame_is_an_experiment_not_the_rule(rt, golden, scene_for):
    """Two kernels per frame unless asked otherwise (the fused frame measured slower: profiles/r05_notes.md); where it
    cannot be used -- no ambient occlusion, RANDOM sampling -- asking changes nothing."""
    c = golden["renders"]["bunny_256_s1_a3"]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    host = rt.Host(opt, 0)
    host.upload_scene(scene)
    assert not host.frame_is_fused
    host.set_frame_form("fused")
    assert host.frame_is_fused
    host.set_device_share(3)  # (a smaller grid: the fused frame still renders the frame)
    host.render()
    assert hashlib.sha256(host.download().tobytes()).hexdigest() == c["float_sha256"]
    host.close()
    ring = rt.FrameRing(opt, scene, hosts=3)
    assert not ring.host(0).frame_is_fused
    ring.close()
    for changes in ({"ao_num_samples": 0}, {"ao_method": 1}):
        other = rt.Options.defaults(width=64, height=64, n_super_samples=1, **changes)
        h = rt.Host(other, 0)
        h.upload_scene(scene)
        h.set_frame_form("fused")
        assert not h.frame_is_fused
        h.render()
        h.close()


def test_fused_frames_through_a_replayed_graph(rt, golden, scene_for):
    """A ring of one host replays ONE captured graph: the frame number the flags are compared with cannot come in as an
    argument; the finishing kernel counts it on the device."""
    c = golden["renders"]["bunny_600_defaults"]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    ring = rt.FrameRing(opt, scene, hosts=1)
    host = ring.host(0)
    host.set_frame_form("fused")
    assert host.frame_is_fused
    for _ in range(5):
        host.poison_hit_list()
        ring.run(3)
        ring.drain()
        assert hashlib.md5(rt.pgm_bytes(ring.download_last())).hexdigest() == c["pgm_md5"]
        assert hashlib.sha256(host.download().tobytes()).hexdigest() == c["float_sha256"]
    ring.close()
