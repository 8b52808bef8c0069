"""The frame ring (include/rt_hip_ring.h, rt_ring_*) on the GPU: the library itself keeps several render hosts of one scene
busy on one GPU -- captured hipGraph per host, streams of different priority, frame bookkeeping, and, with a
communicator attached, the RCCL band gather.  The reference renders one blocking frame per OpenCLHost::operator()()
(src/opencl_host.cc:137-149); every frame of a ring must be exactly that frame, and the frames must really overlap."""
import hashlib

import numpy as np
import pytest

from conftest import options_for

pytestmark = pytest.mark.gpu

HEADLINE = "bunny_1080p_s1_a3"


def md5_of(rt, u8):
    return hashlib.md5(rt.pgm_bytes(u8)).hexdigest()


def test_ring_frames_are_golden_and_overlap(rt, golden, scene_for):
    """Three hosts, graph replay.  Every frame is the golden one, the hosts' streams differ, and the frames overlap ON
    THE DEVICE: in the steady state another frame is on the device for at least half of every frame's ambient-occlusion
    pass, and often the next frame's ordering step has ended before that pass has -- its whole primary pass ran beside
    it (HIP events for a frame's begin and end, the device clock stamped by the kernels themselves for the
    ambient-occlusion pass)."""
    c = golden["renders"][HEADLINE]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    ring = rt.FrameRing(opt, scene, hosts=3)
    assert ring.size == 3 and ring.slots == 6 and ring.local_rows == opt.height
    assert len({h.stream_handle for h in ring.hosts}) == 3 and all(h.stream_handle for h in ring.hosts)
    frames = []
    for frame in range(9):  # three in flight at any time from the third on
        if frame >= 3:
            frames.append(ring.collect())
        assert ring.submit() == frame
    with pytest.raises(rt.RtError) as e:
        ring.submit()
    assert e.value.code == rt.api.RT_E_STATE
    while len(frames) < 9:
        frames.append(ring.collect())
    with pytest.raises(rt.RtError):
        ring.collect()
    for u8 in frames:
        assert md5_of(rt, u8) == c["pgm_md5"]
    for h in ring.hosts:
        assert hashlib.sha256(h.download().tobytes()).hexdigest() == c["float_sha256"]
        st = h.stats()
        assert st["primary_hits"] == c["counters"]["primary_hits"] and st["ao_occluded"] == c["counters"]["ao_occluded"]
    # a steady stream: one call into the library for 30 frames
    ring.reset_clock()
    ring.keep_frame_times(True)
    first = ring.submit()
    ring.collect_info()
    ring.run(30)
    ring.drain()
    assert md5_of(rt, ring.download_last()) == c["pgm_md5"]
    t = {f: ring.frame_times(f) for f in range(first + 4, first + 30)}
    primary_inside = 0
    for f in range(first + 10, first + 29):  # (the first frames after the blocking one are the ring filling up)
        begin, ao_begin, ao_end, end = t[f]
        assert 0.0 < begin < ao_begin < ao_end <= end, (f, t[f])
        # another frame was on the device for at least half of this frame's ambient-occlusion pass (the union of the
        # other frames' spans: with three hosts the pass begins beside the frame before and ends beside the one after)
        pieces = sorted((max(ao_begin, t[g][0]), min(ao_end, t[g][3])) for g in t if g != f and t[g][3] > ao_begin and t[g][0] < ao_end)
        shared, reached = 0.0, ao_begin
        for lo, hi in pieces:
            if hi > max(lo, reached):
                shared += hi - max(lo, reached)
                reached = hi
        assert shared >= 0.5 * (ao_end - ao_begin), (f, t[f], shared)
        nxt = t[f + 1]
        if nxt[1] < ao_end:  # ... and the next frame's primary pass + ordering step were over before that pass was
            primary_inside += 1
    assert primary_inside >= 6, (primary_inside, t)
    timers = ring.timers()
    assert timers["frames"] >= 31 and timers["ao_frames"] >= 31 and 0.0 < timers["ao_ms"] < timers["kernel_ms"]
    cpu = ring.cpu_times()
    assert cpu["frames"] == 31
    ring.close()


@pytest.mark.parametrize("graph", [True, False])
@pytest.mark.parametrize("name", ["bunny_600_defaults", "bunny_101x77_s9_a2", "blob_33x17_s1_a0"])
def test_ring_of_one_is_the_blocking_frame(rt, golden, scene_for, name, graph):
    """A ring of one host = the reference's blocking operator(): same floats, same bytes, graph replay or plain
    launches, also for frames without an ambient-occlusion pass and for odd sizes."""
    c = golden["renders"][name]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    ring = rt.FrameRing(opt, scene, hosts=1)
    ring.set_graph_mode(graph)
    for _ in range(3):
        ring.step()
        assert ring.in_flight == 0
        assert md5_of(rt, ring.download_last()) == c["pgm_md5"]
    assert hashlib.sha256(ring.host(0).download().tobytes()).hexdigest() == c["float_sha256"]
    ring.close()


def test_ring_recaptures_after_a_new_scene(rt, golden, scene_for):
    """The captured graph bakes the scene's device pointers in: a second upload must lead to a new capture."""
    a, b = golden["renders"]["blob_128x96_s4_a3"], golden["renders"]["bunny_101x77_s9_a2"]
    opt = options_for(rt, a)
    ring = rt.FrameRing(opt, scene_for(a["mesh"], a["bvh"])[0], hosts=2)
    ring.run(4)
    ring.drain()
    assert md5_of(rt, ring.download_last()) == a["pgm_md5"]
    ring.upload_scene(scene_for(b["mesh"], b["bvh"])[0])  # another mesh under the same options
    ring.run(4)
    ring.drain()
    other = ring.download_last()
    host = rt.Host(opt, 0)
    host.upload_scene(scene_for(b["mesh"], b["bvh"])[0])
    host.render()
    assert np.array_equal(other, host.download_u8()) and md5_of(rt, other) != a["pgm_md5"]
    host.close()
    ring.close()


def test_ring_holds_one_copy_of_the_scene(rt, golden, scene_for):
    """The reference uploads a scene once (src/opencl_host.cc:120-136).  A ring's hosts share ONE set of scene arrays on
    the device: the bytes of the scene do not depend on the number of hosts, only the per-host frame buffers do, and
    the frames are the golden ones whichever host renders them."""
    case = golden["renders"]["bunny_101x77_s9_a2"]
    opt = options_for(rt, case)
    scene = scene_for(case["mesh"], case["bvh"])[0]
    seen = {}
    for hosts in (1, 3, 6):
        ring = rt.FrameRing(opt, scene, hosts=hosts)
        scene_bytes, copies, total = ring.device_bytes()
        assert copies == 1 and scene_bytes > 0
        seen[hosts] = (scene_bytes, total)
        ring.run(2 * hosts + 1)  # every host and every band buffer at least once
        ring.drain()
        assert md5_of(rt, ring.download_last()) == case["pgm_md5"]
        ring.close()
    assert seen[1][0] == seen[3][0] == seen[6][0]
    # total = scene + (what the hosts share besides) + hosts x (a host's own buffers)
    per_host = (seen[6][1] - seen[3][1]) // 3
    shared = seen[3][1] - seen[3][0] - 3 * per_host
    assert seen[6][1] == seen[6][0] + shared + 6 * per_host and seen[1][1] == seen[1][0] + shared + per_host
    # ... shared: the table of walk intervals, 1 + ao_dirs of two words per tile (rt_walk_entries); a host's own buffers
    # follow what is HIT: 36 bytes (hit record + occlusion counter) per hit sub-pixel, not 64 slots for every tile -- plus
    # the float image, the 8-bit bands and three words per tile
    sub = opt.total_width * opt.total_height
    tiles = ((opt.total_width + 7) // 8) * ((opt.total_height + 7) // 8)
    hits = case["counters"]["primary_hits"]
    dirs = case["counters"]["ao_rays"] // hits
    assert dirs * hits == case["counters"]["ao_rays"]
    # (the image and the words per tile cover whole bands of tile rows: a few rows more than the frame)
    assert 8 * (dirs + 1) * tiles <= shared <= 1.15 * 8 * (dirs + 1) * tiles + 64
    assert per_host <= 36 * hits + 1.15 * 4 * sub + 2 * opt.width * opt.height + 16 * tiles + 8192
    assert per_host < 0.75 * (36 * 64 * tiles)  # (the old layout's hit list alone -- 64 slots per tile -- was larger than all of it)


def test_both_forms_of_the_ao_pass_render_the_same_frame(rt, golden, scene_for):
    """The ambient-occlusion pass exists with and without look-ahead loads in its node loop (kernels.hip,
    OCRT_PF_SUCCESSORS); a ring measures both at upload and keeps the faster.  Whatever it keeps, the frame is the golden
    one -- here: each form forced on a host, then a ring with the calibration on and one with it off."""
    case = golden["renders"][HEADLINE]
    opt = options_for(rt, case)
    scene = scene_for(case["mesh"], case["bvh"])[0]
    images = []
    for form in (False, True):
        host = rt.Host(opt, 0)
        host.set_ao_prefetch(form)
        host.upload_scene(scene)
        host.render()
        images.append(host.download())
        assert md5_of(rt, host.download_u8()) == case["pgm_md5"]
        st = host.stats()
        assert st["ao_occluded"] == case["counters"]["ao_occluded"] and st["primary_hits"] == case["counters"]["primary_hits"]
        host.close()
    assert np.array_equal(images[0].view(np.uint32), images[1].view(np.uint32))
    ring = rt.FrameRing(opt, scene, hosts=3)
    without, with_, in_use = ring.calibration()
    assert 0.2 < without < 20.0 and 0.2 < with_ < 20.0  # measured, in ms, on this frame
    assert in_use == (with_ <= without)
    ring.run(7)
    ring.drain()
    assert md5_of(rt, ring.download_last()) == case["pgm_md5"]
    ring.close()
    ring = rt.FrameRing(opt, None, hosts=2)
    ring.set_calibration(False)
    ring.upload_scene(scene)
    assert ring.calibration()[:2] == (0.0, 0.0)
    ring.run(5)
    ring.drain()
    assert md5_of(rt, ring.download_last()) == case["pgm_md5"]
    ring.close()


def test_a_borrowed_host_cannot_outlive_its_ring(rt, golden, scene_for):
    """FrameRing.host() hands out views into the ring: a view keeps the ring alive, and once the ring is closed every
    call on it fails cleanly instead of touching freed memory."""
    import gc

    case = golden["renders"]["blob_128x96_s4_a3"]
    opt = options_for(rt, case)
    view = rt.FrameRing(opt, scene_for(case["mesh"], case["bvh"])[0], hosts=2).host(0)  # the ring itself is a temporary
    gc.collect()
    assert view.stats()["primary_rays"] == 0  # nothing rendered yet; the call reaches a live host
    ring = view._owner
    ring.run(2)
    ring.drain()
    assert view.stats()["primary_rays"] > 0
    ring.close()
    with pytest.raises(rt.RtError):
        view.stats()
    assert view.last_kernel_ms == 0.0


class _DeviceBytes:
    """A few bytes of device memory straight from the HIP runtime the library itself uses (torch brings its own copy
    of the runtime, which must be loaded BEFORE the library -- bench.py does that, a test process cannot)."""

    def __init__(self, size):
        import ctypes as C

        self.hip, self.size, self.ptr = C.CDLL("libamdhip64.so"), size, C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(size)) == 0
        assert self.hip.hipMemset(self.ptr, 0, C.c_size_t(size)) == 0

    def numpy(self, rows, width):
        import ctypes as C

        out = np.empty((rows, width), dtype=np.uint8)
        assert self.hip.hipMemcpy(C.c_void_p(out.ctypes.data), self.ptr, C.c_size_t(rows * width), 2) == 0  # device -> host
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def test_ring_writes_into_bound_device_memory(rt, golden, scene_for):
    """rt_ring_bind_output: the frames' bands land in caller-owned device memory (the send buffer of a caller-side
    collective in bench.py's gloo rehearsal)."""
    c = golden["renders"]["bunny_600_defaults"]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    ring = rt.FrameRing(opt, scene, hosts=2)
    bands = [_DeviceBytes(ring.local_rows * opt.width) for _ in range(ring.slots + 1)]
    for k in range(ring.slots):
        ring.bind_output(k, bands[k].ptr.value)
    for k in range(0, ring.slots, 2):
        ring.submit()
        ring.submit()
        for j in (k, k + 1):
            frame, slot, ptr = ring.collect_info()
            assert (frame, slot, ptr) == (j, j, bands[j].ptr.value)
            assert md5_of(rt, bands[j].numpy(opt.height, opt.width)) == c["pgm_md5"]
    ring.submit()
    assert rt.load_library().rt_ring_collect_into_device(ring._r, bands[-1].ptr.value) == 0
    assert md5_of(rt, bands[-1].numpy(opt.height, opt.width)) == c["pgm_md5"]
    ring.close()
    for b in bands:
        b.free()


@pytest.mark.parametrize("name,nranks", [("interior_4k_s1_a3", 8), ("bunny_1080p_s64_a3", 8), (HEADLINE, 3)])
def test_full_size_frames_split_over_ranks_reassemble(rt, golden, scene_for, name, nranks):
    """BASELINE configs 4 and 5 as the driver's 8-GPU run cuts them -- the 4K frame in 270 bands of 8 rows dealt to 8
    ranks (unevenly: 34 / 33 bands), the 64-spp frame in bands of ONE output row -- rendered rank by rank on the one
    GPU through rings of two hosts, rows put in place by the partition arithmetic: the golden PGM."""
    c = golden["renders"][name]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    image = np.zeros((opt.height, opt.width), dtype=np.uint8)
    seen = np.zeros(opt.height, dtype=np.int32)
    hits = occluded = 0
    for rank in range(nranks):
        ring = rt.FrameRing(opt, scene, 0, rank, nranks, hosts=2)
        rows = rt.partition_rows(opt, rank, nranks)
        assert ring.local_rows == rows.size
        if rank == 1:  # (one share with EVERY tile of its bands cast in quarters by the primary pass, one with none)
            ring.host(1).set_primary_split(1)
        elif rank == 2:
            ring.host(1).set_primary_split(0)
        ring.submit()
        ring.submit()
        ring.collect_info()
        ring.collect_info()
        local = ring.host(1).download_u8_local()
        keep = rows < opt.height
        image[rows[keep]] = local[keep]
        seen[rows[keep]] += 1
        st = ring.host(1).stats()
        hits += st["primary_hits"]
        occluded += st["ao_occluded"]
        ring.close()
    assert (seen == 1).all()
    assert md5_of(rt, image) == c["pgm_md5"]
    assert hits == c["counters"]["primary_hits"] and occluded == c["counters"]["ao_occluded"]


def test_ring_gathers_over_rccl_in_a_world_of_one(rt, golden, scene_for):
    """The library's own RCCL leg on the box: unique id, ncclCommInitRank, a checked grouped self send/recv, and the
    exchange step behind the frames (with one rank it moves nothing, the assembly kernel still builds the image)."""
    if not rt.rccl_available():
        pytest.fail("librccl.so.1 cannot be opened on the GPU box")
    c = golden["renders"]["bunny_600_defaults"]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    ring = rt.FrameRing(opt, scene, hosts=3)
    ring.attach_rccl(rt.rccl_unique_id())
    ring.rccl_self_test()
    ring.run(7)
    ring.drain()
    assert ring.last_image_device() != 0
    assert md5_of(rt, ring.download_last()) == c["pgm_md5"]
    ring.close()
