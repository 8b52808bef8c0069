"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle on the same inputs and against the golden vectors generated from the
reference's own kernel.  Bar: bit-exact floats and bytes (integer PGM)."""
import hashlib

import numpy as np
import pytest

from conftest import bits, options_for

pytestmark = pytest.mark.gpu

# golden cases small enough for the oracle to re-render in the test itself
ORACLE_CASES = [
    "bunny_256_s1_a0", "bunny_256_s1_a3", "bunny_64_s1_a3", "bunny_101x77_s9_a2", "bunny_50x40_s5_a1_f15",
    "bunny_96x54_s1_a4_alpha", "blob_128x96_s4_a3", "blob_128x96_s4_a3_sah", "blob_80_s1_a5_noshade",
    "blob_33x17_s1_a0", "ties_33_s1_a3", "ties_33_s1_a3_sah", "ties_64_s4_a3", "ties_5x3_s1_a1", "single_32_s1_a3",
    "interior_hard_160x90_s4_a3",
]
# full-size cases are checked against the committed digests only
# (BASELINE.json configs 2-5: bunny 1080p, interior stand-in 1080p and 4K, bunny with 64 samples per pixel)
DIGEST_CASES = ["bunny_1080p_s1_a0", "bunny_1080p_s1_a3", "bunny_600_defaults", "interior_1080p_s1_a3",
                "interior_4k_s1_a3", "bunny_1080p_s64_a3", "interior_hard_1080p_s1_a3", "interior_hard_4k_s1_a3"]


def render_hip(rt, scene, opt, rank=0, nranks=1, frames=None):
    """`frames`: announced to the host before the upload (rt_expect_frames) -- None: nothing is, the host is the
    reference's one-shot OpenCLHost and its upload prepares no walk intervals; many: it does."""
    host = rt.Host(opt, 0, rank, nranks)
    if frames is not None:
        host.expect_frames(frames)
    host.upload_scene(scene)
    host.render()
    return host


STREAM = 1000  # frames announced by the tests that want what an upload prepares for a stream of frames


@pytest.mark.parametrize("frames", [None, STREAM], ids=["one_shot", "stream"])
@pytest.mark.parametrize("name", ORACLE_CASES)
def test_matches_oracle_and_golden(rt, oracle, golden, scene_for, name, frames):
    import orc

    c = golden["renders"][name]
    opt = options_for(rt, c)
    scene, arrays = scene_for(c["mesh"], c["bvh"])
    host = render_hip(rt, scene, opt, frames=frames)
    img = host.download()
    u8 = host.download_u8()
    ref_img, counters, _ = oracle.render(orc.params_from_options(opt), arrays)
    mism = np.count_nonzero(bits(img) != bits(ref_img))
    assert mism == 0, f"{mism} float words differ from the oracle"
    assert np.array_equal(u8, oracle.resize(ref_img, opt.width, opt.height, opt.n_super_samples))
    assert hashlib.sha256(img.tobytes()).hexdigest() == c["float_sha256"]
    assert hashlib.md5(rt.pgm_bytes(u8)).hexdigest() == c["pgm_md5"]
    st = host.stats()
    assert st["primary_rays"] == counters["primary_rays"]
    assert st["primary_hits"] == counters["primary_hits"]
    assert st["ao_rays"] == counters["ao_rays"]
    assert st["ao_occluded"] == counters["ao_occluded"]
    host.close()


@pytest.mark.parametrize("name", DIGEST_CASES)
def test_full_size_golden_digests(rt, golden, scene_for, name):
    c = golden["renders"][name]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    # (the 1080p frames as the one-shot host renders them, the others with what a stream's upload prepares)
    host = render_hip(rt, scene, opt, frames=None if "1080p_s1" in name else STREAM)
    img = host.download()
    assert hashlib.sha256(img.tobytes()).hexdigest() == c["float_sha256"]
    assert hashlib.md5(rt.pgm_bytes(host.download_u8())).hexdigest() == c["pgm_md5"]
    # device resize == host resize of the downloaded floats
    assert np.array_equal(host.download_u8(), rt.resize_cpu(opt, img))
    st = host.stats()
    assert st["primary_hits"] == c["counters"]["primary_hits"]
    assert st["ao_occluded"] == c["counters"]["ao_occluded"]
    host.close()


@pytest.mark.parametrize("mesh", ["interior", "interior_hard"])
def test_interior_standin_is_the_pinned_mesh(rt, golden, scene_for, mesh):
    """The interior stand-ins are generated (tools/make_interior_mesh.py), not stored: their arrays must be the ones
    the reference's loader and builder produced when the goldens were made."""
    import hashlib as h

    scene, arrays = scene_for(mesh, "longest")
    g = golden["scenes"][f"{mesh}/longest"]
    assert h.sha256(arrays.vertices.tobytes()).hexdigest() == g["vertices"]
    assert h.sha256(arrays.nodes.tobytes()).hexdigest() == g["nodes"]
    assert h.sha256(arrays.aabbs.tobytes()).hexdigest() == g["aabbs"]
    assert h.sha256(arrays.faces.tobytes()).hexdigest() == g["sorted_faces"]


def test_zero_normals_on_a_rebuilt_tree(rt, oracle, scene_for, monkeypatch):
    """Zero-length vertex normals give NaN shading normals, hence NaN ambient-occlusion rays: their packets take
    the exact form of the shared walk -- here on the REBUILT tree of a regular scene (no OCRT_KEEP_TREE), where
    that form is not the reference's own walk of the uploaded array but must still give its result."""
    import orc

    monkeypatch.delenv("OCRT_KEEP_TREE", raising=False)
    _, arrays = scene_for("blob", "longest")
    normals = arrays.normals.copy()
    normals[::7] = 0.0  # every 7th vertex normal: triangles with one, two or three zero normals
    normals[arrays.faces.reshape(-1, 3)[::5].ravel()] = 0.0  # ... and every 5th triangle with all three zero: NaN normals
    damaged = orc.SceneArrays(arrays.faces, arrays.nodes, arrays.aabbs, arrays.vertices, normals)
    opt = rt.Options.defaults(width=160, height=120, n_super_samples=1, ao_num_samples=3, ao_max_distance=0.4)
    host = rt.Host(opt, 0)
    host.upload(damaged.faces, damaged.nodes, damaged.aabbs, damaged.vertices, damaged.normals)
    host.render()
    ref_img, counters, _ = oracle.render(orc.params_from_options(opt), damaged)
    got = host.download()
    # the case really occurs in this frame: a NaN normal shades to clamp(NaN) = 0 on a hit sub-pixel
    clean_img, _, _ = oracle.render(orc.params_from_options(opt), arrays)
    assert np.count_nonzero((ref_img == 0.0) & (clean_img > 0.0)) > 5
    same = (bits(got) == bits(ref_img)) | (np.isnan(got) & np.isnan(ref_img))
    assert same.all(), int((~same).sum())
    st = host.stats()
    assert st["primary_hits"] == counters["primary_hits"] and st["ao_occluded"] == counters["ao_occluded"]
    host.close()


@pytest.mark.parametrize("samples,width,height,ao", [(4, 37, 11, 2), (9, 37, 11, 2), (16, 261, 5, 2), (25, 19, 7, 2), (49, 23, 9, 2),
                                                     (64, 70, 3, 2), (81, 13, 6, 2), (256, 21, 5, 2), (1024, 9, 4, 2), (4225, 3, 2, 2),
                                                     (16, 29, 6, 0), (64, 300, 2, 0)])
def test_supersampled_frames_finish_like_the_oracle(rt, oracle, scene_for, samples, width, height, ao):
    """The frame's last kernel at every grid size: n x n sub-pixels per pixel with n = 2 ... 32 through the kernel that
    sweeps along the sub-pixel rows (one, several and a fraction of a workgroup's run of pixels per row; widths that are no
    multiple of anything), n = 65 through the one-thread-per-pixel form it falls back to beyond its LDS cells -- floats
    (the ambient-occlusion factor written back) and bytes (box filter in the reference's order, src/ray_tracer.cc:7-13)
    against the oracle's, whole and cut into the bands of three ranks; also without ambient occlusion (no tags: the box
    filter alone)."""
    import orc

    scene, arrays = scene_for("blob", "longest")
    opt = rt.Options.defaults(width=width, height=height, n_super_samples=samples, ao_num_samples=ao, enable_ao=int(ao != 0))
    ref_img, counters, _ = oracle.render(orc.params_from_options(opt), arrays)
    ref_u8 = oracle.resize(ref_img, opt.width, opt.height, opt.n_super_samples)
    host = render_hip(rt, scene, opt)
    assert np.count_nonzero(bits(host.download()) != bits(ref_img)) == 0
    assert np.array_equal(host.download_u8(), ref_u8)
    assert host.stats()["ao_occluded"] == counters["ao_occluded"]
    host.close()
    occluded = 0
    rows_seen = np.zeros(height, dtype=bool)
    for rank in range(3):
        part = render_hip(rt, scene, opt, rank, 3)
        rows = part.local_to_global_rows()
        local = part.download_u8_local()
        keep = rows < height
        assert np.array_equal(local[keep], ref_u8[rows[keep]]), rank
        rows_seen[rows[keep]] = True
        occluded += part.stats()["ao_occluded"]
        part.close()
    assert rows_seen.all() and occluded == counters["ao_occluded"]


def test_tree_independence_1080p(rt, golden, scene_for):
    """the PGM does not depend on the BVH strategy (bunny has no
    equal-distance ties between different leaves at these pixels)."""
    c = dict(golden["renders"]["bunny_1080p_s1_a3"])
    opt = options_for(rt, c)
    scene_sah, _ = scene_for("bunny", "sah")
    host = render_hip(rt, scene_sah, opt)
    assert hashlib.md5(rt.pgm_bytes(host.download_u8())).hexdigest() == c["pgm_md5"]
    host.close()


@pytest.mark.parametrize("nranks", [2, 3, 8])
@pytest.mark.parametrize("name", ["bunny_256_s1_a3", "bunny_101x77_s9_a2", "blob_128x96_s4_a3"])
def test_band_partition_reassembles(rt, golden, scene_for, name, nranks):
    """Multi-GPU sharding on one device: every rank's bands, reassembled, equal
    the unpartitioned frame (floats and bytes)."""
    c = golden["renders"][name]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    full = render_hip(rt, scene, opt)
    full_u8 = full.download_u8()
    full.close()
    out = np.full_like(full_u8, 255)
    seen = np.zeros(opt.height, dtype=np.int32)
    for rank in range(nranks):
        host = render_hip(rt, scene, opt, rank, nranks)
        rows = host.local_to_global_rows()
        assert np.array_equal(rows, rt.partition_rows(opt, rank, nranks))
        local = host.download_u8_local()
        keep = rows < opt.height
        out[rows[keep]] = local[keep]
        seen[rows[keep]] += 1
        host.close()
    assert np.all(seen == 1)
    assert np.array_equal(out, full_u8)
    assert hashlib.md5(rt.pgm_bytes(out)).hexdigest() == c["pgm_md5"]


def test_rerender_is_idempotent_and_timed(rt, golden, scene_for):
    c = golden["renders"]["bunny_256_s1_a3"]
    opt = options_for(rt, c)
    scene, _ = scene_for("bunny", "longest")
    host = render_hip(rt, scene, opt)
    first = host.download()
    host.reset_timers()
    for _ in range(3):
        host.render_async()
    host.sync()
    assert host.kernel_launches == 3
    assert host.total_kernel_ms > 0 and host.last_kernel_ms > 0
    assert np.array_equal(bits(first), bits(host.download()))
    host.close()


def test_error_paths(rt, scene_for):
    opt = rt.Options.defaults(width=16, height=16, n_super_samples=1)
    host = rt.Host(opt, 0)
    with pytest.raises(rt.RtError) as e:
        host.render()
    assert e.value.code == -4  # RT_E_STATE: render before upload
    scene, arrays = scene_for("blob", "longest")
    bad_nodes = arrays.nodes.copy()
    bad_nodes[1] = 10_000_000  # subtree runs past the array
    with pytest.raises(rt.RtError) as e:
        host.upload(arrays.faces, bad_nodes, arrays.aabbs, arrays.vertices, arrays.normals)
    assert e.value.code == -1
    # an inner node with three children: the reference's triangle counter (intersect_kernel.cl:191) has no
    # meaning on such an array, so upload refuses it
    inner = int(np.flatnonzero(arrays.nodes > 8)[5])
    three = arrays.nodes.copy()
    three[[a for a in range(inner) if a + three[a] > inner]] -= 1
    with pytest.raises(rt.RtError) as e:
        host.upload(arrays.faces, np.delete(three, inner), np.delete(arrays.aabbs, [2 * inner, 2 * inner + 1], axis=0),
                    arrays.vertices, arrays.normals)
    assert e.value.code == -1 and "binary tree" in str(e.value)
    bad_faces = arrays.faces.copy()
    bad_faces[0] = 10_000_000
    with pytest.raises(rt.RtError):
        host.upload(bad_faces, arrays.nodes, arrays.aabbs, arrays.vertices, arrays.normals)
    host.close()
    with pytest.raises(rt.RtError):
        rt.Host(opt, 99)  # device index out of range


def test_interior_standin_matches_oracle(rt, oracle):
    """The closed interior scene (stand-in for the missing sibenik.off): every
    primary ray hits, AO rays see close occluders everywhere."""
    import orc
    from tools.meshes import interior_path

    scene = rt.Scene.load_off(interior_path()).build_bvh(0)
    arrays = orc.SceneArrays.from_scene(scene)
    for (w, h, ss, ao) in ((160, 90, 1, 3), (64, 36, 4, 2)):
        opt = rt.Options.defaults(width=w, height=h, n_super_samples=ss, ao_num_samples=ao)
        host = render_hip(rt, scene, opt)
        ref_img, counters, _ = oracle.render(orc.params_from_options(opt), arrays)
        assert np.array_equal(bits(host.download()), bits(ref_img))
        assert np.array_equal(host.download_u8(), oracle.resize(ref_img, w, h, ss))
        st = host.stats()
        assert st["primary_hits"] == counters["primary_hits"] == opt.total_width * opt.total_height
        assert st["ao_occluded"] == counters["ao_occluded"]
        host.close()


@pytest.mark.parametrize("name", ["bunny_256_s1_a3", "bunny_256_s1_a0", "blob_128x96_s4_a3", "ties_64_s4_a3", "bunny_600_defaults"])
def test_tiles_cast_in_quarters_change_nothing(rt, golden, scene_for, name):
    """The primary pass of a stream of frames casts its costliest tiles in quarters, the four waves of a workgroup at once
    (kernels/primary.hip.h): which rays share a packet never changes what a ray finds -- with every tile in quarters, with
    some, with none: the same floats, the same hit list (the ambient-occlusion factors), the same tile words."""
    c = golden["renders"][name]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    host = render_hip(rt, scene, opt, frames=STREAM)
    assert hashlib.sha256(host.download().tobytes()).hexdigest() == c["float_sha256"]
    words = host.tile_order()["words"].copy()
    for above in (1, 8, 33, 64, 0):
        host.set_primary_split(above)
        for _ in range(2):
            host.render()
            assert hashlib.sha256(host.download().tobytes()).hexdigest() == c["float_sha256"], above
        st = host.stats()
        assert st["primary_hits"] == c["counters"]["primary_hits"] and st["ao_occluded"] == c["counters"]["ao_occluded"], above
        assert np.array_equal(host.tile_order()["words"], words), above
    host.close()


@pytest.mark.parametrize("name", ["bunny_256_s1_a3", "blob_128x96_s4_a3", "ties_64_s4_a3", "bunny_600_defaults"])
def test_measured_tile_order_changes_nothing(rt, golden, scene_for, name):
    """What the AO pass claims first is decided once per upload -- by the cost classes of the counting pass, or, for a
    stream of frames, by what the tiles' claims were measured to take (rt_debug_measure_tile_costs) -- and is nobody's
    business but the scheduler's: the same tiles in another order, the same floats.  Also with a list turned round."""
    c = golden["renders"][name]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    host = render_hip(rt, scene, opt, frames=STREAM)
    before = host.tile_order()
    assert hashlib.sha256(host.download().tobytes()).hexdigest() == c["float_sha256"]
    host.measure_tile_costs(2)
    after = host.tile_order()
    work = after["words"] >> 8 != 0
    assert work.any() and (after["costs"][work] > 0).all() and (after["costs"][~work] == 0).all()
    assert np.array_equal(after["constants"], before["constants"])
    assert sorted(after["order"].tolist()) == sorted(before["order"].tolist())  # the same entries ...
    host.render()
    assert hashlib.sha256(host.download().tobytes()).hexdigest() == c["float_sha256"]
    assert host.stats()["ao_occluded"] == c["counters"]["ao_occluded"]
    # heavy tiles claimed half a tile at a time (kernels/ao.hip.h): with the threshold at a hundredth of the pass's ideal length
    # that is most of the list's head, with it out of reach none -- the same floats
    for split_above in (0.01, 0.25, 1e9, 0.0):
        host.set_order_policy(2.0, 2.0, split_above)
        host.render()
        assert hashlib.sha256(host.download().tobytes()).hexdigest() == c["float_sha256"], split_above
        assert host.stats()["ao_occluded"] == c["counters"]["ao_occluded"]
    # ... and each group's list backwards (cheapest first: the worst order there is)
    order, at = after["order"].copy(), 0
    n = int(np.sqrt(opt.n_super_samples))
    tiles_x, rows = (opt.width * n + 7) // 8, (opt.height * n + 7) // 8
    for g in range(8):
        count = int(after["constants"][g][0])
        order[at:at + count] = order[at:at + count][::-1].copy()
        at += (((tiles_x + 1) // 2 + 7 - g) >> 3) * 2 * rows
    host.set_tile_order(order, after["constants"])
    host.render()
    assert hashlib.sha256(host.download().tobytes()).hexdigest() == c["float_sha256"]
    with pytest.raises(rt.RtError):
        host.set_tile_order(order[:-1], after["constants"])
    bad = order.copy()
    bad[0] = 0x03FFFFFF
    with pytest.raises(rt.RtError):
        host.set_tile_order(bad, after["constants"])
    host.close()


@pytest.mark.parametrize("mesh", ["interior", "bunny"])
def test_walk_intervals_confine_the_walks_and_change_nothing(rt, oracle, scene_for, mesh):
    """An upload finds, per tile and per table direction of a full tile, the interval of the node array its any-hit rays
    can reach within AO_MAX_DISTANCE (entry_kernel), and the packets walk that alone.  Whatever the distance -- a hundredth
    of the scene, the reference's default, more than the scene -- the image is the oracle's; and the narrowing is there
    when the distance is short and gone when it is not."""
    import orc

    scene, arrays = scene_for(mesh, "longest")
    shares = []
    for distance in (0.02, 0.2, 0.9, 50.0):
        opt = rt.Options.defaults(width=128, height=72, n_super_samples=4, ao_num_samples=2, ao_max_distance=distance)
        host = render_hip(rt, scene, opt)  # nothing announced: the whole array for every packet
        ref_img, counters, _ = oracle.render(orc.params_from_options(opt), arrays)
        assert np.array_equal(bits(host.download()), bits(ref_img)), distance
        e = host.walk_entries()
        assert e["tiles_narrowed"] == 0 and e["mean_share"] == 1.0 and e["mean_packet_share"] == 1.0
        host.expect_frames(STREAM)  # announced after the upload: the hit list is laid out again, with the table
        host.render()
        assert np.array_equal(bits(host.download()), bits(ref_img)), distance
        assert host.stats()["ao_occluded"] == counters["ao_occluded"]
        e = host.walk_entries()
        assert e["tiles_hit"] > 0 and 0.0 < e["mean_packet_share"] <= e["mean_share"] <= 1.0
        shares.append(e["mean_share"])
        if distance == 50.0:
            assert e["tiles_narrowed"] == 0 and e["mean_share"] == 1.0
        host.close()
    assert shares[0] < shares[1] <= shares[2] <= shares[3]
    assert shares[0] < 0.5  # (a tile and a fiftieth of the scene around it: half the tree at most, on average)


def test_render_cli_writes_the_golden_pgm(rt, golden, tmp_path):
    """End to end through the `render` binary: same flags as the reference CLI (src/render.cc:19-43), the
    reference's phase lines and ray notice (:63-128), PGM file byte-identical to the golden one -- on one device,
    with the reference's own host-side resize, and with the frame split over several ranks (`--gpus`, here mapped
    onto the one GPU of the box) and gathered."""
    import os
    import subprocess

    from conftest import ROOT, mesh_file

    exe = os.path.join(ROOT, "opencl_raytracer_amd", "bin", "render")
    env = dict(os.environ, OCRT_SHARE_DEVICES="1")
    runs = (("bunny_256_s1_a3", []), ("blob_128x96_s4_a3_sah", ["-r", "sah"]), ("bunny_256_s1_a3", ["--host-resize", "1"]),
            ("bunny_101x77_s9_a2", ["--gpus", "3"]), ("bunny_600_defaults", ["--gpus=8"]), ("blob_128x96_s4_a3", ["--gpus", "2"]),
            ("bunny_600_defaults", ["--gpus", "5", "--gather", "peer"]),
            # the RCCL form of the exchange step (ncclCommInitAll, grouped send / receive, row assembly): RCCL refuses
            # two ranks on one device, so on a one-GPU box it is the group of one; with two GPUs, of two
            ("bunny_600_defaults", ["--gpus", "1", "--gather", "rccl"]),
            ("bunny_256_s1_a3", ["--gpus", str(min(2, rt.device_count())), "--gather", "rccl"]),
            # a steady stream of frames through a ring of render hosts (HipHostRing)
            ("bunny_600_defaults", ["--frames", "12"]), ("bunny_101x77_s9_a2", ["--frames=7", "--in-flight", "2"]),
            ("blob_128x96_s4_a3", ["--in-flight", "4"]))
    for name, extra in runs:
        c = golden["renders"][name]
        out = tmp_path / (name + ".pgm")
        cmd = [exe, "-w", str(c["width"]), "-h", str(c["height"]), "-s", str(c["ss"]),
               "--ambient-occlusion-samples=" + str(c["ao"]), "-d", str(c["aod"]), "-f", str(c["focal"])] + extra + \
              [mesh_file(c["mesh"]), str(out)]
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        for phase in ("Building BVH", "Loading OpenCL kernel", "Rendering image", "Total time (without loading memory"):
            assert phase in r.stdout, (phase, extra)
        if "--host-resize" in extra:
            assert "Loading memory" in r.stdout and "Resizing image on host" in r.stdout
        if c["ao"]:
            assert "IMPORTANT INFO: You've enabled 'Uniform AO hemispheres'" in r.stdout
            if c["ao"] == 3:
                assert "This will result in 25 rays." in r.stdout  # the reference's own estimate (the kernel casts 28)
        # the device table of OpenCLHost::printInfo (reference src/opencl_host.cc:76-119), HIP edition
        for line in ("Hardware information", "Max compute units", "Wavefront size", "Using Device" if "--gpus" not in " ".join(extra) else "Rank 0 of"):
            assert line in r.stdout, line
        if "--gpus" in " ".join(extra):
            assert ("Rank 1 of" in r.stdout) == (extra[extra.index("--gpus") + 1] != "1" if "--gpus" in extra else True)
            if "--gather" in extra:
                assert ("over RCCL" if "rccl" in extra else "by peer copies") in r.stdout, extra
        if "--frames" in " ".join(extra):
            assert "render hosts taking frames in turn" in r.stdout and "over the whole stream" in r.stdout
        assert hashlib.md5(out.read_bytes()).hexdigest() == c["pgm_md5"], (name, extra)
    # more ranks than GPUs without the rehearsal knob: refused, like any other bad device request
    env.pop("OCRT_SHARE_DEVICES")
    if rt.device_count() < 4:
        r = subprocess.run([exe, "--gpus", "4", mesh_file("blob"), str(tmp_path / "x.pgm")], capture_output=True, text=True, env=env)
        assert r.returncode != 0 and "more ranks than visible HIP devices" in (r.stdout + r.stderr)


@pytest.mark.parametrize("mesh,samples", [("blob", 6), ("bunny", 16)])
def test_random_ao_statistical_parity(rt, oracle, scene_for, mesh, samples):
    """`-m random` (reference intersect_kernel.cl:128-183,257-276) calls acos/sin/
    cos/cospi/sinpi on per-pixel data; device and host libm round differently, so
    this mode is outside the bit-exact contract.  The integer
    generator and the ray bookkeeping are exact, so: identical ray counts, and
    8-bit images that differ only where a ray flipped between hit and miss --
    tolerance: mean |delta| <= 0.25 grey levels, >= 97 % of the pixels identical,
    no pixel further off than three occlusion steps."""
    import orc

    opt = rt.Options.defaults(width=160, height=120, n_super_samples=1, ao_num_samples=samples, ao_method=1,
                              ao_max_distance=0.3)
    scene, arrays = scene_for(mesh, "longest")
    host = render_hip(rt, scene, opt)
    gpu = host.download_u8().astype(np.int32)
    st = host.stats()
    ref_img, counters, _ = oracle.render(orc.params_from_options(opt), arrays)
    ref = oracle.resize(ref_img, opt.width, opt.height, 1).astype(np.int32)
    assert st["primary_hits"] == counters["primary_hits"]
    assert st["ao_rays"] == counters["ao_rays"] == counters["primary_hits"] * (samples + 2)
    delta = np.abs(gpu - ref)
    assert delta.mean() <= 0.25, delta.mean()
    assert (delta == 0).mean() >= 0.97, (delta == 0).mean()
    assert delta.max() <= 3 * 255 // (samples + 1) + 1, delta.max()
    assert abs(st["ao_occluded"] - counters["ao_occluded"]) <= 0.002 * counters["ao_rays"] + 8
    host.close()


@pytest.mark.parametrize("damage", ["inverted_box", "nan_leaf_box", "huge_box", "inf_box", "nan_vertex",
                                    "child_outside_parent", "collinear_triangle"])
def test_irregular_scene_arrays_match_oracle(rt, oracle, scene_for, damage):
    """Scene arrays a BVH builder would never emit (inverted / NaN / infinite /
    overflowing boxes, a NaN vertex, a parent box that does not contain its
    child) arrive through the upload API: the kernels must then fall back to
    the reference's own slab test and the exact form of the shared walk and
    still agree with the oracle bit for bit."""
    import orc

    _, arrays = scene_for("blob", "longest")
    nodes, aabbs, verts, faces = arrays.nodes.copy(), arrays.aabbs.copy(), arrays.vertices.copy(), arrays.faces.copy()
    inner = int(np.flatnonzero(nodes > 8)[5])
    leaf = int(np.flatnonzero(nodes == 1)[40])
    if damage == "inverted_box":
        aabbs[2 * inner, 0], aabbs[2 * inner + 1, 0] = aabbs[2 * inner + 1, 0], aabbs[2 * inner, 0]
    elif damage == "nan_leaf_box":
        aabbs[2 * leaf, 1] = np.nan
    elif damage == "huge_box":
        aabbs[0, :3] = -3.0e38
        aabbs[1, :3] = 3.0e38
    elif damage == "inf_box":
        aabbs[0, 2] = -np.inf
        aabbs[1, 0] = np.inf
    elif damage == "nan_vertex":
        verts[int(arrays.faces[3 * 17]), 1] = np.nan
    elif damage == "child_outside_parent":
        # finite, lo <= hi, but the node's box is now a sliver its children stick out of
        aabbs[2 * inner + 1, 0] = aabbs[2 * inner, 0] + 1.0e-3
    elif damage == "collinear_triangle":
        # two equal corners: D = 0, so the record carries no reciprocal of it (TriRec::inv_d is a NaN, tri_predicate.h)
        for face in (17, 40, 41):
            faces[3 * face + 2] = faces[3 * face + 1]
    damaged = orc.SceneArrays(faces, nodes, aabbs, verts, arrays.normals)
    opt = rt.Options.defaults(width=96, height=64, n_super_samples=1, ao_num_samples=2, ao_max_distance=0.5)
    host = rt.Host(opt, 0)
    host.upload(damaged.faces, damaged.nodes, damaged.aabbs, damaged.vertices, damaged.normals)
    host.render()
    ref_img, counters, _ = oracle.render(orc.params_from_options(opt), damaged)
    got = host.download()
    same = (bits(got) == bits(ref_img)) | (np.isnan(got) & np.isnan(ref_img))
    assert same.all(), int((~same).sum())
    assert host.stats()["primary_hits"] == counters["primary_hits"]
    host.close()


def test_zero_direction_components(rt, oracle, scene_for):
    """Odd image sizes put a sub-pixel exactly on the optical axis: dx or dy is 0 and its
    reciprocal infinite.  The walk picks near/far planes by the reciprocal's sign like the
    reference, so (b - o) * inf and the NaN of 0 * inf must come out as they do there."""
    import orc

    scene, arrays = scene_for("blob", "longest")
    opt = rt.Options.defaults(width=97, height=65, n_super_samples=1, ao_num_samples=3)
    host = render_hip(rt, scene, opt)
    ref_img, counters, _ = oracle.render(orc.params_from_options(opt), arrays)
    assert np.array_equal(bits(host.download()), bits(ref_img))
    assert host.stats()["ao_occluded"] == counters["ao_occluded"]
    host.close()


@pytest.mark.parametrize("name", ["blob_128x96_s4_a3", "ties_64_s4_a3", "bunny_256_s1_a3"])
def test_lane_by_lane_walk_still_matches(rt_knobs, golden, scene_for_knobs, name, monkeypatch):
    """OCRT_NO_SHARED_WALK=1 selects the kernels in which every lane walks the
    tree on its own (the first generation; only the A/B build of the library holds it)."""
    monkeypatch.setenv("OCRT_NO_SHARED_WALK", "1")
    rt = rt_knobs
    case = golden["renders"][name]
    scene, _ = scene_for_knobs(case["mesh"], case["bvh"])
    host = render_hip(rt, scene, options_for(rt, case))
    assert host.stats()["ao_occluded"] == case["counters"]["ao_occluded"]
    assert hashlib.sha256(host.download().tobytes()).hexdigest() == case["float_sha256"]
    host.close()


@pytest.mark.parametrize("knobs", [
    {"OCRT_BATCH_BELOW": "0"},          # every leaf tested on the spot
    {"OCRT_BATCH_BELOW": "65"},         # every triangle test deferred and batched
    {"OCRT_AO_CLAIM_MAX": "1"},         # one direction per claim
    {"OCRT_AO_CLAIM_MAX": "28", "OCRT_AO_GUIDE": "1000"},
    {"OCRT_COST_SHIFT": "0"},           # finest ordering keys
    {"OCRT_NO_SORT": "1"},              # blocks in spatial order
    {"OCRT_FORCE_EXACT_WALK": "1"},     # per-lane cursors + select-based slab test in every packet
    {"OCRT_AO_BLOCKS": "3"},            # three workgroups do the whole AO pass
    {"OCRT_KEEP_TREE": "1"},            # walk the uploaded tree instead of the rebuilt one
    {"OCRT_KEEP_TREE": "1", "OCRT_NO_SHARED_WALK": "1"},
    {"OCRT_ENTRY_PER_TILE": "1"},       # the tiles' own walk intervals only (what a frame with too large a table gets)
    {"OCRT_ENTRY_PER_TILE": "1", "OCRT_KEEP_TREE": "1"},
])
def test_scheduling_knobs_do_not_change_the_image(rt_knobs, golden, scene_for_knobs, knobs, monkeypatch):
    """Claim sizes, tile order, batching thresholds and the form of the walk only change who does what when.  (The
    knobs exist in the A/B build of the library only.)"""
    for key, value in knobs.items():
        monkeypatch.setenv(key, value)
    rt = rt_knobs
    for name in ("bunny_256_s1_a3", "bunny_101x77_s9_a2", "ties_64_s4_a3"):
        case = golden["renders"][name]
        scene, _ = scene_for_knobs(case["mesh"], case["bvh"])
        host = render_hip(rt, scene, options_for(rt, case))
        assert hashlib.sha256(host.download().tobytes()).hexdigest() == case["float_sha256"], (name, knobs)
        assert host.stats()["ao_occluded"] == case["counters"]["ao_occluded"]
        host.close()



def test_product_library_ignores_the_knobs(rt, golden, scene_for, monkeypatch):
    """The product library reads no scheduling knob from the environment: with OCRT_AO_BLOCKS=3 and the first
    generation asked for, a frame takes the time it always takes (three workgroups would need 100x as long)."""
    c = golden["renders"]["bunny_600_defaults"]
    scene, _ = scene_for(c["mesh"], c["bvh"])
    plain = render_hip(rt, scene, options_for(rt, c))
    plain.render()
    monkeypatch.setenv("OCRT_AO_BLOCKS", "3")
    monkeypatch.setenv("OCRT_NO_SHARED_WALK", "1")
    knobbed = render_hip(rt, scene, options_for(rt, c))
    knobbed.render()
    assert knobbed.last_kernel_ms < 3.0 * plain.last_kernel_ms + 0.5
    assert hashlib.sha256(knobbed.download().tobytes()).hexdigest() == c["float_sha256"]
    import subprocess

    from conftest import ROOT
    import os

    symbols = subprocess.run(["nm", "-C", os.path.join(ROOT, "opencl_raytracer_amd", "lib", "libocrt_hip.so")],
                             capture_output=True, text=True).stdout
    assert "primary_kernel<true>" in symbols and "primary_kernel<false>" not in symbols
    plain.close()
    knobbed.close()


FUZZ_SLICE = 64


@pytest.mark.parametrize("index", range(FUZZ_SLICE))
def test_fuzz_slice(rt, rt_knobs, oracle, index):
    """A seeded slice of tools/fuzz_parity.py (random size, supersampling, AO rings / distance / angles incl. distances
    the scaled node test refuses, focal length, mesh incl. bunny and the interior scene, tree, hosts that share the
    GPU, rings of hosts replaying their graph): floats and statistics equal the oracle's bit for bit.  Cases without a
    scheduling knob run on the product library, the others on the A/B build."""
    import random

    import orc
    from tools import fuzz_parity

    rng = random.Random(20261004)
    for _ in range(index + 1):
        case = fuzz_parity.draw_case(rng)
    binding = rt_knobs if case["knobs"] else rt
    scenes = _FUZZ_SCENES.setdefault(id(binding), {})
    same, stats_ok = fuzz_parity.run_case(binding, orc, oracle, scenes, case)
    assert same and stats_ok, case


_FUZZ_SCENES = {}


def test_three_renderers_in_flight_give_the_golden_frames(rt, golden, scene_for):
    """bench.py keeps three renderers of one scene busy on one GPU, frames enqueued in turn on their own streams
    (different stream priorities: different hardware queues) and collected later -- rt.FrameRing is that as a class:
    every frame must still be the golden one, and the streams the hosts report must be three different ones."""
    c = golden["renders"]["bunny_1080p_s1_a3"]
    opt = options_for(rt, c)
    scene, _ = scene_for(c["mesh"], c["bvh"])
    ring = rt.FrameRing(opt, scene, hosts=3)
    assert len({h.stream_handle for h in ring.hosts}) == 3 and all(h.stream_handle for h in ring.hosts)
    frames = []
    for frame in range(9):  # three in flight at any time from the third on
        if frame >= 3:
            frames.append(ring.collect())
        ring.submit()
    with pytest.raises(RuntimeError):
        ring.submit()
    while len(frames) < 9:
        frames.append(ring.collect())
    with pytest.raises(RuntimeError):
        ring.collect()
    for u8 in frames:
        assert hashlib.md5(rt.pgm_bytes(u8)).hexdigest() == c["pgm_md5"]
    for h in ring.hosts:
        assert hashlib.sha256(h.download().tobytes()).hexdigest() == c["float_sha256"]
        st = h.stats()
        assert st["primary_hits"] == c["counters"]["primary_hits"] and st["ao_occluded"] == c["counters"]["ao_occluded"]
    ring.close()


def _device_check(tmp_path, source):
    """Compiles one of the tests' device programs against the product's headers and runs it: its JSON line, exit status."""
    import json
    import os
    import subprocess

    from conftest import ROOT

    exe = tmp_path / "check"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-w", "-I",
                    os.path.join(ROOT, "opencl_raytracer_amd", "csrc"), "-o", str(exe), os.path.join(ROOT, "tests", source)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode in (0, 1), r.stderr[-1000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), r.returncode


def test_fast_reciprocal_is_exact_for_every_float(tmp_path):
    """csrc/exact_reciprocal.h (the ray set-up's 1 / direction): v_rcp_f32 + one Newton step in fma form has the bits of
    the IEEE division for every float its gate lets through, and the gate lets through exactly the biased exponents
    1 ... 252 -- all 2^32 bit patterns, on the device (tests/reciprocal_check.hip)."""
    out, status = _device_check(tmp_path, "reciprocal_check.hip")
    assert status == 0 and out["wrong_bits"] == 0 and out["gate_differs"] == 0, out
    assert out["inputs"] == 2 ** 32 and out["let_through"] == 2 * 252 * 2 ** 23


def test_triangle_zones_never_contradict_the_divisions(tmp_path):
    """csrc/tri_predicate.h (the any-hit triangle test): wherever the products by TriRec::inv_d decide, the reference's
    two divisions decide the same -- 2^32 (X, Y, D) triples, most of them within 1e-10 ... 1e-4 of a threshold or a zone
    edge, plus infinities, NaNs, subnormals and unusable D (tests/tri_predicate_check.hip).  At least a third of the
    triples must have been decided without a division, or the sweep says nothing about the short form."""
    out, status = _device_check(tmp_path, "tri_predicate_check.hip")
    assert status == 0 and out["said_in_but_rejected"] == 0 and out["said_out_but_accepted"] == 0 and out["nan_inverse_decided"] == 0, out
    assert out["triples"] == 2 ** 32 and out["decided_without_division"] > out["triples"] // 3 and out["undecided"] > 2 ** 24
