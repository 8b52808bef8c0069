"""CPU, world_size 2 (and 3) over gloo: the band partition + single gather that
bench.py uses on N GPUs, with the oracle standing in for each rank's renderer.
The assembled image must equal the unpartitioned frame byte for byte."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import opencl_raytracer_amd as rt
        import orc
        from conftest import mesh_file, options_for
        from opencl_raytracer_amd.multi_gpu import BandLayout, gather_bands

        opt = options_for(rt, case)
        scene = rt.Scene.load_off(mesh_file(case["mesh"])).build_bvh(opt.bvh_method)
        arrays = orc.SceneArrays.from_scene(scene)
        oracle = orc.Oracle()
        p = orc.params_from_options(opt)
        n = int(p.height // opt.height)
        layout = BandLayout(opt, world)
        rows = layout.rows[rank]
        # this rank renders only the sub-pixel rows of its own bands ...
        image = np.zeros((p.height, p.width), dtype=np.float32)
        own = rows[rows < opt.height]
        for y in own:
            oracle.render(p, arrays, rows=(int(y) * n, int(y) * n + n), image=image, nthreads=2)
        # ... and box-filters them into its compact band buffer
        full_u8 = oracle.resize(image, opt.width, opt.height, opt.n_super_samples)
        band = np.zeros((layout.max_rows, opt.width), dtype=np.uint8)
        band[: own.size] = full_u8[own]
        final = gather_bands(torch.from_numpy(band), layout, rank)
        if rank == 0:
            np.save(out_path, final.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name", [(2, "blob_128x96_s4_a3"), (2, "bunny_101x77_s9_a2"), (3, "ties_64_s4_a3")])
def test_gloo_band_gather_reassembles_frame(golden, tmp_path, world, name):
    case = golden["renders"][name]
    out = str(tmp_path / "final.npy")
    mp.spawn(_worker, args=(world, _free_port(), case, out), nprocs=world, join=True)
    final = np.load(out)
    header = f"P5 {case['width']} {case['height']} 255\n".encode()
    assert hashlib.md5(header + final.tobytes()).hexdigest() == case["pgm_md5"]


def _pipelined_worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import opencl_raytracer_amd as rt
        from opencl_raytracer_amd.multi_gpu import BandGatherer, BandLayout

        opt = rt.Options.defaults(width=37, height=29, n_super_samples=4)
        layout = BandLayout(opt, world)
        rows = layout.rows[rank]
        own = rows[rows < opt.height]
        gatherers = [BandGatherer(layout, rank, "cpu") for _ in range(2)]
        bands = [torch.zeros((layout.max_rows, opt.width), dtype=torch.uint8) for _ in range(2)]
        finals, open_slot = [], None
        for frame in range(5):  # bench.py's order: start frame i, then finish frame i - 1
            k = frame & 1
            bands[k].zero_()
            bands[k][: own.size] = torch.from_numpy(((own[:, None] * 7 + np.arange(opt.width)[None, :] + frame) % 251).astype(np.uint8))
            gatherers[k].start(bands[k])
            if open_slot is not None:
                final = gatherers[open_slot].finish()
                finals.append(None if final is None else final.clone())
            open_slot = k
        final = gatherers[open_slot].finish()
        finals.append(None if final is None else final.clone())
        with pytest.raises(RuntimeError):
            gatherers[0].finish()  # nothing pending
        if rank == 0:
            np.save(out_path, np.stack([f.numpy() for f in finals]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_pipelined_gather_keeps_frames_apart(tmp_path, world):
    """bench.py keeps two frames in flight (two band buffers, two gatherers: start i, finish i - 1): every frame must
    still come out whole and in order."""
    out = str(tmp_path / "frames.npy")
    mp.spawn(_pipelined_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    frames = np.load(out)
    assert frames.shape == (5, 29, 37)
    y, x = np.mgrid[0:29, 0:37]
    for frame in range(5):
        assert np.array_equal(frames[frame], ((y * 7 + x + frame) % 251).astype(np.uint8)), frame
