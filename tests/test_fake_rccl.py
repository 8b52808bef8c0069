"""The RCCL exchange step with SEVERAL ranks on the one GPU of the box.  The real RCCL refuses two ranks on one device,
so on a one-GPU box `render --gpus N --gather rccl` could only ever run with N = 1; here a stand-in library
(tests/fake_rccl/fake_rccl.cc: grouped ncclSend / ncclRecv pairs become stream-ordered device copies, a send without
its receive fails) is put in front of the loader's path, and the product's own code -- ocrt::GroupGather: which band
buffer, which offset of the receive buffer, how many bytes, which peer, which stream, one group per frame; the row
assembly kernel behind it -- runs for 2, 3, 5 and 8 ranks through the CLI; and the frame rings' exchange step
(ocrt::BandGather, what bench.py's ranks use: ncclCommInitRank per rank, per frame one ncclSend on every rank but the
first and the receives + row assembly on rank 0, behind the frames rendered next) runs with every rank of the frame in
one process.  What this proves is the product's side of the exchange; RCCL itself is exercised at world 1 (and 2 where
two GPUs exist) by the other tests."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, mesh_file

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fake_dir(tmp_path_factory):
    out = tmp_path_factory.mktemp("fake_rccl")
    lib = out / "librccl.so.1"
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wl,-soname,librccl.so.1", "-o", str(lib),
                    os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cc")], check=True)
    return str(out)


@pytest.mark.parametrize("name,ranks", [("bunny_600_defaults", 8), ("bunny_101x77_s9_a2", 3), ("blob_128x96_s4_a3", 2),
                                        ("bunny_256_s1_a3", 5), ("bunny_1080p_s1_a3", 8)])
def test_group_gather_with_many_ranks_on_one_gpu(golden, tmp_path, fake_dir, name, ranks):
    c = golden["renders"][name]
    exe = os.path.join(ROOT, "opencl_raytracer_amd", "bin", "render")
    out = tmp_path / (name + ".pgm")
    env = dict(os.environ, OCRT_SHARE_DEVICES="1", LD_LIBRARY_PATH=fake_dir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    cmd = [exe, "-w", str(c["width"]), "-h", str(c["height"]), "-s", str(c["ss"]), "-a", str(c["ao"]), "-d", str(c["aod"]),
           "-f", str(c["focal"]), "--gpus", str(ranks), "--gather", "rccl", mesh_file(c["mesh"]), str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert f"fake rccl: {ranks} ranks in one process" in r.stderr  # the stand-in was the library that ran
    assert "over RCCL" in r.stdout and f"Rank {ranks - 1} of {ranks}" in r.stdout
    assert f"{ranks - 1} send/receive pairs" in r.stderr  # one frame: every other rank's bands, in one group
    assert hashlib.md5(out.read_bytes()).hexdigest() == c["pgm_md5"]


# (ranks, hosts per rank, frames: bench.py takes 6 hosts per GPU at 2 - 7 ranks and 8 from eight on; more frames than the
# 2 x hosts band buffers of a ring, so that every slot is bound, gathered from and reused)
@pytest.mark.parametrize("name,ranks,hosts,frames", [("bunny_600_defaults", 8, 8, 20), ("bunny_101x77_s9_a2", 3, 2, 7),
                                                     ("blob_128x96_s4_a3", 2, 6, 15), ("blob_128x96_s4_a3", 2, 1, 7),
                                                     ("bunny_1080p_s1_a3", 4, 3, 7)])
def test_ring_exchange_step_with_every_rank_in_one_process(golden, fake_dir, name, ranks, hosts, frames):
    c = golden["renders"][name]
    env = dict(os.environ, LD_LIBRARY_PATH=fake_dir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl", "ring_ranks_driver.py"), name, str(ranks), str(hosts),
                        str(frames)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    assert f"fake rccl: communicator of {ranks} ranks" in r.stderr and "never received" not in r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["md5"] == [c["pgm_md5"]] * frames  # every frame, with the next ones already in flight behind it
    assert out["primary_hits"] == c["counters"]["primary_hits"] and out["ao_occluded"] == c["counters"]["ao_occluded"]
    assert out["last_image_elsewhere"] == 0  # the assembled image lives on rank 0 only


def test_a_rank_that_posts_one_frame_fewer_does_not_hang_the_others(fake_dir):
    """The exchange step's liveness guard (band_gather.cc, BandGather::wait): a peer that died or is a frame out of step
    leaves rank 0's receive waiting for ever; the ring must fail with a device error within its deadline -- so that the
    rank exits non-zero and the job ends -- and its close() must not wait for the operation that cannot finish."""
    env = dict(os.environ, LD_LIBRARY_PATH=fake_dir + ":" + os.environ.get("LD_LIBRARY_PATH", ""), FAKE_RCCL_HANG_ON_MISSING_SEND="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl", "ring_missing_frame_driver.py")], capture_output=True,
                       text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    assert "finds no posted send: blocking its stream" in r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["error"] and "waited 2 s for the exchange step" in out["error"] and "peer rank is gone" in out["error"]
    assert 1.9 < out["waited_s"] < 30.0 and out["close_s"] < 30.0
    assert "aborted" in r.stderr  # ncclCommAbort, not ncclCommDestroy
    assert out["comm_ranks"] == 2  # what the communicator says about itself (rt_ring_rccl_info)
