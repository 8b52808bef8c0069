"""Driver of tests/test_fake_rccl.py::test_a_rank_that_posts_one_frame_fewer_does_not_hang_the_others (run with the
stand-in librccl.so.1 in front of the loader's path and FAKE_RCCL_HANG_ON_MISSING_SEND=1): two ranks of a frame in this
one process, rings with the exchange step attached; rank 1 renders one frame FEWER than rank 0.  Rank 0's last receive
then waits for a send that never comes -- exactly what a dead or out-of-step peer looks like.  The ring's bounded wait
must turn that into an error within its deadline (set to 2 s here; 30 s by default) instead of sleeping for ever inside
the library.  Prints what happened as JSON."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import opencl_raytracer_amd as rt  # noqa: E402
from conftest import mesh_file, options_for  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
    c = json.load(f)["renders"]["blob_128x96_s4_a3"]
opt = options_for(rt, c)
scene = rt.Scene.load_off(mesh_file(c["mesh"])).build_bvh(0)
rings = [rt.FrameRing(opt, scene, 0, r, 2, hosts=2) for r in range(2)]
uid = rt.rccl_unique_id()
for ring in rings:
    ring.attach_rccl(uid)
    ring.set_gather_timeout(2.0)
info = rings[0].rccl_info()
frames = 4
for frame in range(frames):
    if frame < frames - 1:  # rank 1 stops one frame early
        rings[1].submit()
        rings[1].collect_info()
    rings[0].submit()
    rings[0].collect_info()  # (enqueues rank 0's receive; the last one finds no send)
t0 = time.perf_counter()
error = None
try:
    rings[0].drain()  # waits for the gathers
except rt.RtError as e:
    error = str(e)
waited = time.perf_counter() - t0
t1 = time.perf_counter()
for ring in rings:
    ring.close()  # must not hang either: the broken communicator is aborted, not waited for
print(json.dumps({"error": error, "waited_s": waited, "close_s": time.perf_counter() - t1, "comm_ranks": info[0], "rccl_version": info[1]}))
