#!/usr/bin/env python3
"""bench.py -- Mrays/s of the ray-casting hot path on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

One "step" renders one full frame of the workload: every rank casts the primary
and ambient-occlusion rays of its image bands (HIP kernel), box-filters them to
8-bit on the device, and -- for N > 1 -- the bands are gathered to rank 0 over
RCCL and assembled into the final image.  Scene and image buffers are resident
in HBM before the timed region starts.  Rays are counted as the reference would
cast them: one primary ray per sub-pixel plus, with AO on, 28 (default ring
set) any-hit rays per hit sub-pixel.

`--gpus N` with N > 1 from a bare shell starts N ranks itself (a child
`python -m torch.distributed.run`, before this process touches torch or the GPU);
under an external torchrun (RANK / WORLD_SIZE in the environment) it is a rank.

The frames go through the library's frame ring (include/rt_hip_ring.h, rt_ring_*): several
render hosts per GPU take them in turn, each replaying its captured hipGraph, and the
ring itself runs the RCCL gather behind the next frames -- a timed block is ONE call
into the library (rt_ring_run), Python is not on the per-frame path.  In the same run
the frame is also timed ONE AT A TIME (a ring of one host = the reference's blocking
OpenCLHost::operator()(), src/opencl_host.cc:137-149): `blocking` in the line.  Every
block of K steps is bracketed by barrier + synchronize and repeated until half a second
of GPU time is covered; the line reports the median block, with min / max beside it.

Rank 0 prints ONE JSON line (contract in the task description) carrying
`roofline` -- the bound the dominant kernel is actually under (vector-instruction
issue; the 12 MB scene is cache-resident): that kernel's instructions per launch over
its launch duration, HIP events around the launch with ONE frame at a time; the
pipelined frame rate in the same units under `roofline.frame_pipelined`; the
contractual HBM line of SURVEY.md 8d (algorithmic bytes of the REFERENCE traversal over
the same kernel time, and the HBM bytes the PMC counters really saw) under
`roofline.hbm` -- and, at N = 1, `cpu_baseline` (the reference's own kernel compiled
for x86-64 when oracle/_ref/ holds it, else this repo's C restatement, on the host
cores) and `end_to_end` (the `render` CLI as a child process: every one-off cost).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import re
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Vector-instruction issue (MI355X_MICROARCH.md, "Wave scheduling"): a SIMD-32 issues a wave64 VALU instruction over
# 2 cycles, i.e. at most 0.5 per clock per SIMD (a lone wave sustains 0.25); 256 CUs x 4 SIMDs at 2.4 GHz.
VALU_PEAK_PER_CLK_SIMD = 0.5
SIMDS, CLOCK_HZ = 1024, 2.4e9

# Workloads = the configs of BASELINE.json that fit one GPU; `golden` names the
# tests/golden/golden.json entry that pins the output and supplies the
# reference-traversal counters for the algorithmic byte count.
WORKLOADS = {
    "bunny_1080p_ao": dict(
        mesh="bunny", bvh="longest", width=1920, height=1080, ss=1, ao=3, golden="bunny_1080p_s1_a3",
        label="bunny.off 1920x1080 -s 1 -a 3 (primary + 28 AO rays per hit sub-pixel), longest-axis BVH"),
    "bunny_1080p_primary": dict(
        mesh="bunny", bvh="longest", width=1920, height=1080, ss=1, ao=0, golden="bunny_1080p_s1_a0",
        label="bunny.off 1920x1080 -s 1 -a 0 (primary rays only), longest-axis BVH"),
    "bunny_600_defaults": dict(
        mesh="bunny", bvh="longest", width=600, height=600, ss=4, ao=3, golden="bunny_600_defaults",
        label="bunny.off 600x600 CLI defaults (-s 4 -a 3)"),
    "bunny_1080p_s4": dict(
        mesh="bunny", bvh="longest", width=1920, height=1080, ss=4, ao=3, golden=None,
        label="bunny.off 1920x1080 -s 4 -a 3 (2x2 supersample grid: 3840x2160 sub-pixels)"),
    "bunny_1080p_s16": dict(
        mesh="bunny", bvh="longest", width=1920, height=1080, ss=16, ao=3, golden=None,
        label="bunny.off 1920x1080 -s 16 -a 3 (4x4 supersample grid: 7680x4320 sub-pixels)"),
    "bunny_1080p_s64": dict(
        mesh="bunny", bvh="longest", width=1920, height=1080, ss=64, ao=3, golden="bunny_1080p_s64_a3",
        label="bunny.off 1920x1080 -s 64 -a 3 (regular 8x8 supersample grid: 15360x8640 sub-pixels)"),
    "interior_4k_ao": dict(
        mesh="interior", bvh="longest", width=3840, height=2160, ss=1, ao=3, golden="interior_4k_s1_a3",
        label="interior stand-in for the missing sibenik.off, 3840x2160 -s 1 -a 3"),
    # the HARDER stand-in (tools/make_interior_mesh.py --hard): the nave's shell as a handful of huge triangles, long thin ones
    # (mullions, steps, ribs), ornament 100 x denser than the shell -- what a midpoint-split BVH copes worst with
    "interior_hard_1080p_ao": dict(
        mesh="interior_hard", bvh="longest", width=1920, height=1080, ss=1, ao=3, golden="interior_hard_1080p_s1_a3",
        label="HARDER interior stand-in for the missing sibenik.off (huge triangles beside fine ornament, slivers), 1920x1080 -s 1 -a 3"),
    "interior_hard_4k_ao": dict(
        mesh="interior_hard", bvh="longest", width=3840, height=2160, ss=1, ao=3, golden="interior_hard_4k_s1_a3",
        label="HARDER interior stand-in for the missing sibenik.off (huge triangles beside fine ornament, slivers), 3840x2160 -s 1 -a 3"),
    "interior_1080p_ao": dict(
        mesh="interior", bvh="longest", width=1920, height=1080, ss=1, ao=3, golden="interior_1080p_s1_a3",
        label="interior stand-in for the missing sibenik.off, 1920x1080 -s 1 -a 3"),
    # Generated height fields (tools/big_meshes.py; in memory, no golden frame: checked against the oracle at a small
    # resolution before anything is timed).  The packed scenes -- 0.6 GB and 6 GB -- are far beyond the 32 MB of L2 and the
    # 256 MB Infinity Cache: the workloads on which north_star's memory roofline means something.
    "terrain_2m_1080p_ao": dict(
        mesh="terrain:1000", bvh="longest", width=1920, height=1080, ss=1, ao=3, golden=None,
        label="generated height field, 2.0 M triangles, 1920x1080 -s 1 -a 3"),
    "terrain_20m_1080p_ao": dict(
        mesh="terrain:3200:4", bvh="longest", width=1920, height=1080, ss=1, ao=3, golden=None,
        label="generated height field, 20.5 M triangles (four times the size of the 2 M one: the reference's triangle test does not see triangles below ~5e-7 units of area), 1920x1080 -s 1 -a 3"),
}
DEFAULT_WORKLOAD = "bunny_1080p_ao"


def workload_options(rt, w):
    return rt.Options.defaults(width=w["width"], height=w["height"], n_super_samples=w["ss"], ao_num_samples=w["ao"],
                               enable_ao=int(w["ao"] != 0), bvh_method=0 if w["bvh"] == "longest" else 1)


def mesh_path(name: str) -> str:
    from tools.meshes import bunny_path, interior_hard_path, interior_path

    return bunny_path() if name == "bunny" else interior_hard_path() if name == "interior_hard" else interior_path()


def load_scene(rt, w):
    """The workload's mesh as an rt.Scene (no BVH yet): an OFF file, or a mesh generated in memory ("terrain:<n>")."""
    if w["mesh"].startswith("terrain:"):
        from tools.big_meshes import terrain

        parts = w["mesh"].split(":")  # terrain:<n>[:<scale about the camera>]
        return rt.Scene.from_arrays(*terrain(int(parts[1]), scale=float(parts[2]) if len(parts) > 2 else 1.0))
    return rt.Scene.load_off(mesh_path(w["mesh"]))


def algorithmic_bytes(counters: dict, subpixels: int) -> dict:
    """B = 36 per node visit (4 B count + 32 B box) + 60 per
    triangle test (12 B indices + 48 B vertices) + 48 per hit primary (normals)
    + 4 per sub-pixel written, with the visit/test counts of the REFERENCE
    traversal on the same tree (tests/golden/golden.json, produced by the oracle
    and identical to the reference kernel's walk).  Split by pass: the AO pass
    (the dominant kernel) owns the AO rays' visits and tests."""
    ao = 36 * counters["ao_node_visits"] + 60 * counters["ao_tri_tests"]
    primary = (36 * counters["primary_node_visits"] + 60 * counters["primary_tri_tests"]
               + 48 * counters["primary_hits"] + 4 * subpixels)
    return {"ao": ao, "primary": primary, "frame": ao + primary}


def kernel_source_sha() -> str:
    """Identifies the kernels the PMC counters under profiles/ were collected for: sha256 over the sources that
    decide what a launch executes (kernels, record layouts, scene packing, the walk tree)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "opencl_raytracer_amd", "csrc")
    parts = sorted(os.path.join("kernels", n) for n in os.listdir(os.path.join(csrc, "kernels")) if n.endswith(".h"))
    for name in ["kernels.hip"] + parts + ["device_types.h", "tri_predicate.h", "exact_reciprocal.h", "scene_pack.cc", "walk_tree.cc"]:
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_for(workload: str):
    """PMC counters of `workload`'s dominant kernel from profiles/pmc.json (collected by tools/pmc_collect.sh in
    separate rocprofv3 --pmc passes, corrected as MI355X_MICROARCH.md prescribes) -- or None when they were
    collected for other kernel sources than the ones in this tree."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc.json")) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return None
    entry = pmc.get("workloads", {}).get(workload)
    if entry is None or pmc.get("kernel_source_sha256") != kernel_source_sha():
        return None
    return dict(entry, source=pmc.get("source"))


def host_cores() -> int:
    """CPU threads this process may really use: the affinity mask capped by the
    cgroup CPU quota (the GPU boxes show 256 CPUs but grant 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def end_to_end(w, golden_md5):
    """One `render` run as a child process, BEFORE this process touches the GPU: the whole drop-in CLI with every
    one-off cost (HIP start-up overlapped with mesh loading and BVH build, scene packing + upload, one blocking frame,
    device resize + download, PGM write), by the CLI's own clock (--timings) and by ours around the process."""
    exe = os.path.join(ROOT, "opencl_raytracer_amd", "bin", "render")
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"bench_end_to_end_{os.getpid()}.pgm")
    cmd = [exe, "-w", str(w["width"]), "-h", str(w["height"]), "-s", str(w["ss"]), "-a", str(w["ao"]), "-r", w["bvh"],
           "--timings", "1", mesh_path(w["mesh"]), out]
    result = {"command": " ".join(["render"] + cmd[1:-2] + [os.path.basename(cmd[-2]), "out.pgm"])}
    try:
        runs = []
        for _ in range(3):  # (the later runs find the files and the driver warm; how long the device takes to come up varies)
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            wall = (time.perf_counter() - t0) * 1e3
            if r.returncode != 0:
                return dict(result, error=(r.stdout + r.stderr)[-300:])
            m = re.search(r"Timings \(ms\):(.*)", r.stdout)
            phases = {}
            if m:
                for name, value in re.findall(r"\s*([^,]+?) ([0-9.]+),", m.group(1)):
                    phases[name.strip().replace(" ", "_").replace("+", "_") + "_ms"] = float(value)
                tail = re.search(r"wall ([0-9.]+)", m.group(1))
                if tail:
                    phases["wall_in_process_ms"] = float(tail.group(1))
            with open(out, "rb") as f:
                md5 = hashlib.md5(f.read()).hexdigest()
            runs.append(dict(phases, process_wall_ms=round(wall, 1), pgm_md5=md5))
        os.remove(out)
        best = min(runs, key=lambda x: x["process_wall_ms"])
        result.update(best, first_run_process_wall_ms=runs[0]["process_wall_ms"],
                      pgm_matches_golden=(best["pgm_md5"] == golden_md5) if golden_md5 else None)
    except (OSError, subprocess.SubprocessError) as e:
        result["error"] = str(e)[-300:]
    return result


def cpu_baseline(opt, scene, gpu_u8, w):
    """Times the CPU checker on the same frame (rank 0, N = 1 only) and checks
    the GPU image against it.  The oracle is used here as the thing to compare
    with and to time beside -- never as part of the measured GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np

    import orc

    p = orc.params_from_options(opt)
    arrays = orc.SceneArrays.from_scene(scene)
    cores = host_cores()
    full_rays = None
    # Bound the sample to roughly 10-30 core-seconds: a band of rows in the
    # middle of the image (where the model is) for the heavy workloads.
    rows = (0, p.height)
    est_subpixels = p.width * p.height * (29 if opt.enable_ao else 1)
    if est_subpixels > 80e6:
        keep = max(8, int(p.height * 80e6 / est_subpixels) // 8 * 8)
        y0 = (p.height - keep) // 2 // 8 * 8
        rows = (y0, y0 + keep)
    lib = orc.ref_kernel(p, opt.n_super_samples, build=False)
    oracle = orc.Oracle()
    # ray count of the sample from the oracle's counters (cheap relative to AO)
    t0 = time.perf_counter()
    orc_img, counters, used = oracle.render(p, arrays, rows=rows, nthreads=cores)
    t_port = time.perf_counter() - t0
    rays = counters["primary_rays"] + counters["ao_rays"]
    kind, seconds, img = "port", t_port, orc_img
    if lib is not None:
        t0 = time.perf_counter()
        ref_img, used = orc.ref_render(lib, p, arrays, rows=rows, nthreads=cores)
        seconds = time.perf_counter() - t0
        kind, img = "reference", ref_img
        if not np.array_equal(ref_img.view(np.uint32), orc_img.view(np.uint32)):
            raise AssertionError("reference kernel and oracle disagree on the baseline sample")
    # the sample's sub-pixel rows -> PGM rows (the band starts and ends on multiples of 8 sub-pixel rows, so on
    # whole output rows), compared byte for byte with the same rows of the GPU frame
    n = int(np.sqrt(opt.n_super_samples))
    band = np.ascontiguousarray(img[rows[0]:rows[1]])
    cpu_u8 = oracle.resize(band, opt.width, (rows[1] - rows[0]) // n, opt.n_super_samples)
    identical = bool(np.array_equal(cpu_u8, gpu_u8[rows[0] // n:rows[1] // n]))
    if not identical:
        raise AssertionError("GPU PGM differs from the CPU baseline image")
    if rows == (0, p.height):
        full_rays = rays
    return {
        "value": round(rays / seconds / 1e6, 3), "unit": "Mrays/s", "cores": int(used), "kind": kind, "cpu_model": cpu_model(),
        "sample": f"rows {rows[0]}..{rows[1]} of {p.height} ({rays} rays, {seconds:.2f} s wall)"
                  + ("; PGM byte-identical to the GPU frame" if rows == (0, p.height) else "; these PGM rows byte-identical to the GPU frame's"),
        "port_value": round(rays / t_port / 1e6, 3),
    }, full_rays


def summary(values):
    return {"median": round(statistics.median(values), 4), "min": round(min(values), 4), "max": round(max(values), 4), "n": len(values)}


def main():
    # The hosts' driver shares device memory between processes by dmabuf only: RCCL's peer-to-peer set-up (and torch's
    # sharing of device tensors) fails with `hipIpcGetMemHandle: invalid argument` under the legacy IPC mode.  The variable
    # is exported on the boxes this runs on; set here too, before the HIP runtime starts, for a shell that lacks it (the
    # ranks a bare `--gpus N` starts inherit it).
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # (a quarter of a second of frames per block)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--in-flight", type=int, default=0, choices=list(range(17)),
                    help="render hosts per GPU taking the frames in turn (1: one frame at a time; 0 = by the number of "
                         "ranks: 3 on one GPU, 6 on two to seven, 8 from eight on)")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="repeat each block of --steps steps until this much time is covered")
    ap.add_argument("--plain-launches", action="store_true", help="launch the kernels one by one instead of replaying the captured graph")
    ap.add_argument("--pacing", type=float, default=-1.0, help="the ring's pacing factor (rt_ring_set_pacing); default: the library's")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ  # under torch.distributed.run
    if not launched and args.gpus > 1:
        # A bare `python bench.py --gpus N`: start the N ranks as FRESH processes.  This process has not imported
        # torch or touched the GPU yet (and must not: a process that initialised HIP may not exec or fork ranks).
        import socket

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)  # rank 0's JSON line goes straight to our stdout
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    # Every step of the set-up is named, logged (stderr, with the time since the step before) and TIME-BOXED: a rank that
    # sits in one step longer than its limit says where -- a JSON line with "error" on stdout from rank 0, a line on stderr
    # from every rank -- and the process ends with a non-zero status instead of hanging until somebody's patience runs out.
    # (The first contact with an 8-GPU node is a run nobody has rehearsed on real hardware: RCCL's rendezvous, peer access and
    # the dmabuf IPC mode are the places a multi-GPU job stalls in, and one log must be enough to tell which.)
    import threading

    stage = {"name": "start", "since": time.monotonic(), "limit": 300.0, "done": False}

    def enter(name, limit=300.0):
        now = time.monotonic()
        if launched:
            print(f"[bench rank {rank}/{world}] {stage['name']} took {now - stage['since']:.2f} s; now: {name}", file=sys.stderr, flush=True)
        stage.update(name=name, since=now, limit=limit)

    def watchdog():
        while not stage["done"]:
            time.sleep(1.0)
            waited = time.monotonic() - stage["since"]
            if not stage["done"] and waited > stage["limit"]:
                message = (f"bench.py rank {rank}/{world}: step '{stage['name']}' has not finished after {waited:.0f} s "
                           f"(limit {stage['limit']:.0f} s); HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}")
                print(message, file=sys.stderr, flush=True)
                if rank == 0:
                    print(json.dumps({"error": message, "stage": stage["name"], "n_gpus": world, "rank": rank}), flush=True)
                os._exit(3)

    threading.Thread(target=watchdog, daemon=True).start()
    w = WORKLOADS[args.workload]
    golden_md5, counters = None, None
    if w["golden"]:
        with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
            g = json.load(f)["renders"][w["golden"]]
        golden_md5, counters = g["pgm_md5"], g["counters"]

    e2e = None
    if world == 1 and not args.no_end_to_end and ":" not in w["mesh"]:  # (generated meshes are not files: no CLI run)
        e2e = end_to_end(w, golden_md5)  # (a child process, before this one opens the GPU)

    import numpy as np
    import torch
    import torch.distributed as dist

    import opencl_raytracer_amd as rt

    rt.load_library()  # raises if the HIP library is missing: no fallback
    if not torch.cuda.is_available() or rt.device_count() < 1:
        sys.exit("bench.py needs a visible MI355X (no CPU fallback)")
    # One rank per GPU over RCCL.  Rehearsal knob for a one-GPU box: OCRT_BENCH_BACKEND=gloo maps every
    # rank to an existing device and stages the band gather through host memory.
    backend = os.environ.get("OCRT_BENCH_BACKEND", "nccl")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    if backend == "nccl" and world > torch.cuda.device_count():
        sys.exit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s): one rank per GPU")
    if launched:  # also for a world of one: the RCCL communicator and the gather are then exercised on a one-GPU box
        import datetime

        enter(f"torch.distributed rendezvous ({backend}; HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')})", 180.0)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=180))
        enter("first collective (an all-reduce of one word over every rank)", 120.0)
        probe = torch.ones(1, dtype=torch.int32, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(probe)
        if int(probe[0]) != world:
            sys.exit(f"bench.py: the first all-reduce counted {int(probe[0])} ranks, the job has {world}")
    reduce_device = device if backend == "nccl" else "cpu"

    enter("scene: load, BVH", 300.0)
    opt = workload_options(rt, w)
    t0 = time.perf_counter()
    scene = load_scene(rt, w)
    t_load = time.perf_counter() - t0
    scene.build_bvh(opt.bvh_method)
    t_scene = time.perf_counter() - t0

    # The frame ring (include/rt_hip_ring.h): `in_flight` render hosts of the scene on this GPU, each on its own stream
    # (consecutive hosts in different priority classes, i.e. hardware queues) with its own captured hipGraph, take the
    # frames in turn: while frame i's ambient-occlusion pass runs out (its last quarter runs at falling occupancy: the
    # queues are drained, the workgroups end one by one) the next frames' passes fill the wave slots it frees, and the
    # latency-bound primary pass runs beside a vector-issue-bound one.  The smaller a rank's share of the frame, the
    # more of it is start and end of passes, hence more hosts where the frame is split -- ms per frame on one GPU
    # (tools/analysis/queues_sweep.py, round 4's final kernels and grids, 600-frame runs, repeatable to 1 %):
    #   whole frame   2 / 3 / 4 / 6 hosts           1.042 / 0.977 / 0.996 / 1.007
    #   a half        3 / 4 / 6                     0.540 / 0.523 / 0.505
    #   a quarter     3 / 4 / 6 / 8 / 12            0.303 / 0.292 / 0.270 / 0.293 / 0.273
    #   an eighth     6 / 8 / 12 / 16               0.153 / 0.142 / 0.161 / 0.143
    # (a multi-GPU ring deals its hosts over two stream priority classes -- the third is the gather's --; that 8 and 16
    # hosts beat 12 points at the runtime's hardware queues, GPU_MAX_HW_QUEUES = 4 by default, dividing evenly or not:
    # with 16 queues the eighth reads 0.161 / 0.154 / 0.151 for 8 / 12 / 16.  The 4K and 64-samples frames do not care:
    # 0.460 / 0.462 / 0.463 and 4.78 / 4.80 / 4.69 ms for an eighth with 3 / 6 / 12.)
    in_flight = args.in_flight if args.in_flight > 0 else 3 if world == 1 else 6 if world < 8 else 8
    def make_rings():
        made = {"pipelined": rt.FrameRing(opt, scene, device_index, rank, world, hosts=in_flight),
                "blocking": rt.FrameRing(opt, scene, device_index, rank, world, hosts=1)}
        for ring in made.values():
            ring.set_graph_mode(not args.plain_launches)
            if args.pacing >= 0.0:
                ring.set_pacing(args.pacing)
        return made

    enter("frame rings: upload, tile costs, calibration, graph capture", 300.0)
    rings = make_rings()

    # The exchange step of a multi-GPU frame: the ring's own RCCL gather (one process per GPU; the unique id is made on
    # rank 0 and handed round by torch.distributed).  The gloo rehearsal on a one-GPU box cannot use RCCL (it refuses
    # two ranks on one device): there the ring writes into torch tensors and the gather is torch's, staged through host
    # memory.  Should the library's own gather fail to come up on a real multi-GPU job (RCCL not loadable, a communicator
    # that cannot be made), the job does not die without a number: every rank learns of it (an all-reduce after each step
    # of the set-up), the rings are made afresh without a gather, and torch.distributed gathers the same band buffers on
    # the device -- slower (Python takes part in every frame) and said so in the line (`config.parallelism`, `config.rccl`).
    from opencl_raytracer_amd.multi_gpu import BandGatherer, BandLayout

    layout = BandLayout(opt, world)
    assert layout.local_rows(rank) == rings["pipelined"].local_rows
    rccl = launched and backend == "nccl"
    rccl_failure = None

    def bind_bands():
        for name, ring in rings.items():
            if name not in bands:
                bands[name] = [torch.zeros((layout.max_rows, opt.width), dtype=torch.uint8, device=device) for _ in range(ring.slots)]
            for k, b in enumerate(bands[name]):
                ring.bind_output(k, b.data_ptr())

    def any_rank_failed(error: str) -> bool:
        flag = torch.tensor([1 if error else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return bool(int(flag[0]))

    bands, gatherers = {}, {}
    if launched:
        bind_bands()
    if rccl:
        # Every collective below is entered by EVERY rank, whatever went wrong on it before: what can fail on one rank alone
        # (loading RCCL, making an id, attaching a communicator) happens inside try blocks that only record the error, the
        # ranks then agree on it (an all-reduce of a flag), and only a step every rank has passed is followed by the next.
        enter("RCCL: library and unique id", 120.0)
        error = ""
        for name, ring in rings.items():
            uid = torch.zeros(128, dtype=torch.uint8, device=device)
            if rank == 0 and not error:
                try:
                    if os.environ.get("OCRT_BENCH_FAIL_RCCL"):  # rehearsal knob: take the fallback
                        raise RuntimeError("OCRT_BENCH_FAIL_RCCL is set")
                    uid = torch.frombuffer(bytearray(rt.rccl_unique_id()), dtype=torch.uint8).to(device)
                except Exception as exc:  # noqa: BLE001
                    error = f"{type(exc).__name__}: {exc}"
            enter(f"RCCL: id of the {name} ring's communicator to every rank (torch.distributed.broadcast)", 120.0)
            dist.broadcast(uid, 0)
            if any_rank_failed(error):
                break
            enter(f"RCCL: ncclCommInitRank for the {name} ring ({world} ranks)", 180.0)
            try:
                ring.attach_rccl(bytes(uid.cpu().numpy().tobytes()))
            except Exception as exc:  # noqa: BLE001
                error = f"{type(exc).__name__}: {exc}"
            if any_rank_failed(error):
                break
            enter(f"RCCL: self send / receive on the {name} ring's communicator", 120.0)
            try:
                ring.rccl_self_test()
            except Exception as exc:  # noqa: BLE001
                error = f"{type(exc).__name__}: {exc}"
            if any_rank_failed(error):
                break
        failed = any_rank_failed(error)
        if failed:
            enter("RCCL could not be set up: rings again, without a gather (torch.distributed gathers instead)", 300.0)
            rccl_failure = error or "the set-up failed on another rank"
            rccl = False
            for ring in rings.values():
                ring.close()
            rings = make_rings()
            bind_bands()
    staged = launched and not rccl
    host_staged = staged and backend != "nccl"  # (gloo: through host memory; the RCCL fallback gathers on the device)
    if launched:
        for name in rings:
            gatherers[name] = BandGatherer(layout, rank, "cpu" if host_staged else device)

    def gathered(name, slot):
        band = bands[name][slot]
        return gatherers[name](band.cpu() if host_staged else band)

    last_image = {}

    def run_steps(name, steps):
        """`steps` frames of a steady stream.  RCCL / single GPU: ONE call into the library."""
        ring = rings[name]
        if not staged:
            ring.run(steps)
            return
        hosts = ring.size
        for _ in range(steps):  # (rehearsal: the gather is torch's, so Python takes part in every frame)
            ring.submit()
            while ring.in_flight > max(1, hosts - 1) or (hosts == 1 and ring.in_flight):
                _, slot, _ = ring.collect_info()
                last_image[name] = gathered(name, slot)

    def fence(name):
        ring = rings[name]
        while staged and ring.in_flight:
            _, slot, _ = ring.collect_info()
            last_image[name] = gathered(name, slot)
        ring.drain()
        torch.cuda.synchronize(device)
        if launched:
            dist.barrier()
        torch.cuda.synchronize(device)

    def final_image(name):
        """The last frame's assembled image on rank 0 (numpy), None elsewhere."""
        if staged:
            return last_image[name].cpu().numpy() if rank == 0 else None
        if rccl and rank != 0:
            return None
        return rings[name].download_last()

    if rccl:
        # Pre-flight, before anything is timed: every band buffer of both rings once (2 x hosts frames each, the real band
        # sizes: 259 KB per rank at 1080p over eight ranks, 1 MB at 4K) through the ring's own gather, and every assembled
        # image on rank 0 against torch.distributed's gather of the same band buffers.  A peer that never posts its half
        # fails the rank within the gather's deadline (rt_ring_set_gather_timeout) instead of hanging it.
        for name, ring in rings.items():
            enter(f"pre-flight: {ring.slots} frames through the {name} ring's RCCL gather, checked against torch.distributed's", 300.0)
            ring.set_gather_timeout(60.0)
            ok = torch.ones(1, dtype=torch.int32, device=device)
            for _ in range(ring.slots):
                ring.submit()
                _, slot, _ = ring.collect_info()
                ring.drain()
                torch.cuda.synchronize(device)
                theirs = gatherers[name](bands[name][slot])
                if rank == 0 and not np.array_equal(theirs.cpu().numpy(), ring.download_last()):
                    ok[0] = 0
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 0:
                sys.exit(f"bench.py: the {name} ring's RCCL gather and torch.distributed's gather disagree")

    def timed_blocks(name):
        """Warm-up, then blocks of exactly --steps steps, each bracketed by barrier + synchronize, the MAX over ranks
        of each block's time; repeated until --min-seconds are covered (the count is agreed on across the ranks)."""
        ring = rings[name]
        run_steps(name, args.warmup)
        fence(name)
        ring.reset_timers()
        ring.reset_clock()
        seconds, planned = [], 1
        while len(seconds) < planned:
            t0 = time.perf_counter()
            run_steps(name, args.steps)
            fence(name)
            dt = time.perf_counter() - t0
            if launched:
                t = torch.tensor([dt], dtype=torch.float64, device=reduce_device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t[0])
            seconds.append(dt)
            if len(seconds) == 1:
                planned = max(1, min(50, math.ceil(args.min_seconds / max(dt, 1e-6))))
        return seconds

    results = {}
    for name in ("pipelined", "blocking"):
        enter(f"timed blocks, {name}", 900.0)
        seconds = timed_blocks(name)
        ring = rings[name]
        timers, cpu = ring.timers(), ring.cpu_times()
        results[name] = {"seconds": seconds, "kernel_ms": timers["kernel_ms"] / max(1, timers["frames"]),
                         "cpu_us": {k: round(cpu[k + "_s"] / max(1, cpu["frames"]) * 1e6, 1) for k in ("submit", "collect", "wait")}}
    # The dominant kernel ALONE, by HIP events right around its launch on the launch stream (plain launches: the events
    # of a replayed graph cannot be timed), one frame at a time -- the duration the roofline is quoted for.
    enter("the dominant kernel alone (plain launches, HIP events)", 600.0)
    ring = rings["blocking"]
    ring.set_graph_mode(False)
    run_steps("blocking", args.warmup)
    fence("blocking")
    ring.reset_timers()
    run_steps("blocking", max(args.steps, 20))
    fence("blocking")
    timers = ring.timers()
    alone_ao_ms = timers["ao_ms"] / max(1, timers["ao_frames"])
    alone_kernel_ms = timers["kernel_ms"] / max(1, timers["frames"])
    ring.set_graph_mode(not args.plain_launches)

    # whole-job numbers: max time over ranks (above), sum of rays over ranks
    st = rings["pipelined"].host(0).stats()
    my_rays = st["primary_rays"] + st["ao_rays"]
    if launched:
        t = torch.tensor([alone_ao_ms, alone_kernel_ms, results["pipelined"]["kernel_ms"]], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        alone_ao_ms, alone_kernel_ms, results["pipelined"]["kernel_ms"] = (float(x) for x in t)
        r = torch.tensor([my_rays, st["primary_hits"], st["ao_occluded"]], dtype=torch.int64, device=reduce_device)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays, total_hits, total_occluded = (int(x) for x in r)
    else:
        total_rays, total_hits, total_occluded = my_rays, st["primary_hits"], st["ao_occluded"]
    enter("statistics, images, the line", 900.0)
    images = {name: final_image(name) for name in rings}
    rccl_described = {"failed": rccl_failure, "world_size": world} if rccl_failure else None
    if rccl:
        comm_ranks, version = rings["pipelined"].rccl_info()
        rccl_described = {"comm_ranks": comm_ranks, "version_code": version, "world_size": world}
        if comm_ranks not in (-1, world):
            sys.exit(f"bench.py: the RCCL communicator has {comm_ranks} ranks, the job {world}")
    scene_bytes, scene_copies, _ = rings["pipelined"].device_bytes()
    calibration = rings["pipelined"].calibration()
    # how far the walk intervals found at upload confine the any-hit packets (rt_walk_entries; this rank's bands)
    intervals = rings["pipelined"].host(0).walk_entries()

    if rank == 0:
        md5 = {name: hashlib.md5(rt.pgm_bytes(img)).hexdigest() for name, img in images.items()}
        if golden_md5:
            for name in md5:
                if md5[name] != golden_md5:
                    sys.exit(f"bench.py: PGM md5 {md5[name]} ({name}) != golden {golden_md5} -- refusing to report a number")
            if total_hits != counters["primary_hits"] or total_occluded != counters["ao_occluded"]:
                sys.exit(f"bench.py: ray statistics differ from the reference traversal: {total_hits} hit sub-pixels, "
                         f"{total_occluded} occluded rays against {counters['primary_hits']}, {counters['ao_occluded']}")
        elif md5["pipelined"] != md5["blocking"]:
            sys.exit("bench.py: the pipelined and the blocking frame differ")

        def per_step_ms(name):
            return [s / args.steps * 1e3 for s in results[name]["seconds"]]

        # `value` is SURVEY 8d's metric: rays over the time from launch to completion of a frame, one frame at a time -- a
        # ring of ONE host, i.e. the reference's blocking OpenCLHost::operator()() (src/opencl_host.cc:137-149).  What the
        # device sustains with several frames in flight is a machine-throughput figure and goes under its own key.
        pipe_ms, block_ms = per_step_ms("pipelined"), per_step_ms("blocking")
        ms_per_step = statistics.median(block_ms)
        pipe_ms_per_step = statistics.median(pipe_ms)
        value = total_rays / (ms_per_step * 1e-3) / 1e6
        gather = ("the ring's RCCL gather to rank 0" if rccl else
                  f"torch.distributed ({backend}) gather to rank 0 on the device -- the library's RCCL gather could not be set up: {rccl_failure}"
                  if rccl_failure else f"{backend} gather to rank 0 (rehearsal, staged through the host)" if launched else None)
        out = {
            # BASELINE.json's metric, verbatim, for the 1920x1080 workloads it is quoted on
            "metric": "Mrays/s at 1920×1080 (bunny.off, sibenik.off); PGM bit-exact vs CPU" if "1080p" in args.workload
                      else "Mrays/s; PGM bit-exact vs CPU",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32",
            "data": ("bunny.off (the reference's mesh asset)" if w["mesh"] == "bunny" else
                     "synthetic height field generated in memory (tools/big_meshes.py)" if w["mesh"].startswith("terrain:") else
                     "synthetic interior scene, stand-in for the missing sibenik.off (tools/make_interior_mesh.py)")
                    + ", fixed camera, no randomness in this path",
            "config": {"workload": w["label"], "rays_per_frame": total_rays, "primary_hits": total_hits,
                       "parallelism": f"image bands x{world}" + (f", {gather}" if gather else ""),
                       "frames_in_flight": 1, "n_hosts_per_gpu": 1,
                       # what the exchange step's communicator says about itself (ncclCommCount, ncclGetVersion): a scaling
                       # record can check that RCCL really saw `n_gpus` ranks; null without RCCL (a single unlaunched process)
                       "rccl": rccl_described,
                       "ipc": {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")},
                       "frame_launch": "plain launches" if args.plain_launches else "hipGraph replay",
                       "pgm_md5": md5["blocking"], "pgm_matches_golden": golden_md5 is not None,
                       "scene_load_s": round(t_load, 3), "scene_build_s": round(t_scene, 3),
                       # one copy of the scene per GPU whatever the number of hosts; which form of the AO pass the ring's
                       # calibration at upload chose for this scene (ms per ao_kernel without / with look-ahead loads)
                       "scene_bytes_on_device": scene_bytes, "scene_copies_on_device": scene_copies,
                       "triangles": scene.num_faces,
                       "ao_pass_calibration": {"ms_without_lookahead": round(calibration[0], 4), "ms_with_lookahead": round(calibration[1], 4),
                                               "lookahead_in_use": calibration[2]},
                       "walk_intervals": {"tiles_hit": intervals["tiles_hit"],
                                          "share_of_node_records_per_tile": round(intervals["mean_share"], 4),
                                          "share_of_node_records_per_packet": round(intervals["mean_packet_share"], 4)},
                       "device": torch.cuda.get_device_name(device)},
            # every block is exactly `steps` frames, one at a time, between barrier + synchronize; `value` / `ms_per_step` are the median block
            "blocks": dict(summary(block_ms), unit="ms per step", seconds_covered=round(sum(results["blocking"]["seconds"]), 3),
                           mrays_per_s_min=round(total_rays / (max(block_ms) * 1e-3) / 1e6, 1),
                           mrays_per_s_max=round(total_rays / (min(block_ms) * 1e-3) / 1e6, 1)),
            "cpu_us_per_step": results["blocking"]["cpu_us"],
            "kernels_ms_per_frame": round(alone_kernel_ms, 4),
            # NOT the metric: the same frames as a steady stream through a ring of `frames_in_flight` hosts (the next
            # frames' passes fill what a finishing pass frees) -- what the device sustains, measured the same way
            "pipelined": dict(value=round(total_rays / (pipe_ms_per_step * 1e-3) / 1e6, 2), unit="Mrays/s",
                              ms_per_frame=summary(pipe_ms), frames_in_flight=in_flight, n_hosts_per_gpu=in_flight,
                              pgm_md5=md5["pipelined"], cpu_us_per_step=results["pipelined"]["cpu_us"],
                              seconds_covered=round(sum(results["pipelined"]["seconds"]), 3)),
        }
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, _ = cpu_baseline(opt, scene, images["blocking"], w)
        has_ao = bool(opt.enable_ao)
        dominant = "ao_kernel" if has_ao else "primary_kernel"
        dominant_ms = alone_ao_ms if has_ao else alone_kernel_ms
        seconds = dominant_ms * 1e-3
        pmc = pmc_for(args.workload) if world == 1 else None  # counters are per launch of the WHOLE frame on one GPU
        # The contractual HBM line: algorithmic bytes of the reference traversal over the measured
        # launch time, beside the HBM bytes the counters really saw.  The scene is cache-resident, so this is NOT
        # the bound the kernel is under; `x_peak` may exceed 1 and is therefore not called a fraction.
        hbm = {"peak": HBM_PEAK_GBS, "unit": "GB/s"}
        if counters is not None:
            sub = opt.total_width * opt.total_height
            parts = algorithmic_bytes(counters, sub)
            bytes_per_launch = (parts["ao"] if has_ao else parts["frame"]) / world  # bands are interleaved over ranks
            hbm.update(algorithmic_bytes_per_launch=int(bytes_per_launch),
                       algorithmic_GBps=round(bytes_per_launch / seconds / 1e9, 1),
                       algorithmic_x_peak=round(bytes_per_launch / seconds / 1e9 / HBM_PEAK_GBS, 3),
                       frame_algorithmic_GBps=round(parts["frame"] / world / (alone_kernel_ms * 1e-3) / 1e9, 1))
        traffic = None
        if pmc is not None:
            traffic = int(pmc["hbm_bytes"])
            hbm.update(measured_bytes_per_launch=traffic, measured_GBps=round(traffic / seconds / 1e9, 1),
                       measured_frac=round(traffic / seconds / 1e9 / HBM_PEAK_GBS, 4))
        hbm["note"] = ("algorithmic bytes = REFERENCE traversal (36 B/node visit + 60 B/triangle test [+ 48 B/hit + 4 B/"
                       "sub-pixel for the primary pass]); they are served by the scalar cache and L2, not by HBM")
        # The binding roofline: vector-instruction issue.  ONE scope for achieved / frac: the dominant kernel's
        # instructions per launch (PMC pass of the same kernel sources, the grid a host alone launches; null if the
        # sources changed since) over its launch duration with one frame at a time, HIP events, measured live.
        roof = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_PER_CLK_SIMD,
                "unit": "wave64 VALU instr/clk/SIMD (1024 SIMDs, 2.4 GHz)", "frac": None, "traffic": traffic,
                "scope": "dominant kernel alone: its instructions per launch over its launch duration, one frame at a time",
                "kernel": dominant, "kernel_ms": round(dominant_ms, 4), "frame_kernels_ms": round(alone_kernel_ms, 4)}
        if pmc is not None:
            rate = pmc["valu_insts"] / (seconds * CLOCK_HZ * SIMDS)
            roof.update(achieved=round(rate, 4), frac=round(rate / VALU_PEAK_PER_CLK_SIMD, 4),
                        insts_per_launch=int(pmc["valu_insts"]), ceiling_measured=pmc.get("valu_ceiling_measured"),
                        levels=pmc.get("levels"), counters_from=pmc.get("source"))
            if pmc.get("valu_ceiling_measured"):
                roof["frac_of_measured_ceiling"] = round(rate / pmc["valu_ceiling_measured"], 4)
            # the hardware's own view, from the counter pass: cycles the SIMDs spent issuing vector instructions
            # (SQ_ACTIVE_INST_VALU, quad-cycles summed over the SIMDs) over the cycles the launch took
            # (GRBM_GUI_ACTIVE, summed over the 8 XCDs).  The counter books 4 cycles per instruction; v_fma / v_mul /
            # v_add on registers issue in 2.3 (tools/microbench/valu_rate_probe.hip), so a saturated SIMD can read > 1.
            if pmc.get("active_inst_valu_quad_cycles") and pmc.get("gui_active_cycles"):
                roof["valu_nominal_issue_share"] = round(pmc["active_inst_valu_quad_cycles"] * 4.0 / SIMDS /
                                                   (pmc["gui_active_cycles"] / 8.0), 4)
            # The other scope, under its own key: what the device does per unit of time with frames in flight -- the
            # vector instructions of a whole frame, every kernel, counted for the grid a host launches when it shares
            # its GPU, over the measured time per frame of the pipelined blocks.
            frame_insts = pmc.get("shared_frame_valu_insts") or pmc.get("frame_valu_insts")
            if frame_insts:
                frame_rate = frame_insts / (pipe_ms_per_step * 1e-3 * CLOCK_HZ * SIMDS)
                roof["frame_pipelined"] = {
                    "valu_insts_per_frame": int(frame_insts), "grid": "shared" if pmc.get("shared_frame_valu_insts") else "alone",
                    "ms_per_frame": round(pipe_ms_per_step, 4), "achieved": round(frame_rate, 4),
                    "frac": round(frame_rate / VALU_PEAK_PER_CLK_SIMD, 4),
                    "frac_of_measured_ceiling": round(frame_rate / pmc["valu_ceiling_measured"], 4) if pmc.get("valu_ceiling_measured") else None,
                    "frames_in_flight": in_flight}
        else:
            roof["counters_from"] = None if world > 1 else "profiles/pmc.json is missing or was collected for other kernel sources"
        roof["hbm"] = hbm
        out["roofline"] = roof
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if golden_md5 is None and cpu is None:
            # a generated scene without a committed golden, and the CPU check of this run switched off: nothing in THIS run
            # has compared the frame with the CPU's (tests/test_big_scenes.py does, at small sizes) -- the metric says so
            out["metric"] = out["metric"].replace("PGM bit-exact vs CPU", "PGM not compared with the CPU in this run")
        elif golden_md5 is None:
            out["config"]["pgm_checked_against"] = "the oracle's rows of this run's cpu_baseline sample, byte for byte"
        if e2e is not None:
            out["end_to_end"] = e2e
        print(json.dumps(out), flush=True)

    enter("closing", 120.0)
    for ring in rings.values():
        ring.close()
    if launched:
        dist.destroy_process_group()
    stage["done"] = True


if __name__ == "__main__":
    main()
