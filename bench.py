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

Rank 0 prints ONE JSON line (contract in the task description) carrying
`roofline` -- the bound the dominant kernel is actually under (vector-instruction
issue; the 12 MB scene is cache-resident), with the contractual HBM line of
SURVEY.md 8d (algorithmic bytes of the REFERENCE traversal over the HIP-event
kernel time, and the HBM bytes the PMC counters really saw) under `roofline.hbm`
-- and, at N = 1, `cpu_baseline` (the reference's own kernel compiled for x86-64
when oracle/_ref/ holds it, else this repo's C restatement, on the host cores).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
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
    "interior_1080p_ao": dict(
        mesh="interior", bvh="longest", width=1920, height=1080, ss=1, ao=3, golden="interior_1080p_s1_a3",
        label="interior stand-in for the missing sibenik.off, 1920x1080 -s 1 -a 3"),
}
DEFAULT_WORKLOAD = "bunny_1080p_ao"


def workload_options(rt, w):
    return rt.Options.defaults(width=w["width"], height=w["height"], n_super_samples=w["ss"], ao_num_samples=w["ao"],
                               enable_ao=int(w["ao"] != 0), bvh_method=0 if w["bvh"] == "longest" else 1)


def mesh_path(name: str) -> str:
    from tools.meshes import bunny_path, interior_path

    return bunny_path() if name == "bunny" else interior_path()


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
    for name in ("kernels.hip", "device_types.h", "scene_pack.cc", "walk_tree.cc"):
        with open(os.path.join(ROOT, "opencl_raytracer_amd", "csrc", name), "rb") as f:
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
        "value": round(rays / seconds / 1e6, 3), "unit": "Mrays/s", "cores": int(used), "kind": kind,
        "sample": f"rows {rows[0]}..{rows[1]} of {p.height} ({rays} rays, {seconds:.2f} s wall)"
                  + ("; PGM byte-identical to the GPU frame" if rows == (0, p.height) else "; these PGM rows byte-identical to the GPU frame's"),
        "port_value": round(rays / t_port / 1e6, 3),
    }, full_rays


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # (a quarter of a second of frames: the pipeline's fill and drain weigh 1-2 % at 50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6],
                    help="renderers per GPU taking the frames in turn (1: one frame at a time; 0 = by the number of "
                         "ranks: 3 up to two GPUs, 4 at four, 6 at eight)")
    args = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ  # under torch.distributed.run
    if not launched and args.gpus > 1:
        # A bare `python bench.py --gpus N`: start the N ranks as FRESH processes.  This process has not imported
        # torch or touched the GPU yet (and must not: a process that initialised HIP may not exec or fork ranks).
        import socket
        import subprocess

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
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    w = WORKLOADS[args.workload]
    opt = workload_options(rt, w)
    t0 = time.perf_counter()
    scene = rt.Scene.load_off(mesh_path(w["mesh"])).build_bvh(opt.bvh_method)
    t_scene = time.perf_counter() - t0

    # Several renderers of the same scene on this GPU, each on its own stream, take the frames in turn: while frame i's
    # ambient-occlusion pass runs out (its last quarter runs at falling occupancy: the queues are drained, the
    # workgroups end one by one) the next frames' passes fill the wave slots it frees, and the latency-bound primary
    # pass runs beside a vector-issue-bound one.  Headline workload: 1.56 ms per frame with one renderer, 1.35 with
    # two, 1.27 with three (default), 1.27 with four.  One frame at a time: --in-flight 1.
    # The smaller a rank's share of the frame, the more of it is start and end of passes: one GPU's eighth of the
    # headline frame takes 0.48 ms with one renderer, 0.31 with three, 0.29 with four, 0.25 with six; the whole frame
    # 1.53 / 1.23 / 1.20 / 1.25 (tools/ring_sweep.py).  A step costs the CPU ~0.1 ms (tools/step_overhead_probe.py).
    in_flight = args.in_flight if args.in_flight > 0 else 3 if world <= 2 else 4 if world <= 4 else 6
    hosts = [rt.Host(opt, device_index, rank, world) for _ in range(in_flight)]
    for h in hosts:
        h.upload_scene(scene)
        h.set_device_share(len(hosts))
    # (each host's own stream, created by the library: streams handed out by torch's pool ended up on ONE hardware
    # queue here -- rocprofv3's kernel trace showed every kernel of both renderers in the same queue, one after the other)
    render_streams = [torch.cuda.ExternalStream(h.stream_handle, device=device) for h in hosts]
    host = hosts[0]
    # The gather and the assembly of the final image run on a third stream (torch.distributed syncs with the current one).
    stream = torch.cuda.Stream(device)
    torch.cuda.set_stream(stream)

    # band buffers: equal-sized on every rank so the gather is one collective
    from opencl_raytracer_amd.multi_gpu import BandGatherer, BandLayout

    layout = BandLayout(opt, world)
    assert layout.local_rows(rank) == host.local_rows
    # One band buffer and one gatherer per renderer: frame i's bands are gathered (RCCL's own stream) and its rows moved
    # into place on rank 0 while frame i + 1 is rendered -- a frame is finished right after the next one has been
    # enqueued, and the last one before the closing fence, so K timed steps are K complete frames.
    slots = len(hosts) if len(hosts) > 1 else 2
    bands = [torch.zeros((layout.max_rows, opt.width), dtype=torch.uint8, device=device) for _ in range(slots)]
    staged = launched and backend != "nccl"  # rehearsal path: the gather is staged through host memory
    gatherers = [BandGatherer(layout, rank, "cpu" if staged else device) for _ in range(slots)]
    released = [None] * slots  # event: the slot's band and gatherer were last read (its frame was assembled)
    result = {"frames": 0}
    open_frames = []  # slots of the frames enqueued and not yet finished, oldest first

    def finish_oldest():
        k = open_frames.pop(0)
        h, rs = hosts[k % len(hosts)], render_streams[k % len(hosts)]
        if len(hosts) > 1:
            # The CPU waits for the frame (the next ones are already queued on the other renderers' streams), and only
            # then issues its gather: a stream-level wait for a frame that has just begun would sit in a hardware
            # queue as a barrier packet, and HIP streams share hardware queues -- with the collective's stream in
            # play such a barrier ended up ahead of another renderer's kernels and the frames ran one after the other.
            h.sync()
        else:
            stream.wait_stream(rs)
        gatherers[k].start(bands[k].cpu() if staged else bands[k])
        result["final"] = gatherers[k].finish()
        released[k] = torch.cuda.Event()
        released[k].record(stream)

    def step():
        k = result["frames"] % slots
        h, rs = hosts[k % len(hosts)], render_streams[k % len(hosts)]
        if released[k] is not None:  # the frame that used this band buffer before has been assembled
            released[k].synchronize() if len(hosts) > 1 else rs.wait_event(released[k])
        h.render_async()
        h.resize_into_device(bands[k].data_ptr())
        open_frames.append(k)
        result["frames"] += 1
        while len(open_frames) > max(1, len(hosts) - 1):  # the oldest frame -- the newer ones are queued behind it
            finish_oldest()

    def fence():
        while open_frames:
            finish_oldest()
        torch.cuda.synchronize(device)
        if launched:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    for h in hosts:
        h.sync()
        h.reset_timers()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    for h in hosts:
        h.sync()  # folds the HIP event pairs into the kernel-time statistics

    # whole-job numbers: max time over ranks, sum of rays over ranks
    st = host.stats()
    my_rays = st["primary_rays"] + st["ao_rays"]
    launches = max(1, sum(h.kernel_launches for h in hosts))
    kernel_ms = sum(h.total_kernel_ms for h in hosts) / launches
    ao_ms = sum(h.total_ao_ms for h in hosts) / launches  # HIP events right around the ao_kernel launch
    if launched:
        t = torch.tensor([elapsed, kernel_ms, ao_ms], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms_max, ao_ms_max = float(t[0]), float(t[1]), float(t[2])
        r = torch.tensor([my_rays, st["primary_hits"], st["ao_occluded"]], dtype=torch.int64,
                         device=device if backend == "nccl" else "cpu")
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        total_rays, total_hits, total_occluded = (int(x) for x in r)
    else:
        kernel_ms_max, ao_ms_max = kernel_ms, ao_ms
        total_rays, total_hits, total_occluded = my_rays, st["primary_hits"], st["ao_occluded"]

    if rank == 0:
        final_u8 = result["final"].cpu().numpy()
        pgm_md5 = hashlib.md5(rt.pgm_bytes(final_u8)).hexdigest()
        golden_md5, counters = None, None
        if w["golden"]:
            with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
                g = json.load(f)["renders"][w["golden"]]
            golden_md5, counters = g["pgm_md5"], g["counters"]
            if pgm_md5 != golden_md5:
                sys.exit(f"bench.py: PGM md5 {pgm_md5} != golden {golden_md5} -- refusing to report a number")
            if total_hits != counters["primary_hits"] or total_occluded != counters["ao_occluded"]:
                sys.exit("bench.py: ray statistics differ from the reference traversal")

        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays / (elapsed / args.steps) / 1e6
        out = {
            # BASELINE.json's metric, verbatim, for the 1920x1080 workloads it is quoted on
            "metric": "Mrays/s at 1920\u00d71080 (bunny.off, sibenik.off); PGM bit-exact vs CPU" if "1080p" in args.workload
                      else "Mrays/s; PGM bit-exact vs CPU",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32",
            "data": ("bunny.off (the reference's mesh asset)" if w["mesh"] == "bunny" else
                     "synthetic interior scene, stand-in for the missing sibenik.off (tools/make_interior_mesh.py)")
                    + ", fixed camera, no randomness in this path",
            "config": {"workload": w["label"], "rays_per_frame": total_rays, "primary_hits": total_hits,
                       "parallelism": f"image bands x{world}" + (f", {'RCCL' if backend == 'nccl' else backend} gather to rank 0" if launched else ""),
                       "frames_in_flight": len(hosts), "pgm_md5": pgm_md5, "pgm_matches_golden": golden_md5 is not None,
                       "scene_build_s": round(t_scene, 3), "device": torch.cuda.get_device_name(device)},
        }
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, _ = cpu_baseline(opt, scene, final_u8, w)
        has_ao = bool(opt.enable_ao)
        dominant = "ao_kernel" if has_ao else "primary_kernel"
        dominant_ms = ao_ms_max if has_ao else kernel_ms_max
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
                       frame_algorithmic_GBps=round(parts["frame"] / world / (kernel_ms_max * 1e-3) / 1e9, 1))
        traffic = None
        if pmc is not None:
            traffic = int(pmc["hbm_bytes"])
            hbm.update(measured_bytes_per_launch=traffic, measured_GBps=round(traffic / seconds / 1e9, 1),
                       measured_frac=round(traffic / seconds / 1e9 / HBM_PEAK_GBS, 4))
        hbm["note"] = ("algorithmic bytes = REFERENCE traversal (36 B/node visit + 60 B/triangle test [+ 48 B/hit + 4 B/"
                       "sub-pixel for the primary pass]); they are served by the scalar cache and L2, not by HBM")
        # The binding roofline: vector-instruction issue.  Instruction count per launch from the PMC pass of the same
        # kernel sources (profiles/pmc.json, null if the sources changed since), launch time measured live.
        roof = {"bound": "valu", "achieved": None, "peak": VALU_PEAK_PER_CLK_SIMD,
                "unit": "wave64 VALU instr/clk/SIMD (1024 SIMDs, 2.4 GHz)", "frac": None, "traffic": traffic,
                "kernel": dominant, "kernel_ms": round(dominant_ms, 4), "frame_kernels_ms": round(kernel_ms_max, 4)}
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
        else:
            roof["counters_from"] = None if world > 1 else "profiles/pmc.json is missing or was collected for other kernel sources"
        # With several frames in flight a launch shares the device with its neighbours' and takes longer than alone
        # (its event pair spans the time it waits for wave slots): the dominant kernel's instructions over ITS launch
        # duration then say little about the kernel.  What the device does per unit of time is the vector
        # instructions of a whole frame -- every kernel, counted by the same PMC pass -- over the measured time per
        # frame; that is what `achieved` / `frac` hold in this case (`scope` says which), the per-launch figures of
        # the timed region stay beside them under `launch`, and `--in-flight 1` gives the kernel alone.
        roof["scope"] = "dominant kernel: its instructions per launch over its launch duration"
        if pmc is not None and pmc.get("frame_valu_insts"):
            frame_rate = pmc["frame_valu_insts"] / (ms_per_step * 1e-3 * CLOCK_HZ * SIMDS)
            frame = {"valu_insts_per_frame": int(pmc["frame_valu_insts"]), "ms_per_frame": round(ms_per_step, 4),
                     "achieved": round(frame_rate, 4), "frac": round(frame_rate / VALU_PEAK_PER_CLK_SIMD, 4),
                     "frac_of_measured_ceiling": round(frame_rate / pmc["valu_ceiling_measured"], 4)
                     if pmc.get("valu_ceiling_measured") else None,
                     "frames_in_flight": len(hosts)}
            roof["frame"] = frame
            if len(hosts) > 1:
                roof["launch"] = {"kernel": dominant, "kernel_ms_sharing_the_device": roof["kernel_ms"],
                                  "achieved": roof["achieved"], "frac": roof["frac"],
                                  "frac_of_measured_ceiling": roof.get("frac_of_measured_ceiling")}
                roof.update(achieved=frame["achieved"], frac=frame["frac"],
                            frac_of_measured_ceiling=frame["frac_of_measured_ceiling"],
                            scope=f"frame: the vector instructions of all its kernels over the time per frame, {len(hosts)} frames in flight")
        roof["hbm"] = hbm
        out["roofline"] = roof
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)

    for h in hosts:
        h.close()
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
