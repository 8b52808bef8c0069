"""Probe (GPU box): what one bench.py step costs on the CPU side -- the same calls (render_async, resize_into_device,
host sync, RCCL gather start / finish in a world of one) on a frame so small (64 x 64, no AO) that the GPU is idle.

    RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29540 python3 tools/step_overhead_probe.py [hosts]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import opencl_raytracer_amd as rt  # noqa: E402
from bench import mesh_path  # noqa: E402
from opencl_raytracer_amd.multi_gpu import BandGatherer, BandLayout  # noqa: E402

n_hosts = int(sys.argv[1]) if len(sys.argv) > 1 else 3
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=device)
opt = rt.Options.defaults(width=64, height=64, n_super_samples=1, ao_num_samples=0, enable_ao=0)
scene = rt.Scene.load_off(mesh_path("bunny")).build_bvh(0)
hosts = [rt.Host(opt, 0) for _ in range(n_hosts)]
for h in hosts:
    h.upload_scene(scene)
stream = torch.cuda.Stream(device)
torch.cuda.set_stream(stream)
layout = BandLayout(opt, 1)
bands = [torch.zeros((layout.max_rows, opt.width), dtype=torch.uint8, device=device) for _ in hosts]
gatherers = [BandGatherer(layout, 0, device) for _ in hosts]
open_frames = []
for steps in (50, 1000):
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % n_hosts
        hosts[k].render_async()
        hosts[k].resize_into_device(bands[k].data_ptr())
        open_frames.append(k)
        while len(open_frames) > max(1, n_hosts - 1):
            j = open_frames.pop(0)
            hosts[j].sync()
            gatherers[j].start(bands[j])
            gatherers[j].finish()
    while open_frames:
        j = open_frames.pop(0)
        hosts[j].sync()
        gatherers[j].start(bands[j])
        gatherers[j].finish()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
print(f"{n_hosts} hosts: {dt:.4f} ms per step with an idle GPU (CPU side of a step, RCCL gather of one rank included)")
dist.destroy_process_group()
