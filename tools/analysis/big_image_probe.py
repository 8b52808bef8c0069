"""Probe (GPU box): a frame far larger than any golden one -- 1920x1080 at 256 samples per pixel (30720 x 17280 sub-pixels,
2.1 GB of float image, ~300 M hit-list slots) -- through the binding: completes, device resize == host resize of the
downloaded floats (bit for bit), ray statistics are 4x those of the 64-samples golden frame to within the sampling, and the
8-bit image is within a grey level of the 64-samples golden PGM on average (more samples of the same picture)."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import opencl_raytracer_amd as rt
from tools.meshes import bunny_path
samples = int(sys.argv[1]) if len(sys.argv) > 1 else 256
opt = rt.Options.defaults(width=1920, height=1080, n_super_samples=samples, ao_num_samples=3)
scene = rt.Scene.load_off(bunny_path()).build_bvh(0)
t0 = time.perf_counter()
host = rt.Host(opt, 0); host.upload_scene(scene)
t1 = time.perf_counter()
host.render(); host.render()
u8 = host.download_u8()
st = host.stats()
print(f"-s {samples}: upload {t1 - t0:.2f} s, kernels {host.last_kernel_ms:.1f} ms, ao {host.last_ao_ms:.1f} ms, "
      f"{(st['primary_rays'] + st['ao_rays']) / host.last_kernel_ms / 1e6:.1f} Grays/s, stats {st}", flush=True)
same = "skipped"
if samples <= 256:  # (the floats of larger frames do not fit a test box's host memory twice over)
    img = host.download()
    same = np.array_equal(u8, rt.resize_cpu(opt, img))
    del img
host.close()
opt64 = rt.Options.defaults(width=1920, height=1080, n_super_samples=64, ao_num_samples=3)
h64 = rt.Host(opt64, 0); h64.upload_scene(scene); h64.render()
u64 = h64.download_u8(); s64 = h64.stats()
golden = json.load(open(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "golden.json")))["renders"]["bunny_1080p_s64_a3"]
print("64-samples frame is the golden one:", hashlib.md5(rt.pgm_bytes(u64)).hexdigest() == golden["pgm_md5"])
diff = np.abs(u8.astype(np.int32) - u64.astype(np.int32))
ratio = st["primary_hits"] / s64["primary_hits"] / (samples / 64.0)
print(f"device resize == host resize: {same}; mean |grey difference| to the 64-samples frame {diff.mean():.3f} (max {diff.max()}), "
      f"hits per sample relative to it {ratio:.5f}, occluded share {st['ao_occluded'] / st['ao_rays']:.5f} vs {s64['ao_occluded'] / s64['ao_rays']:.5f}")
