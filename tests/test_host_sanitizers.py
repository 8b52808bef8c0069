"""CPU: the product's host code under ASan + UBSan -- the scene build on its own, and the WHOLE `render` CLI with the device
stubbed out (the reference's Debug build is an ASan/UBSan build of its whole CLI, CMakeLists.txt:34-40; GPU sanitizers
are not available on the pool, so the sanitizers run on the CPU build only)."""
import hashlib
import os
import subprocess

from conftest import ROOT, mesh_file

CSRC = os.path.join(ROOT, "opencl_raytracer_amd", "csrc")


def test_host_scene_build_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    srcs = [os.path.join(ROOT, "tests", "host_sanitize.cc")] + [os.path.join(CSRC, f) for f in
                                                               ("mesh.cc", "bvh.cc", "ray_tracer.cc", "scene_pack.cc", "walk_tree.cc")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off",
                    "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-pthread",
                    "-DOCRT_DEBUG_KNOBS",  # (the program compares the rebuilt walk tree with the kept one: OCRT_KEEP_TREE)
                    "-I", CSRC, "-o", exe] + srcs,
                   check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe] + [mesh_file(m) for m in ("single", "ties", "blob", "bunny")], capture_output=True,
                       text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok (") == 4


SANITIZE = ["-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fsanitize=address,undefined",
            "-fno-sanitize-recover=undefined", "-pthread"]
SAN_ENV = dict(ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def test_whole_cli_under_asan_ubsan(tmp_path):
    """csrc/render.cc with every line of host code it runs around the device -- option parsing, OFF loader, normals, both BVH
    strategies, face sort, scene packing + validation, walk-tree rebuild, RayTracer::resize, PGM writer, the error exits --
    and tests/hip_host_stub.cc in place of the device (a deterministic pattern instead of ray casting)."""
    exe = str(tmp_path / "render_sanitized")
    srcs = [os.path.join(CSRC, f) for f in ("render.cc", "mesh.cc", "bvh.cc", "ray_tracer.cc", "scene_pack.cc", "walk_tree.cc",
                                            "cli_support.cc")] + [os.path.join(ROOT, "tests", "hip_host_stub.cc")]
    subprocess.run(["g++"] + SANITIZE + ["-I", CSRC, "-o", exe] + srcs, check=True)
    env = dict(os.environ, **SAN_ENV)

    def run(*args, expect=0, **more):
        r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env=dict(env, **more))
        assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-3000:]
        if expect is None:  # (an exception nobody catches, as in the reference's main(): what() on stderr, abort)
            assert r.returncode != 0, (args, r.stdout[-1500:], r.stderr[-1500:])
        else:
            assert r.returncode == expect, (args, r.returncode, r.stdout[-1500:], r.stderr[-1500:])
        return r

    blob, ties = mesh_file("blob"), mesh_file("ties")
    out = tmp_path / "out.pgm"
    # the reference's flags in their three spellings (--opt=value, --opt value, -o value), -h = height
    r = run("-w", 37, "-h", 23, "--supersamples=4", "--ambient-occlusion-samples", "2", "-d", "0.3", "-m", "uniform", "-f", "1.2",
            "-r", "sah", blob, out)
    for phase in ("Building BVH", "Loading OpenCL kernel", "Rendering image"):
        assert phase in r.stdout
    data = out.read_bytes()
    assert data.startswith(b"P5 37 23 255\n") and len(data) == len(b"P5 37 23 255\n") + 37 * 23
    first = hashlib.md5(data).hexdigest()
    # the reference's own flow (float image to the host, RayTracer::resize there) writes the same bytes
    run("-w", 37, "-h", 23, "-s", 4, "-a", 2, "-d", 0.3, "-r", "sah", "--host-resize", 1, blob, out)
    assert hashlib.md5(out.read_bytes()).hexdigest() == first
    # a stream of frames (HipHostRing), several devices (HipHostGroup), the timings line, defaults, a 1 x 1 image
    run("-w", 40, "-h", 30, "--frames", 3, "--in-flight", 2, "--timings", 1, ties, out)
    run("-w", 16, "-h", 16, "-a", 0, "--gpus", 2, ties, out)
    run("-w", 1, "-h", 1, "-s", 1, "--warm-up", 0, ties, out)
    run(blob, out)
    assert out.read_bytes().startswith(b"P5 600 600 255\n")
    # errors: the usage exits, loader errors, an output that cannot be opened, no device
    run(expect=1)
    run(blob, expect=1)
    run("--help", blob, out, expect=0)
    run("-m", "fancy", blob, out, expect=1)
    run("-r", "median", blob, out, expect=1)
    run("--frames", 0, blob, out, expect=1)
    assert "Cannot read file" in run(tmp_path / "missing.off", out, expect=None).stderr
    bad = tmp_path / "bad.off"
    for text in ("", "OFF\n", "OFF\n3 1 0\n0 0 0\n1 0 0\n", "OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n3 0 1 7\n", "PLY\n",
                 "OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n4 0 1 2 0\n", "OFF\n-3 1 0\n", "OFF\n3 1 0\n0 0 x\n"):
        bad.write_text(text)
        r = subprocess.run([exe, str(bad), str(out)], capture_output=True, text=True, env=env)
        assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, (text, r.stderr[-2000:])
        assert r.returncode != 0, text
    run(blob, tmp_path / "no_such_dir" / "out.pgm", expect=1)
    r = subprocess.run([exe, blob, str(out)], capture_output=True, text=True, env=dict(env, OCRT_STUB_NO_DEVICE="1"))
    assert r.returncode != 0 and "No device found" in r.stderr and "ERROR: AddressSanitizer" not in r.stderr
