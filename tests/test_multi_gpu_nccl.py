"""The RCCL leg of the multi-GPU path on real hardware.

The reference is single-device (src/opencl_host.cc:16-32); splitting a frame over ranks
and gathering the bands is new code, rehearsed over gloo on the CPU
(tests/test_distributed_cpu.py).  Here bench.py itself runs as rank(s) under
torch.distributed.run with the "nccl" backend (= RCCL): communicator creation, the band
gather into rank 0's stacked buffer, the barrier and the max/sum all-reduces of the timing
contract all execute on the GPU box -- with one rank on a one-GPU box, with two where the
box has two GPUs.  The assembled PGM must be the golden one."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _run_bench(nproc, extra_env=None, bare=False, extra_args=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra_env or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    args = ["--gpus", str(nproc), "--steps", "3", "--warmup", "1", "--workload", "bunny_600_defaults", "--no-cpu-baseline",
            "--no-end-to-end", "--min-seconds", "0.05"] + list(extra_args)
    if bare:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_rccl_world_of_one_runs_the_gather(golden):
    out = _run_bench(1)
    assert out["n_gpus"] == 1 and "RCCL gather" in out["config"]["parallelism"]
    assert out["pipelined"]["pgm_md5"] == golden["renders"]["bunny_600_defaults"]["pgm_md5"] and out["pipelined"]["value"] > 0
    assert out["config"]["pgm_md5"] == golden["renders"]["bunny_600_defaults"]["pgm_md5"]
    assert out["config"]["pgm_matches_golden"] is True and out["value"] > 0


def test_the_job_survives_a_gather_that_cannot_be_set_up(golden):
    """Should the library's own RCCL gather fail to come up (here: the rehearsal knob refuses it on every rank), the ranks
    agree on it, make their rings afresh without a gather, and torch.distributed gathers the same band buffers on the
    device: the line still comes, with the golden image, and says which exchange step it was measured with."""
    out = _run_bench(1, extra_env={"OCRT_BENCH_FAIL_RCCL": "1"})
    assert out["n_gpus"] == 1 and "could not be set up" in out["config"]["parallelism"]
    assert out["config"]["rccl"]["failed"] and "OCRT_BENCH_FAIL_RCCL" in out["config"]["rccl"]["failed"]
    assert out["config"]["pgm_md5"] == golden["renders"]["bunny_600_defaults"]["pgm_md5"]
    assert out["pipelined"]["pgm_md5"] == golden["renders"]["bunny_600_defaults"]["pgm_md5"] and out["value"] > 0


def test_rccl_two_ranks_assemble_the_golden_frame(golden):
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per GPU under RCCL)")
    out = _run_bench(2)
    assert out["n_gpus"] == 2 and out["config"]["pgm_md5"] == golden["renders"]["bunny_600_defaults"]["pgm_md5"]


def test_bare_gpus_flag_starts_its_own_ranks(golden):
    """`python bench.py --gpus N` from a bare shell (the driver's command shape) spawns the ranks itself."""
    import torch

    n = 2 if torch.cuda.device_count() >= 2 else 1
    if n == 1:
        # one GPU: rehearse the spawn path with two gloo ranks sharing the device (host-staged gather)
        out = _run_bench(2, extra_env={"OCRT_BENCH_BACKEND": "gloo"}, bare=True)
        assert out["n_gpus"] == 2
    else:
        out = _run_bench(2, bare=True)
        assert out["n_gpus"] == 2 and "RCCL" in out["config"]["parallelism"]
    assert out["config"]["pgm_md5"] == golden["renders"]["bunny_600_defaults"]["pgm_md5"]


def test_four_gloo_ranks_rehearse_the_eight_gpu_run(golden):
    """The shape of the driver's 8-GPU run on the one GPU of the box, as far as its process limit allows (six processes
    may hold the GPU: this test runner, the torchrun agent and four ranks; gloo, because RCCL refuses two ranks on one
    device): each rank with the eight render hosts per GPU that bench.py uses at eight ranks and a blocking ring beside
    them, bands of 4 rows dealt round-robin, the gather, barriers and all-reduces of the timing contract.  The assembled
    PGM is the golden one in both modes."""
    out = _run_bench(4, extra_env={"OCRT_BENCH_BACKEND": "gloo"}, bare=True, extra_args=["--in-flight", "8"])
    assert out["n_gpus"] == 4 and out["config"]["frames_in_flight"] == 1 and out["pipelined"]["frames_in_flight"] == 8
    assert out["config"]["pgm_md5"] == golden["renders"]["bunny_600_defaults"]["pgm_md5"]
    assert out["pipelined"]["pgm_md5"] == golden["renders"]["bunny_600_defaults"]["pgm_md5"]
    assert out["config"]["rays_per_frame"] == 1440000 + 28 * golden["renders"]["bunny_600_defaults"]["counters"]["primary_hits"]
