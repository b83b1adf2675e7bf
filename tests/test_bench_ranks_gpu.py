"""bench.py's multi-rank path on a one-GPU box: two ranks started by the script itself (a fresh child process tree,
never a re-exec), gloo for the barrier and the max-over-ranks, both ranks on device 0. The RCCL ('nccl') branch needs
one GPU per rank: it runs here with a single rank under the launcher (process group on the device, barrier,
max-over-ranks on a device tensor -- everything but a second participant)."""
import json
import math
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_rehearsal():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR")}
    env["MASTER_PORT"] = str(port)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-c4", "--batch-pairs", "6", "--width", "320", "--height", "200", "--disp", "64"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 prints, rank 1 does not
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["pairs_per_step"] == 2 and j["steps"] == 2 and j["warmup"] == 1
    assert math.isfinite(j["value"]) and j["value"] > 0
    assert abs(j["value"] - 2 * 320 * 200 * 64 * 1e-6 / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    b = j["batch_256"]                                  # the host-fed leg runs on every rank
    assert b["n_gpus"] == 2 and b["pairs_per_gpu"] == 6 and b["pairs_per_s"] > 0
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dist_backend"] == "gloo"


def test_bench_one_rank_under_launcher_rccl():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-c4", "--batch-pairs", "4", "--width", "320", "--height", "200", "--disp", "64"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["dist_backend"] == "nccl" and j["steps"] == 2
    assert math.isfinite(j["value"]) and j["value"] > 0
    assert j["batch_256"]["pairs_per_gpu"] == 4 and j["batch_256"]["pairs_per_s"] > 0
