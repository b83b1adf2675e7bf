"""CPU, world_size 2 over gloo: the frame-sharding helpers bench.py uses for --gpus N."""
import importlib
import os
import socket
import sys
import time

import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    batch = importlib.import_module("stereo-matching_amd.batch")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = batch.shard_pairs(7, rank, world)
    done = []

    def step(i):
        time.sleep(0.02 * (rank + 1))      # rank 1 is twice as slow
        done.append(i)

    elapsed = batch.timed_steps(step, steps=5, warmup=1, sync_fn=lambda: None, dist=dist)
    q.put((rank, mine, len(done), elapsed))
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, n0, e0), (r1, s1, n1, e1) = res
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5]           # disjoint, complete
    assert n0 == n1 == 6                                     # warmup + steps on every rank
    assert abs(e0 - e1) < 1e-9                               # both see the max over ranks
    assert e0 >= 5 * 0.04 - 1e-3                             # ... which is the slower rank's time
    batch = importlib.import_module("stereo-matching_amd.batch")
    assert abs(batch.throughput(2, 5, 1920, 1080, 256, 1.0) - 2 * 5 * 530.8416) < 1e-6


def test_single_process_path():
    batch = importlib.import_module("stereo-matching_amd.batch")
    assert batch.shard_pairs(5, 0, 1) == [0, 1, 2, 3, 4]
    assert batch.shard_pairs(2, 3, 4) == []
    calls = []
    e = batch.timed_steps(lambda i: calls.append(i), steps=3, warmup=2, sync_fn=lambda: None)
    assert calls == [0, 1, 0, 1, 2] and e >= 0
