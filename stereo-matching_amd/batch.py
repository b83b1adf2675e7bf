"""Frame-sharded multi-GPU runs: one process per GPU, independent stereo pairs, no data-path collective.

Frames are the only axis the matcher shards along (every stage couples all pixels and all disparities
of a frame), so N GPUs are N replicas. torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used
for exactly two things: a barrier around the timed region and a max-over-ranks of the elapsed time.
"""
import time


def shard_pairs(n_pairs, rank, world):
    """Indices of the pairs rank `rank` matches: i with i % world == rank (the reference loops its dataset
    list on one GPU, main.cpp:62-255; this deals that list round-robin over the replicas)."""
    return list(range(rank, n_pairs, world))


def timed_steps(step_fn, steps, warmup, sync_fn, dist=None, device=None):
    """Run `warmup` untimed + `steps` timed calls of step_fn(i) between barrier+sync brackets.

    Returns the elapsed seconds of the timed region, maximised over ranks when torch.distributed
    (`dist`) is initialised."""
    use_dist = dist is not None and dist.is_available() and dist.is_initialized()
    for i in range(warmup):
        step_fn(i)
    sync_fn()
    if use_dist:
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for i in range(steps):
        step_fn(i)
    sync_fn()
    if use_dist:
        dist.barrier()
    sync_fn()
    elapsed = time.perf_counter() - t0
    if use_dist:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def throughput(world, steps, width, height, disparities, elapsed_s):
    """Whole-job Mpix*disp/s: every rank matched `steps` pairs of width x height x disparities."""
    return world * steps * (width * height * disparities * 1e-6) / elapsed_s
