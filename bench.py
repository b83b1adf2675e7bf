#!/usr/bin/env python3
"""Headline benchmark: AD-Census Match() throughput on synthetic 1920x1080 pairs, 256 disparities.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) the script starts the N ranks itself
through torch.distributed.run before anything touches a GPU, and exits with their status. A launcher
environment whose WORLD_SIZE differs from --gpus is an error.

A step is one Match of one stereo pair whose images are already resident in HBM (the C-ABI's
adc_match_device). With N > 1 every rank matches its own pairs on its own GPU -- frames are
independent, there is no collective in the data path; torch.distributed is only used for the
barrier and the max-over-ranks of the timed region. Rank 0 prints one JSON line.

Extra objects on that line:
  roofline         -- the dominant kernel (one launch of the cross-aggregation kernel). Algorithmic bytes per
                      launch follow the stage-boundary model: an aggregation ITERATION reads and writes the
                      volume once (8 B per pixel*disparity); this build runs an iteration as two launches, so
                      one launch carries 4 B per pixel*disparity. The per-launch duration is the mean over the
                      eight launches of a match, each timed on its own with HIP events on the launch stream.
                      `per_pass` repeats the figure with the 8 B a single pass actually has to move.
  roofline_kernels -- the same accounting for every volume kernel of a match, launch by launch.
  t_match          -- the host-pointer boundary (adc_match) on pinned host images: wall time per frame with the
                      upload of the two images and the download of the map inside, and those two separately.
  cpu_baseline     -- the CPU oracle (a port of the reference path; the reference has no CPU-only path of its
                      own) timed on a bounded sample of the same workload on all host cores, and
                      cpu_baseline_1thread on one core; rank 0, N = 1 only.
"""
import argparse
import hashlib
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def kernel_source_stamp():
    """sha256 over the kernel sources: profiles/traffic.json is only valid for the sources it was measured on."""
    d = os.path.join(ROOT, "stereo-matching_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith(".hip") or name.endswith(".h"):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--disp", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on a one-GPU box: gloo backend, every rank on device 0")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher: start one rank per GPU ourselves, before this process initialises any GPU state
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the matcher has no CPU fallback")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = importlib.import_module("stereo-matching_amd")
    synth = importlib.import_module("stereo-matching_amd.synth")
    W, H, D = args.width, args.height, args.disp

    # two distinct pairs per rank, alternated, so no step re-reads the previous step's inputs
    pairs = [synth.pair(1000 + 2 * rank + i, W, H, D)[:2] for i in range(2)]
    d_l = [torch.from_numpy(p[0]).to(dev) for p in pairs]
    d_r = [torch.from_numpy(p[1]).to(dev) for p in pairs]
    d_out = torch.empty((H, W), dtype=torch.float32, device=dev)

    h = pkg.ADCensusStereo(local_rank)
    if not h.Initialize(W, H, pkg.ADCensusOption(max_disparity=D)):
        raise SystemExit("Initialize failed: " + h.last_error())
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        h.match_device(d_l[i & 1].data_ptr(), d_r[i & 1].data_ptr(), d_out.data_ptr(), stream)

    batch = importlib.import_module("stereo-matching_amd.batch")
    elapsed = batch.timed_steps(step, args.steps, args.warmup, torch.cuda.synchronize,
                                dist if world > 1 else None, None if args.rehearse else dev)

    valid_fraction = float(torch.isfinite(d_out).float().mean().item())

    # every volume kernel timed on its own: HIP events on the launch stream around each launch
    h.set_profiling(True)
    nprof = 5
    stage_acc, kern_acc = {}, {}
    for i in range(nprof):
        step(i)
        torch.cuda.synchronize()
        for k, v in h.stage_times().items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v / nprof
        for k, v in h.kernel_times().items():
            kern_acc[k] = kern_acc.get(k, 0.0) + v / nprof
    h.set_profiling(False)

    # the host-pointer boundary: adc_match on pinned host memory, transfers inside the wall time
    t_match = None
    if rank == 0:
        pin_l, pin_r = torch.from_numpy(pairs[0][0]).pin_memory(), torch.from_numpy(pairs[0][1]).pin_memory()
        pin_o = torch.empty((H, W), dtype=torch.float32).pin_memory()
        nl, nr, no = pin_l.numpy(), pin_r.numpy(), pin_o.numpy()
        h.set_profiling(True)
        for _ in range(2):
            assert h.Match(nl, nr, no), h.last_error()
        nrep, h2d, d2h = 5, 0.0, 0.0
        t0 = time.perf_counter()
        for _ in range(nrep):
            assert h.Match(nl, nr, no), h.last_error()
            x = h.transfer_times()
            h2d += x["h2d"] / nrep
            d2h += x["d2h"] / nrep
        wall = (time.perf_counter() - t0) / nrep * 1e3
        h.set_profiling(False)
        t_match = {"ms_per_frame": wall, "h2d_ms": h2d, "d2h_ms": d2h, "value": W * H * D * 1e-6 / (wall * 1e-3),
                   "unit": "Mpix*disp/s", "bytes_in": 2 * 3 * W * H, "bytes_out": 4 * W * H,
                   "note": "adc_match, images and map in pinned host memory, one frame at a time (upload, match, "
                           "download one after the other); PCIe-inclusive, never the headline value"}

    vol = float(W) * H * D                     # pixel*disparities per launch of a volume kernel

    def entry(kernel, names, bytes_per_unit, what):
        ms = float(np.mean([kern_acc[n] for n in names]))
        a = bytes_per_unit * vol / (ms * 1e-3) / 1e9
        return {"kernel": kernel, "launches": len(names), "launch_ms": ms, "algorithmic_bytes_per_launch": bytes_per_unit * vol,
                "model": what, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS}

    cost_fused = kern_acc["cost"] < 0.05 * kern_acc["aggr0"]
    aggr_all = ["aggr%d" % i for i in range(8)]
    per_kernel = []
    if cost_fused:
        per_kernel.append(entry("k_agg_pair<COST> (initial cost computed inside the first aggregation launch)", ["aggr0"], 8.0,
                                "cost-volume write 4 B + half an aggregation iteration 4 B"))
    else:
        per_kernel.append(entry("k_cost_init", ["cost"], 4.0, "cost-volume write 4 B"))
        per_kernel.append(entry("aggregation launch 0", ["aggr0"], 4.0, "half an aggregation iteration"))
    per_kernel.append(entry("k_agg_pair (aggregation launches 1..7)", aggr_all[1:], 4.0,
                            "half an aggregation iteration: the model charges 8 B per iteration, an iteration is two launches"))
    per_kernel.append(entry("scanline pass, horizontal (launches 0,1)", ["scan0", "scan1"], 8.0, "volume read 4 B + write 4 B"))
    per_kernel.append(entry("scanline pass, vertical (launches 2,3; the last also writes the left WTA map)", ["scan2", "scan3"], 8.0,
                            "volume read 4 B + write 4 B"))
    per_kernel.append(entry("k_wta_right", ["wta_right"], 4.0, "volume read 4 B"))

    # HBM bytes per aggregation launch as measured by the PMC passes under profiles/ -- only if they were taken
    # on exactly these kernel sources and this workload
    traffic, traffic_src = None, None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj["workload"] == {"width": W, "height": H, "disparities": D} and tj.get("source_stamp") == kernel_source_stamp():
            traffic = tj["aggregation_pass"]["hbm_bytes_per_launch"]
            traffic_src = tj.get("profiles")
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        ms_per_frame = elapsed / args.steps * 1e3
        agg = entry("k_agg_pair (one launch of the cross-aggregation kernel; 8 per match)", aggr_all, 4.0,
                    "8 B per pixel*disparity per aggregation iteration, two launches per iteration")
        agg_pass = 8.0 * vol / (agg["launch_ms"] * 1e-3) / 1e9
        out = {
            "metric": "Mpix*disp/s",
            "value": batch.throughput(world, args.steps, W, H, D, elapsed),
            "unit": "Mpix*disp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_frame,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "synthetic stereo pairs %dx%d, %d disparities, full AD-Census Match per step, "
                                   "one pair per GPU per step (frame-sharded replicas)" % (W, H, D),
                       "width": W, "height": H, "disparities": D, "pairs_per_step": world},
            "ms_per_frame": ms_per_frame,
            "whole_match_roofline_frac": (72.0 * W * H * D + 120.0 * W * H) / (ms_per_frame * 1e-3) / (HBM_PEAK_GBS * 1e9),
            "stage_ms": {k: round(v, 4) for k, v in stage_acc.items()},
            "kernel_ms": {k: round(v, 4) for k, v in kern_acc.items()},
            "valid_fraction": valid_fraction,
            "roofline": {"bound": "hbm", "kernel": agg["kernel"], "achieved": agg["achieved"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": agg["frac"], "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": agg["algorithmic_bytes_per_launch"], "launch_ms": agg["launch_ms"],
                         "per_pass": {"algorithmic_bytes_per_launch": 8.0 * vol, "achieved": agg_pass,
                                      "frac": agg_pass / HBM_PEAK_GBS,
                                      "note": "what one pass has to move on its own: read 4 B + write 4 B"}},
            "roofline_kernels": per_kernel,
            "t_match": t_match,
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as orc          # reported baseline only; never on the measured path
            nthr = orc.num_threads()
            o = orc.default_option(max_disparity=D)
            # all cores: whole frames until ~10 s of CPU time (1..4 frames)
            t1 = time.perf_counter()
            n_cpu = 0
            while n_cpu < 4 and (n_cpu == 0 or time.perf_counter() - t1 < 10.0):
                orc.match(pairs[n_cpu & 1][0], pairs[n_cpu & 1][1], o)
                n_cpu += 1
            cpu_s = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": n_cpu * W * H * D * 1e-6 / cpu_s, "unit": "Mpix*disp/s",
                                   "cores": nthr, "kind": "port",
                                   "sample": "%d whole %dx%dx%d pair(s), %.1f s of OpenMP CPU time on %d threads"
                                             % (n_cpu, W, H, D, cpu_s, nthr)}
            # one core: a band of full-width rows of the same pair (about a sixth of the frame)
            band = max(96, H // 6)
            orc.set_num_threads(1)
            t1 = time.perf_counter()
            orc.match(np.ascontiguousarray(pairs[0][0][:band]), np.ascontiguousarray(pairs[0][1][:band]), o)
            cpu1_s = time.perf_counter() - t1
            orc.set_num_threads(nthr)
            out["cpu_baseline_1thread"] = {"value": W * band * D * 1e-6 / cpu1_s, "unit": "Mpix*disp/s", "cores": 1, "kind": "port",
                                           "sample": "the top %d full-width rows of one %dx%dx%d pair, %.1f s on 1 thread"
                                                     % (band, W, H, D, cpu1_s)}
        print(json.dumps(out), flush=True)
    h.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
