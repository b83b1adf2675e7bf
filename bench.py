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
  roofline         -- the dominant kernel (the roofline_kernels entry with the largest share of a match). Algorithmic
                      bytes per launch follow the stage-boundary model: an aggregation ITERATION reads and writes
                      the volume once (8 B per pixel*disparity), i.e. 4 B per pass; a launch that runs two passes
                      (the fused aggregation kernel) carries 8 B, one that runs one pass 4 B. Launch durations are
                      measured per launch with HIP events on the launch stream.
  roofline_kernels -- the same accounting for every volume kernel of a match, launch by launch.
  aggregation_stage-- all aggregation launches together against the model's 32 B (+4 B for the cost volume).
  t_match          -- the host-pointer boundary (adc_match) on pinned host images: wall time per frame with the
                      upload of the two images and the download of the map inside, and those two separately;
                      value_pcie is its rate (the reference's own `speed` includes everything, ADC/main.cpp:221-230).
  batch_256        -- adc_match_batch over 256 host-resident pairs (8 distinct ones) on this rank's initialised handle:
                      the reference's dataset loop (ADC/main.cpp:62-255) with uploads, matches and downloads of
                      neighbouring pairs overlapped; every rank runs it on its own GPU, the rate is the sum over ranks.
  c4               -- one synthetic 3840x2160 pair with 512 disparities (two 17 GB volumes), rank 0 at N = 1 only.
  cpu_baseline     -- the CPU oracle (a port of the reference path; the reference has no CPU-only path of its
                      own) timed on a bounded sample of the same workload on all host cores, and
                      cpu_baseline_1thread on one core; cpu_baseline_small_synthetic: the same for whole synthetic
                      450x375x64 pairs (the small configuration's size); rank 0, N = 1 only.
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


def c4_leg(pkg, synth, torch, dev, device_index):
    """One synthetic 3840x2160 pair, 512 disparities (two 17 GB volumes resident in HBM): ms per Match with the inputs
    on the device, the stage and volume-kernel times of a profiled Match, and the rate against the byte model."""
    W, H, D = 3840, 2160, 512
    L, R = synth.pair(2026, W, H, D)[:2]
    dl, dr = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
    do = torch.empty((H, W), dtype=torch.float32, device=dev)
    h = pkg.ADCensusStereo(device_index)
    if not h.Initialize(W, H, pkg.ADCensusOption(max_disparity=D)):
        raise SystemExit("c4: Initialize failed: " + h.last_error())
    stream = torch.cuda.current_stream().cuda_stream
    h.match_device(dl.data_ptr(), dr.data_ptr(), do.data_ptr(), stream)
    torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        h.match_device(dl.data_ptr(), dr.data_ptr(), do.data_ptr(), stream)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    h.set_profiling(True)
    h.match_device(dl.data_ptr(), dr.data_ptr(), do.data_ptr(), stream)
    torch.cuda.synchronize()
    res = {"workload": "synthetic 3840x2160 pair (seed 2026), 512 disparities, full Match, inputs resident in HBM",
           "ms_per_frame": ms, "value": W * H * D * 1e-6 / (ms * 1e-3), "unit": "Mpix*disp/s",
           "whole_match_roofline_frac": (72.0 * W * H * D + 120.0 * W * H) / (ms * 1e-3) / (HBM_PEAK_GBS * 1e9),
           "stage_ms": {k: round(v, 3) for k, v in h.stage_times().items()},
           "kernel_ms": {k: round(v, 3) for k, v in h.kernel_times().items()},
           "valid_fraction": float(torch.isfinite(do).float().mean().item())}
    h.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--disp", type=int, default=256)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch-pairs", type=int, default=256, help="pairs of the host-fed batch leg (0 = skip it)")
    ap.add_argument("--no-c4", action="store_true", help="skip the 3840x2160x512 leg")
    ap.add_argument("--quick", action="store_true", help="the resident-input steps and the per-kernel times only "
                    "(no CPU baseline, no batch leg, no 3840x2160x512 leg): what the A/B and profiling tools run")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on a one-GPU box: gloo backend, every rank on device 0")
    args = ap.parse_args()
    if args.quick:
        args.no_cpu_baseline, args.no_c4, args.batch_pairs = True, True, 0

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher: start one rank per GPU ourselves, before this process initialises any GPU state
        port = os.environ.get("MASTER_PORT")
        if not port:                       # any free port of this host
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = str(sk.getsockname()[1])
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port,
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Under a launcher (WORLD_SIZE set) the process group is always formed, also with a single rank: the one-rank run is
    # then the same code as the N-rank run (RCCL init on the device, barrier, max-over-ranks on a device tensor).
    use_dist = env_world is not None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the matcher has no CPU fallback")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if use_dist:
        backend = "gloo" if args.rehearse else "nccl"
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
                                dist if use_dist else None, None if args.rehearse else dev)

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
        for _ in range(2):
            assert h.Match(nl, nr, no), h.last_error()
        nrep, h2d, d2h = 10, 0.0, 0.0
        t0 = time.perf_counter()
        for _ in range(nrep):                  # the wall time: no events between the kernels
            assert h.Match(nl, nr, no), h.last_error()
        wall = (time.perf_counter() - t0) / nrep * 1e3
        h.set_profiling(True)                  # the two transfers on their own, from profiled calls
        for _ in range(3):
            assert h.Match(nl, nr, no), h.last_error()
            x = h.transfer_times()
            h2d += x["h2d"] / 3
            d2h += x["d2h"] / 3
        h.set_profiling(False)
        t_match = {"ms_per_frame": wall, "h2d_ms": h2d, "d2h_ms": d2h, "value": W * H * D * 1e-6 / (wall * 1e-3),
                   "unit": "Mpix*disp/s", "bytes_in": 2 * 3 * W * H, "bytes_out": 4 * W * H,
                   "note": "adc_match, images and map in pinned host memory, one frame at a time (upload, match, "
                           "download one after the other); PCIe-inclusive, never the headline value"}

    # the host-fed batch: adc_match_batch over pairs in ordinary host memory on this rank's handle (every rank)
    batch_leg = None
    if args.batch_pairs > 0:
        n_distinct = 8
        bp = [pairs[i] if i < 2 else synth.pair(2000 + n_distinct * rank + i, W, H, D)[:2] for i in range(n_distinct)]
        outs = [np.empty((H, W), np.float32) for _ in range(n_distinct)]
        n = args.batch_pairs
        lefts, rights, dst = [bp[i % n_distinct][0] for i in range(n)], [bp[i % n_distinct][1] for i in range(n)], [outs[i % n_distinct] for i in range(n)]
        opt = pkg.ADCensusOption(max_disparity=D)
        pkg.match_batch(lefts[:4], rights[:4], opt, [local_rank], handles=[h], out=dst[:4])      # second staging set allocated
        b_elapsed = batch.timed_steps(lambda i: pkg.match_batch(lefts, rights, opt, [local_rank], handles=[h], out=dst), 1, 0,
                                      torch.cuda.synchronize, dist if use_dist else None, None if args.rehearse else dev)
        batch_leg = {"pairs_per_gpu": n, "distinct_pairs": n_distinct, "n_gpus": world, "seconds": b_elapsed,
                     "ms_per_frame": b_elapsed / n * 1e3, "pairs_per_s": world * n / b_elapsed,
                     "value": world * n * W * H * D * 1e-6 / b_elapsed, "unit": "Mpix*disp/s",
                     "note": "adc_match_batch on caller handles, pairs in pageable host memory: host copy into pinned staging, "
                             "upload, match and download of neighbouring pairs overlap (three streams per GPU)"}

    vol = float(W) * H * D                     # pixel*disparities per launch of a volume kernel

    def entry(kernel, names, bytes_per_unit, what):
        ms = float(np.mean([kern_acc[n] for n in names]))
        a = bytes_per_unit * vol / (ms * 1e-3) / 1e9
        return {"kernel": kernel, "launches": len(names), "launch_ms": ms, "algorithmic_bytes_per_launch": bytes_per_unit * vol,
                "model": what, "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS}

    cost_fused = kern_acc["cost"] < 0.05 * kern_acc["aggr0"]
    # large frames run the passes 1+2, 3+4, 5+6 as one launch each (ids 1, 3, 5; ids 2, 4, 6 then take no time)
    pairs_fused = all(kern_acc["aggr%d" % i] < 0.05 * kern_acc["aggr%d" % (i - 1)] for i in (2, 4, 6))
    per_kernel = []
    if cost_fused:
        per_kernel.append(entry("k_agg_pair<COST> (initial cost computed inside the first aggregation launch)", ["aggr0"], 8.0,
                                "cost-volume write 4 B + half an aggregation iteration 4 B"))
    else:
        per_kernel.append(entry("k_cost_init", ["cost"], 4.0, "cost-volume write 4 B"))
        per_kernel.append(entry("aggregation launch 0", ["aggr0"], 4.0, "half an aggregation iteration"))
    if pairs_fused:
        per_kernel.append(entry("k_agg_fused (passes 1+2, 3+4, 5+6: a dividing pass and the same-axis pass behind it per launch)",
                                ["aggr1", "aggr3", "aggr5"], 8.0,
                                "two half iterations: the model charges 8 B per iteration = 4 B per pass; the volume between "
                                "the two passes stays in LDS, so the launch really moves 8 B"))
        per_kernel.append(entry("k_agg_pair (last aggregation pass)", ["aggr7"], 4.0, "half an aggregation iteration"))
        agg_names = ["aggr0", "aggr1", "aggr3", "aggr5", "aggr7"]
    else:
        per_kernel.append(entry("k_agg_pair (aggregation launches 1..7)", ["aggr%d" % i for i in range(1, 8)], 4.0,
                                "half an aggregation iteration: the model charges 8 B per iteration, an iteration is two launches"))
        agg_names = ["aggr%d" % i for i in range(8)]
    per_kernel.append(entry("scanline pass, horizontal (launches 0,1)", ["scan0", "scan1"], 8.0, "volume read 4 B + write 4 B"))
    per_kernel.append(entry("scanline pass, vertical (launches 2,3; the last also writes the left WTA map)", ["scan2", "scan3"], 8.0,
                            "volume read 4 B + write 4 B"))
    per_kernel.append(entry("k_wta_right", ["wta_right"], 4.0, "volume read 4 B"))
    aggr_stage = {"launches": len(agg_names), "ms": float(sum(kern_acc[n] for n in agg_names)),
                  "algorithmic_bytes": (36.0 if cost_fused else 32.0) * vol}
    aggr_stage["achieved"] = aggr_stage["algorithmic_bytes"] / (aggr_stage["ms"] * 1e-3) / 1e9
    aggr_stage["frac"] = aggr_stage["achieved"] / HBM_PEAK_GBS

    # HBM bytes per aggregation launch as measured by the PMC passes under profiles/ -- only if they were taken
    # on exactly these kernel sources and this workload
    traffic_json = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj["workload"] == {"width": W, "height": H, "disparities": D} and tj.get("source_stamp") == kernel_source_stamp():
            traffic_json = tj
    except (OSError, KeyError, ValueError):
        pass

    if rank == 0:
        ms_per_frame = elapsed / args.steps * 1e3
        # the dominant kernel: the entry with the largest share of the match
        dom = max(per_kernel, key=lambda e: e["launches"] * e["launch_ms"])
        traffic, traffic_src = None, None
        if traffic_json:
            import re
            pat = (r"k_agg_fused" if "k_agg_fused" in dom["kernel"] else
                   r"k_agg_pair<\d, (true|false), false," if "k_agg_pair" in dom["kernel"] else
                   r"k_scanline" if "scanline" in dom["kernel"] else r"k_wta_right")
            hits = [v["hbm_bytes_per_launch"] for name, v in traffic_json.get("per_kernel", {}).items() if re.search(pat, name)]
            if hits:
                traffic = float(np.mean(hits))
                traffic_src = traffic_json.get("profiles")
        out = {
            "metric": "Mpix*disp/s",
            "value": batch.throughput(world, args.steps, W, H, D, elapsed),
            "value_pcie": t_match["value"] if t_match else None,
            "unit": "Mpix*disp/s",
            "n_gpus": world,
            "dist_backend": backend,
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
            "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": dom["frac"], "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "launch_ms": dom["launch_ms"],
                         "launches_per_match": dom["launches"], "model": dom["model"]},
            "aggregation_stage": aggr_stage,
            "roofline_kernels": per_kernel,
            "t_match": t_match,
            "batch_256": batch_leg,
        }
        if batch_leg:
            batch_leg["efficiency_vs_resident"] = ms_per_frame / batch_leg["ms_per_frame"]
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
            # the small configuration (a pair of Teddy's size, 64 disparities), whole frames, all cores and one
            sw, sh, sd = 450, 375, 64
            sl, sr = synth.pair(77, sw, sh, sd)[:2]
            so = orc.default_option(max_disparity=sd)
            small = {}
            for label, nt in (("all", nthr), ("one", 1)):
                orc.set_num_threads(nt)
                t1 = time.perf_counter()
                reps = 0
                while reps < 8 and (reps == 0 or time.perf_counter() - t1 < 2.0):
                    orc.match(sl, sr, so)
                    reps += 1
                small[label] = reps * sw * sh * sd * 1e-6 / (time.perf_counter() - t1)
            orc.set_num_threads(nthr)
            out["cpu_baseline_small_synthetic"] = {"value": small["all"], "value_1thread": small["one"], "unit": "Mpix*disp/s",
                                         "cores": nthr, "kind": "port",
                                         "sample": "whole synthetic %dx%dx%d pairs (the size of the Middlebury 2003 configuration)"
                                                   % (sw, sh, sd)}
        if world == 1 and not args.no_c4:
            h.close()                               # its volumes make room
            out["c4"] = c4_leg(pkg, synth, torch, dev, local_rank)
        print(json.dumps(out), flush=True)
    h.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
