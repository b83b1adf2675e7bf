#!/usr/bin/env python3
"""What would capturing a whole Match in a hipGraph buy at a small frame size? Captures one adc_match_device call with
torch's graph capture and replays it; prints ms per frame against the plain stream launches. (The replays carry the
captured frame's voting stamp, so they skip fewer unchanged voting entries than real frames would: an upper bound on
the graph's cost, not on its gain.)"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("stereo-matching_amd")
synth = importlib.import_module("stereo-matching_amd.synth")
W, H, D = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "450x375x64").split("x"))
N = 200
dev = torch.device("cuda:0")
l, r = synth.pair(1000, W, H, D)[:2]
dl, dr = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
out = torch.empty((H, W), dtype=torch.float32, device=dev)
h = pkg.ADCensusStereo(0)
assert h.Initialize(W, H, pkg.ADCensusOption(max_disparity=D)), h.last_error()
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    for _ in range(5):
        h.match_device(dl.data_ptr(), dr.data_ptr(), out.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    ref = out.clone()
    t0 = time.perf_counter()
    for _ in range(N):
        h.match_device(dl.data_ptr(), dr.data_ptr(), out.data_ptr(), s.cuda_stream)
    torch.cuda.synchronize()
    t_plain = (time.perf_counter() - t0) / N * 1e3
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    h.match_device(dl.data_ptr(), dr.data_ptr(), out.data_ptr(), s.cuda_stream)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    g.replay()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / N * 1e3
print("%dx%dx%d  stream launches %.3f ms/frame   graph replays %.3f ms/frame   identical maps: %s"
      % (W, H, D, t_plain, t_graph, bool(torch.equal(ref, out))))
