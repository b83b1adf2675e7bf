#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points at the bench workload (1920x1080, 256 disparities):
adc_match called in a loop (upload, match, download one after the other) against adc_match_batch (the same
steps pipelined over two staging slots). Prints ms per frame for both; the resident-input rate is bench.py's."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (one HIP runtime per process: torch's, loaded first)

pkg = importlib.import_module("stereo-matching_amd")
synth = importlib.import_module("stereo-matching_amd.synth")
W, H, D, N = 1920, 1080, 256, int(os.environ.get("N_PAIRS", "24"))
pairs = [synth.pair(1000 + i, W, H, D)[:2] for i in range(2)]
lefts = [pairs[i % 2][0] for i in range(N)]
rights = [pairs[i % 2][1] for i in range(N)]
opt = pkg.ADCensusOption(max_disparity=D)

h = pkg.ADCensusStereo(0)
assert h.Initialize(W, H, opt), h.last_error()
ref = [h.match(lefts[i], rights[i]) for i in range(2)]       # warm-up, and the maps to compare with
t0 = time.perf_counter()
for i in range(N):
    h.match(lefts[i], rights[i])
t_loop = (time.perf_counter() - t0) / N
h.close()

pkg.match_batch(lefts[:4], rights[:4], opt, [0])             # warm-up (includes handle creation)
t0 = time.perf_counter()
out = pkg.match_batch(lefts, rights, opt, [0])
t_batch = (time.perf_counter() - t0) / N                     # includes one adc_create + adc_initialize per call
same = all(((o == ref[i % 2]) | ((o != o) & (ref[i % 2] != ref[i % 2]))).all() for i, o in enumerate(out))
print("adc_match loop      : %.2f ms/frame  (%.1f Gpix*disp/s)" % (t_loop * 1e3, W * H * D / t_loop / 1e9))
print("adc_match_batch x%d : %.2f ms/frame  (%.1f Gpix*disp/s), maps identical to adc_match: %s"
      % (N, t_batch * 1e3, W * H * D / t_batch / 1e9, same))
