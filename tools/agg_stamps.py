#!/usr/bin/env python3
"""Where do the waves of the fused aggregation kernel spend their cycles? Needs the diagnostic build
(tools/build_variant.sh stamp adc_aggregate.hip -DAGG_STAMP) named by ADC_LIB_PATH; prints the share of each
part of an iteration, summed over all waves of the fused launches of a few frames."""
import ctypes as C
import importlib
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault("ADC_LIB_PATH", os.path.join(root, "stereo-matching_amd", "lib", "variants", "libadc_stamp.so"))
pkg = importlib.import_module("stereo-matching_amd")
synth = importlib.import_module("stereo-matching_amd.synth")
adc = importlib.import_module("stereo-matching_amd.adcensus")
W, H, D = 1920, 1080, 256
l, r = synth.pair(1000, W, H, D)[:2]
dev = torch.device("cuda:0")
dl, dr = torch.from_numpy(l).to(dev), torch.from_numpy(r).to(dev)
out = torch.empty((H, W), dtype=torch.float32, device=dev)
h = pkg.ADCensusStereo(0)
assert h.Initialize(W, H, pkg.ADCensusOption(max_disparity=D))
s = torch.cuda.current_stream().cuda_stream
lib = C.CDLL(os.environ["ADC_LIB_PATH"])
buf = (C.c_ulonglong * 8)()
for i in range(2):
    h.match_device(dl.data_ptr(), dr.data_ptr(), out.data_ptr(), s)
torch.cuda.synchronize()
lib.adc_debug_agg_stamps(buf)
n = 3
for i in range(n):
    h.match_device(dl.data_ptr(), dr.data_ptr(), out.data_ptr(), s)
torch.cuda.synchronize()
lib.adc_debug_agg_stamps(buf)
if os.environ.get("AGG_STAMP_QUAD"):
    names = ["descriptors read from the staging buffer", "stage 1: window sums (matrix pipe)", "stage 1: divide, permute, into ring 2 (+ stage-2 skipped time)",
             "stage 2: window sums (matrix pipe)", "stage 2: permute, store", "wait + barrier"]
    tot = float(sum(buf[i] for i in range(6)))
    print("computing waves: %d  mean cycles per wave: %.0f" % (buf[7], tot / max(buf[7], 1)))
    for i, nm in enumerate(names):
        print("%-70s %6.2f %%" % (nm, 100.0 * buf[i] / tot))
    print("loader waves: kilocycles issuing %d, waiting for fills %d (sum over all loader waves; computing waves total %d)" % (buf[6] & 0xffffffff, buf[6] >> 32, int(tot) >> 10))
    sys.exit(0)
names = ["descriptor DMA + early fills issued", "stage 1 (windows of ring 1, divide, into ring 2) + late fills", "stage 2 (windows of ring 2, stores)",
         "wait for the previous iteration's fills", "barrier", "ring counters + next descriptors from the staging buffer"]
tot = float(sum(buf[i] for i in range(6)))
print("waves: %d  mean cycles per wave: %.0f (s_memtime ticks)" % (buf[7], tot / max(buf[7], 1)))
for i, nm in enumerate(names):
    print("%-52s %6.2f %%" % (nm, 100.0 * buf[i] / tot))
li, lw, lb = buf[6] & 0xfffff, (buf[6] >> 20) & 0xfffff, buf[6] >> 40
if li + lw + lb:
    print("loader waves: issuing %.1f %%, waiting for fills %.1f %%, at the barrier %.1f %% of their time"
          % (100.0 * li / (li + lw + lb), 100.0 * lw / (li + lw + lb), 100.0 * lb / (li + lw + lb)))
