#!/usr/bin/env python3
"""Where does a synchronous adc_match (pinned host memory) spend the ~1.1 ms it takes beyond a resident-input Match?
Wall time per call against the device time between the first and the last kernel (stage 'total', from events), the
two transfers, and the same for resident inputs issued back to back and one at a time."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

pkg = importlib.import_module("stereo-matching_amd")
synth = importlib.import_module("stereo-matching_amd.synth")
W, H, D, N = 1920, 1080, 256, 20
L, R = synth.pair(1000, W, H, D)[:2]
dev = torch.device("cuda:0")
h = pkg.ADCensusStereo(0)
assert h.Initialize(W, H, pkg.ADCensusOption(max_disparity=D)), h.last_error()
pl, pr = torch.from_numpy(L).pin_memory(), torch.from_numpy(R).pin_memory()
po = torch.empty((H, W), dtype=torch.float32).pin_memory()
nl, nr, no = pl.numpy(), pr.numpy(), po.numpy()
dl, dr = torch.from_numpy(L).to(dev), torch.from_numpy(R).to(dev)
do = torch.empty((H, W), dtype=torch.float32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    h.Match(nl, nr, no)


def timed(fn, n=N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def resident_sync():
    h.match_device(dl.data_ptr(), dr.data_ptr(), do.data_ptr(), stream)
    torch.cuda.synchronize()


print("resident inputs, back to back        %.3f ms/frame" % timed(lambda: h.match_device(dl.data_ptr(), dr.data_ptr(), do.data_ptr(), stream)))
print("resident inputs, one at a time (sync) %.3f ms/frame" % timed(resident_sync))
print("adc_match, pinned host memory         %.3f ms/frame" % timed(lambda: h.Match(nl, nr, no)))
h.set_profiling(True)
tot = wall = 0.0
for _ in range(N):
    t0 = time.perf_counter()
    h.Match(nl, nr, no)
    wall += (time.perf_counter() - t0) * 1e3 / N
    tot += h.stage_times()["total"] / N
x = h.transfer_times()
print("adc_match with events: wall %.3f ms, first-to-last-kernel device time %.3f ms, h2d %.3f, d2h %.3f" % (wall, tot, x["h2d"], x["d2h"]))
tot = 0.0
for _ in range(N):
    resident_sync()
    tot += h.stage_times()["total"] / N
print("resident, one at a time, with events: device time %.3f ms" % tot)
for _ in range(N):
    h.match_device(dl.data_ptr(), dr.data_ptr(), do.data_ptr(), stream)
torch.cuda.synchronize()
print("resident, back to back, with events: device time of the last %.3f ms" % h.stage_times()["total"])
