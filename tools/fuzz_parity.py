#!/usr/bin/env python3
"""One-off fuzz on a GPU box: tools/fuzz_parity.py FIRST_SEED COUNT [large] -- random frame sizes, disparity ranges and
options (as tests/test_parity_gpu.py::test_random_configurations, wider ranges), every tap of the HIP path against the CPU
oracle. 'large': frames of 300..2100 x 200..1200 pixels (several rounds of workgroups over the chip, lines cut into
pieces, more rows than SIMDs) with 64..256 disparities. Prints the configurations that differ; exit code 1 if any."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from oracle import oracle as orc  # noqa: E402

pkg = importlib.import_module("stereo-matching_amd")
first, count = int(sys.argv[1]), int(sys.argv[2])
large = len(sys.argv) > 3 and sys.argv[3] == "large"
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(90000 + seed)
    W, H = int(rng.integers(1, 400)), int(rng.integers(1, 260))
    D = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 100, 127, 128, 129, 200, 256, 257, 300, 384, 512]))
    dmin = int(rng.integers(-20, 21))
    L1 = int(rng.choice([1, 2, 5, 17, 33, 34, 35, 44, 63, 64, 90, 140]))
    if large:
        W, H = int(rng.integers(300, 2101)), int(rng.integers(200, 1201))
        D = int(rng.choice([64, 100, 128, 192, 256]))
        L1 = int(rng.choice([17, 33, 34, 35, 44]))
    opt = dict(min_disparity=dmin, max_disparity=dmin + D, cross_L1=L1, cross_L2=int(rng.integers(1, L1 + 1)),
               cross_t1=int(rng.integers(5, 60)), cross_t2=int(rng.integers(2, 40)),
               so_p1=float(rng.uniform(0.2, 2.0)), so_p2=float(rng.uniform(1.0, 4.0)), so_tso=int(rng.integers(5, 30)),
               lrcheck_thres=float(rng.uniform(0.5, 2.0)), irv_ts=int(rng.integers(5, 40)), irv_th=float(rng.uniform(0.2, 0.6)),
               do_lr_check=bool(rng.integers(0, 4) > 0), do_filling=bool(rng.integers(0, 4) > 0),
               do_discontinuity_adjustment=bool(rng.integers(0, 2)))
    L = rng.integers(0, 256, size=(H, W, 3)).astype(np.uint8)
    k = int(rng.integers(1, 12))
    c = np.cumsum(np.pad(L.astype(np.int32), ((0, 0), (k, 0), (0, 0)), mode="edge"), axis=1)
    L = np.ascontiguousarray(((c[:, k:] - c[:, :-k]) // k).astype(np.uint8))
    R = np.ascontiguousarray(np.roll(L, -int(rng.integers(0, 12)), axis=1))
    if rng.integers(0, 2):
        R[:, ::7] = rng.integers(0, 256, size=R[:, ::7].shape).astype(np.uint8)   # mismatches: something to vote on
    h = pkg.ADCensusStereo(0)
    h.set_debug_taps(True)
    assert h.Initialize(W, H, pkg.ADCensusOption(**opt)), h.last_error()
    got_final = h.match(L, R)
    disp, want = orc.match(L, R, orc.default_option(**opt), want="all")
    want["disp_final"] = disp
    diff = []
    for name in want:
        g = got_final if name == "disp_final" else h.get_buffer(name)
        w = want[name]
        same = g.shape == w.shape and np.array_equal(g.view(np.uint8).reshape(-1), np.ascontiguousarray(w).view(np.uint8).reshape(-1))
        if not same:
            diff.append(name)
    h.close()
    if diff:
        bad += 1
        print("seed %d  %dx%d D=%d dmin=%d %s  DIFFER: %s" % (seed, W, H, D, dmin, opt, diff), flush=True)
    elif large or seed % 25 == 0:
        print("seed %d ok (%dx%dx%d, L1=%d)" % (seed, W, H, D, L1), flush=True)
print("%d of %d configurations differ" % (bad, count))
sys.exit(1 if bad else 0)
