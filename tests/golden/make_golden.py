"""Regenerates tests/golden/*.json|npz from the CPU oracle (run in the build container).

The reference itself holds no golden outputs, so these vectors are the oracle's; they pin the oracle
against drift and let the GPU box check full-size results without re-running the CPU path.
Usage: python tests/golden/make_golden.py [--skip-artroom]
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib  # noqa: E402

from conftest import artroom1, fnv1a64, middlebury2003  # noqa: E402
from oracle import oracle as orc  # noqa: E402

synth = importlib.import_module("stereo-matching_amd.synth")


def hashes(disp, taps):
    out = {"disp_final": fnv1a64(disp)}
    for k, v in taps.items():
        out[k] = fnv1a64(v)
    return out


def main():
    gold = {}
    # (1) Teddy crop, every tap in full
    L, R, gt = middlebury2003("teddy")
    Lc = np.ascontiguousarray(L[150:198, 200:264])
    Rc = np.ascontiguousarray(R[150:198, 200:264])
    opt = orc.default_option(max_disparity=16)
    disp, taps = orc.match(Lc, Rc, opt, want="all")
    np.savez_compressed(os.path.join(HERE, "teddy_crop_64x48x16.npz"), left=Lc, right=Rc, disp_final=disp, **taps)
    # (2) Teddy / Cones full, hashes + final maps + bad-2.0
    for name in ("teddy", "cones"):
        L, R, gt = middlebury2003(name)
        disp, taps = orc.match(L, R, orc.default_option(max_disparity=64), want="all")
        bad, rms = orc.bad_fraction(disp, gt, 4.0, 2.0)
        gold[name] = {"W": L.shape[1], "H": L.shape[0], "D": 64, "bad2": bad, "rms": rms, "hash": hashes(disp, taps)}
        np.savez_compressed(os.path.join(HERE, "%s_disp.npz" % name), disp_final=disp,
                            disp_left_wta=taps["disp_left_wta"], disp_right_wta=taps["disp_right_wta"])
    # (3) synthetic pairs
    for seed, W, H, dmin, dmax in [(7, 320, 200, 0, 64), (11, 203, 117, 0, 48), (13, 230, 150, 3, 133)]:
        L, R, _ = synth.pair(seed, W, H, dmax - dmin)
        disp, taps = orc.match(L, R, orc.default_option(min_disparity=dmin, max_disparity=dmax), want="all")
        gold["synth_%d" % seed] = {"W": W, "H": H, "dmin": dmin, "dmax": dmax, "input_hash": fnv1a64(L) + fnv1a64(R),
                                   "hash": hashes(disp, taps)}
    # (4) full-resolution 2021 scene and the 1080p synthetic bench pair
    if "--skip-artroom" not in sys.argv:
        L, R = artroom1()
        t = time.time()
        disp, taps = orc.match(L, R, orc.default_option(max_disparity=256),
                               want=["disp_left_wta", "disp_right_wta", "arms", "sup_count_h", "census_left"])
        print("artroom1 oracle: %.1f s on %d threads" % (time.time() - t, orc.num_threads()))
        gold["artroom1"] = {"W": L.shape[1], "H": L.shape[0], "D": 256, "hash": hashes(disp, taps),
                            "valid_fraction": float(np.isfinite(disp).mean())}
        rng = np.random.default_rng(2026)
        idx = rng.integers(0, disp.size, size=4096)
        np.savez_compressed(os.path.join(HERE, "artroom1_sample.npz"), idx=idx, disp_final=disp.ravel()[idx],
                            disp_left_wta=taps["disp_left_wta"].ravel()[idx])
        L, R, _ = synth.pair(1000, 1920, 1080, 256)
        t = time.time()
        disp, taps = orc.match(L, R, orc.default_option(max_disparity=256), want=["disp_left_wta", "disp_right_wta"])
        print("synth 1080p oracle: %.1f s" % (time.time() - t))
        gold["synth_1000_1080p"] = {"W": 1920, "H": 1080, "D": 256, "input_hash": fnv1a64(L) + fnv1a64(R),
                                    "hash": hashes(disp, taps)}
    else:
        old = json.load(open(os.path.join(HERE, "golden.json")))
        for k in ("artroom1", "synth_1000_1080p"):
            if k in old:
                gold[k] = old[k]
    json.dump(gold, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "golden.json"))


if __name__ == "__main__":
    main()
