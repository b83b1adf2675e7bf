#!/usr/bin/env python3
"""How far is the canonical output from what the shipped CUDA binary most likely produced?

Runs the CPU oracle on Teddy and Cones (450x375, 64 disparities) once canonically and once per
literal-defect switch (and with all of them), and prints bad-2.0 (the reference's
PerformanceEval0103, performance_eval.cpp:218-253) plus the share of pixels whose final disparity
differs from the canonical map. A report only: nothing gates on these numbers, and the product
library implements the canonical semantics alone.

CPU only (about a minute on 8 cores):  python tools/literal_sensitivity.py [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402  (test infrastructure; this tool is a study, not the product)

CASES = [
    ("canonical", 0),
    ("#5 horizontal passes return early", oracle.LIT_HPASS_RETURN),
    ("#6 nested MIN macro", oracle.LIT_MIN_MACRO),
    ("#7 nested MAX macro", oracle.LIT_MAX_MACRO),
    ("#8 low sentinel 0x9F9F9F9F", oracle.LIT_LOW_SENTINEL),
    ("#12 interpolation return", oracle.LIT_INTERP_RETURN),
    ("all five", oracle.LIT_ALL),
]


def load(name):
    from PIL import Image
    d = os.path.join(ROOT, "tests", "data", name)
    bgr = lambda f: np.ascontiguousarray(np.asarray(Image.open(os.path.join(d, f)).convert("RGB"))[:, :, ::-1])
    return bgr("im2.png"), bgr("im6.png"), np.ascontiguousarray(np.asarray(Image.open(os.path.join(d, "disp2.png"))))


def same(a, b):
    return (a == b) | (np.isinf(a) & np.isinf(b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    args = ap.parse_args()
    rows = []
    for scene in ("teddy", "cones"):
        L, R, gt = load(scene)
        base = None
        for label, flags in CASES:
            with oracle.literal(flags):
                disp, _ = oracle.match(L, R, oracle.default_option(max_disparity=64))
            if base is None:
                base = disp
            bad, rms = oracle.bad_fraction(disp, gt, 4.0, 2.0)
            diff = 1.0 - float(same(disp, base).mean())
            rows.append({"scene": scene, "case": label, "flags": flags, "bad2": bad, "rms": rms,
                         "pixels_differing_from_canonical": diff,
                         "invalid_fraction": float(np.isinf(disp).mean())})
            print("%-6s %-36s bad-2.0 %.4f  rms %.3f  differs %.4f  invalid %.4f"
                  % (scene, label, bad, rms, diff, rows[-1]["invalid_fraction"]), flush=True)
    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
