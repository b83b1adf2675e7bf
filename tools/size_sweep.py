#!/usr/bin/env python3
"""bench.py at other frame sizes: tools/size_sweep.py WxHxD [WxHxD ...] (default: a small-to-4K ladder)."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = sys.argv[1:] or ["450x375x64", "640x480x128", "1280x720x128", "1280x720x256", "1920x1080x256", "3840x2160x512"]
for sz in sizes:
    W, H, D = (int(v) for v in sz.split("x"))
    steps = "50" if W * H * D < 2e8 else "6"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--quick", "--steps", steps, "--warmup", "2",
                        "--width", str(W), "--height", str(H), "--disp", str(D)], capture_output=True, text=True, timeout=600)
    if p.returncode:
        print(sz, "FAILED", p.stderr[-800:])
        sys.exit(1)
    j = json.loads(p.stdout.strip().splitlines()[-1])
    print("%-16s ms/frame %8.3f  Gpix*disp/s %5.1f  frac of 72 B model %.2f  stages %s"
          % (sz, j["ms_per_frame"], j["value"] / 1e3, j["whole_match_roofline_frac"], j["stage_ms"]), flush=True)
