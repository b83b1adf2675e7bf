#!/usr/bin/env python3
"""A/B of library variants at another frame size: tools/ab_size.py WxHxD name [name ...] ('base' = the shipped library)."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, D = (int(v) for v in sys.argv[1].split("x"))
for name in sys.argv[2:]:
    env = dict(os.environ)
    if name != "base":
        env["ADC_LIB_PATH"] = os.path.join(root, "stereo-matching_amd", "lib", "variants", "libadc_%s.so" % name)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--quick", "--steps", "100", "--warmup", "5",
                        "--width", str(W), "--height", str(H), "--disp", str(D)], env=env, capture_output=True, text=True, timeout=300)
    if p.returncode != 0:
        print(name, "FAILED", p.stderr[-500:])
        sys.exit(1)
    j = json.loads(p.stdout.strip().splitlines()[-1])
    print("%-12s ms/frame %7.3f  %s" % (name, j["ms_per_frame"], j["stage_ms"]), flush=True)
