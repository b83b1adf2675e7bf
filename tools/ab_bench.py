#!/usr/bin/env python3
"""Run bench.py once per library variant (lib/variants/libadc_<name>.so; 'base' = the shipped library)
in a child process each and print the stage times side by side."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sys.argv[1:]:
    env = dict(os.environ)
    if name != "base":
        env["ADC_LIB_PATH"] = os.path.join(root, "stereo-matching_amd", "lib", "variants", "libadc_%s.so" % name)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--quick", "--steps", os.environ.get("AB_STEPS", "10")],
                       env=env, capture_output=True, text=True, timeout=300)
    if p.returncode != 0:
        print(name, "FAILED", p.stderr[-500:])
        sys.exit(1)
    j = json.loads(p.stdout.strip().splitlines()[-1])
    print("%-12s ms/frame %7.3f  launch_ms %6.3f  %s" % (name, j["ms_per_frame"], j["roofline"]["launch_ms"], j["stage_ms"]), flush=True)
