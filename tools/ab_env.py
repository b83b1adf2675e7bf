#!/usr/bin/env python3
"""A/B over environment settings in one GPU call: tools/ab_env.py VAR=a VAR=b ... (each run twice, alternating)."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = sys.argv[1:]
for case in cases + cases:
    env = dict(os.environ)
    env["ADC_DEBUG_ENV"] = "1"   # the library reads its ADC_* switches only with this set
    for kv in case.split(","):
        k, v = kv.split("=")
        env[k] = v
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--quick", "--steps", os.environ.get("AB_STEPS", "10")],
                       env=env, capture_output=True, text=True, timeout=300)
    if p.returncode:
        print(case, "FAILED", p.stderr[-800:])
        sys.exit(1)
    j = json.loads(p.stdout.strip().splitlines()[-1])
    k = j["kernel_ms"]
    print("%-28s ms/frame %.3f  aggr0/1/3/5/7 %.3f %.3f %.3f %.3f %.3f  %s"
          % (case, j["ms_per_frame"], k["aggr0"], k["aggr1"], k["aggr3"], k["aggr5"], k["aggr7"], j["stage_ms"]), flush=True)
