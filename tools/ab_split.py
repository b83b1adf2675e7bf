import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for ver in ("1", "2", "1", "2"):
    env = dict(os.environ, ADC_SPLIT_VERSION=ver)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "10"], env=env, capture_output=True, text=True, timeout=300)
    if p.returncode: print(ver, "FAILED", p.stderr[-800:]); sys.exit(1)
    j = json.loads(p.stdout.strip().splitlines()[-1])
    print("split v%s ms/frame %.3f scan %.4f" % (ver, j["ms_per_frame"], j["stage_ms"]["scan"]), flush=True)
