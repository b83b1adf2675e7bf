import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for W, H, D in ((450, 375, 64), (640, 480, 128), (1280, 720, 128), (1280, 720, 256)):
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "50", "--warmup", "5", "--width", str(W), "--height", str(H), "--disp", str(D)], capture_output=True, text=True, timeout=300)
    if p.returncode: print("FAILED", p.stderr[-800:]); sys.exit(1)
    j = json.loads(p.stdout.strip().splitlines()[-1])
    print("%dx%dx%d ms/frame %.3f  Gpix*disp/s %.1f  stages %s" % (W, H, D, j["ms_per_frame"], j["value"] / 1e3, j["stage_ms"]), flush=True)
