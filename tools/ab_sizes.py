import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for W, H in ((2560, 1440), (1920, 1200)):
    for case in ("ADC_SCAN_HYBRID=0", "ADC_SCAN_HYBRID=2,ADC_HYB_MAIN=1024", "ADC_SCAN_HYBRID=2,ADC_HYB_MAIN=768", "ADC_SCAN_HYBRID=0", "ADC_SCAN_HYBRID=2,ADC_HYB_MAIN=1024"):
        env = dict(os.environ)
        for kv in case.split(","):
            k, v = kv.split("="); env[k] = v
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "6", "--width", str(W), "--height", str(H)], env=env, capture_output=True, text=True, timeout=300)
        if p.returncode: print(case, "FAILED", p.stderr[-800:]); sys.exit(1)
        j = json.loads(p.stdout.strip().splitlines()[-1])
        print("%dx%d %-40s ms/frame %.3f scan %.4f" % (W, H, case, j["ms_per_frame"], j["stage_ms"]["scan"]), flush=True)
