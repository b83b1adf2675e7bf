# full GPU suite, the issue probe, a full default bench run and one SQ counter pass (run on the GPU box from the repo root)
set -o pipefail
bash tools/gpu_suite.sh || exit 1
mkdir -p gpurun_out/probe
timeout -k 10 120 tools/microbench/bin/issue_probe | tee gpurun_out/probe/issue_probe.txt || exit 1
timeout -k 10 500 python bench.py > gpurun_out/suite/bench.json 2> gpurun_out/suite/bench.err || { tail -20 gpurun_out/suite/bench.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open("gpurun_out/suite/bench.json").read().strip().splitlines()[-1])
print("ms/frame", round(j["ms_per_frame"], 3), "value", round(j["value"]), "value_pcie", round(j["value_pcie"]), "stage", j["stage_ms"])
print("kernel_ms", j["kernel_ms"])
print("t_match", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in j["t_match"].items() if k != "note"})
print("batch_256", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in j["batch_256"].items() if k != "note"})
print("c4", j.get("c4"))
print("cpu", j.get("cpu_baseline"))
PY
