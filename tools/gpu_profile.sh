#!/bin/bash
# Run on the GPU box from the repo root: GPU tests, bench, kernel trace and the two HBM-traffic PMC passes.
# Usage: tools/gpu_profile.sh <tag>      (outputs under gpurun_out/<tag>/)
set -o pipefail
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > "$out/pytest.log" 2>&1 || { tail -30 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
fi
timeout -k 10 300 python bench.py > "$out/bench.json" 2> "$out/bench.err" || { tail "$out/bench.err"; exit 1; }
cat "$out/bench.json"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 "$root/bench.py" --steps 10 --warmup 2 --quick > "$out/kt.log" 2>&1 || { tail "$out/kt.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$root/bench.py" --steps 2 --warmup 1 --quick > "$out/pmc_fetch.log" 2>&1 || { tail "$out/pmc_fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$root/bench.py" --steps 2 --warmup 1 --quick > "$out/pmc_write.log" 2>&1 || { tail "$out/pmc_write.log"; exit 1; }
cd "$root"
python tools/pmc_summary.py "$out/pmc_fetch" > "$out/pmc_fetch_summary.txt"
python tools/pmc_summary.py "$out/pmc_write" > "$out/pmc_write_summary.txt"
python tools/make_traffic.py "$out/pmc_fetch_summary.txt" "$out/pmc_write_summary.txt" "profiles/${tag}_pmc_fetch_summary.txt, profiles/${tag}_pmc_write_summary.txt" || exit 1
cp profiles/traffic.json "$out/traffic.json"
find "$out/kt" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
# per-launch durations of the volume kernels of one match, in launch order (from the trace, before it is deleted)
python - "$out/kt" > "$out/volume_kernels_by_launch.txt" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
R = [r for r in csv.DictReader(open(f)) if "adc::" in r["Kernel_Name"]]
R.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(R) if "k_pack_gray" in r["Kernel_Name"]]
acc = {}
for a, b in zip(starts[2:-1], starts[3:]):
    j = 0
    for r in R[a:b]:
        n = r["Kernel_Name"].split("(")[0]
        if any(k in n for k in ("k_agg", "k_scanline", "k_wta_right", "k_cost_init")):
            acc.setdefault((j, n[-60:]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            j += 1
for (j, n), v in sorted(acc.items()):
    print("%2d %-62s n=%3d mean %8.1f us  min %8.1f  max %8.1f" % (j, n, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
PY
# raw traces are large; keep only the summaries
rm -rf "$out/kt" "$out/pmc_fetch" "$out/pmc_write"
head -40 "$out/kernel_stats.csv"
