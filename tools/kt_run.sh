#!/bin/bash
# tools/kt_run.sh <tag>: rocprofv3 kernel trace of a short bench run; prints per-kernel average durations.
set -o pipefail
tag=$1
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 "$root/bench.py" --steps 10 --warmup 2 --quick > "$out/kt.log" 2>&1 || { tail "$out/kt.log"; exit 1; }
cd "$root"
find "$out/kt" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
find "$out/kt" -name '*kernel_trace.csv' -exec cp {} "$out/kernel_trace.csv" \;
rm -rf "$out/kt"
python - "$out/kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:26]:
    print("%-44s calls %4s avg_us %9.1f  %6s%%" % (r['Name'].split('(')[0][-44:], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'][:6]))
PY
