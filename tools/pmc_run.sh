#!/bin/bash
# tools/pmc_run.sh <tag> <counters...>: one rocprofv3 --pmc pass over a short bench run, summary printed.
set -o pipefail
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/raw" -- python3 "$root/bench.py" --steps 2 --warmup 1 --quick > "$out/log.txt" 2>&1 || { tail "$out/log.txt"; exit 1; }
cd "$root"
python tools/pmc_summary.py "$out/raw" > "$out/summary.txt"
rm -rf "$out/raw"
cat "$out/summary.txt"
