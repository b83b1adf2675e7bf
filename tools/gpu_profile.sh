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
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 "$root/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$out/kt.log" 2>&1 || { tail "$out/kt.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_fetch.log" 2>&1 || { tail "$out/pmc_fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_write.log" 2>&1 || { tail "$out/pmc_write.log"; exit 1; }
cd "$root"
python tools/pmc_summary.py "$out/pmc_fetch" > "$out/pmc_fetch_summary.txt"
python tools/pmc_summary.py "$out/pmc_write" > "$out/pmc_write_summary.txt"
find "$out/kt" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
# raw traces are large; keep only the summaries
rm -rf "$out/kt" "$out/pmc_fetch" "$out/pmc_write"
head -40 "$out/kernel_stats.csv"
