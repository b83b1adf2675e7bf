#!/bin/bash
# tools/h_sweep.sh <tag> <heights...>: kernel-trace bench.py at several image heights (env passes through),
# print the volume kernels' average durations. Shows how a pass's time steps with scanlines per SIMD.
set -o pipefail
tag=$1; shift
root=$(pwd); export TMPDIR=/tmp; cd /tmp
for hh in "$@"; do
  out=$root/gpurun_out/hs_${tag}_$hh; mkdir -p $out
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/bench.py --steps 6 --warmup 2 --quick --height $hh > $out/log.txt 2>&1 || exit 1
  find $out/kt -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
  rm -rf $out/kt
  python3 - $out/kernel_stats.csv $hh <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'k_scanline' in n or 'k_agg_pair' in n:
        print("H=%s %-40s calls %3s avg_us %8.1f" % (sys.argv[2], n.split('(')[0][-40:], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
