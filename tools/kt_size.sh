#!/bin/bash
# tools/kt_size.sh <tag> <W> <H> <D>: kernel trace of bench.py at another frame size; per-kernel averages and the busy share.
set -o pipefail
tag=$1; W=$2; H=$3; D=$4
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp; cd /tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 "$root/bench.py" --steps 40 --warmup 5 --quick --width $W --height $H --disp $D > "$out/kt.log" 2>&1 || { tail "$out/kt.log"; exit 1; }
cd "$root"
find "$out/kt" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
find "$out/kt" -name '*kernel_trace.csv' -exec cp {} "$out/kernel_trace.csv" \;
rm -rf "$out/kt"
python3 - "$out/kernel_trace.csv" <<'PY'
import csv, sys
R = [r for r in csv.DictReader(open(sys.argv[1])) if 'adc::' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']]
R.sort(key=lambda r: int(r['Start_Timestamp']))
st = [i for i, r in enumerate(R) if 'k_pack_gray' in r['Kernel_Name']]
a, b = st[10], st[30]
fr = R[a:b]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in fr)
span = int(R[b]['Start_Timestamp']) - int(fr[0]['Start_Timestamp'])
print("frames 20, kernels/frame %.1f, span/frame %.1f us, sum of kernel durations/frame %.1f us" % (len(fr) / 20, span / 20e3, busy / 20e3))
acc = {}
for r in fr:
    n = r['Kernel_Name'].split('(')[0][-40:]
    acc.setdefault(n, [0, 0]); acc[n][0] += 1; acc[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-42s per frame: %4.1f launches %8.1f us" % (n, c / 20, t / 20e3))
PY
