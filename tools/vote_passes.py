#!/usr/bin/env python3
"""Per-pass durations of k_vote / k_apply of one match from a rocprofv3 kernel trace: tools/vote_passes.py <kernel_trace.csv>"""
import csv
import sys

R = [r for r in csv.DictReader(open(sys.argv[1])) if "adc::" in r["Kernel_Name"]]
R.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(R) if "k_pack_gray" in r["Kernel_Name"]]
acc = {}
for a, b in zip(starts[2:-1], starts[3:]):
    j = 0
    for r in R[a:b]:
        n = r["Kernel_Name"].split("(")[0]
        if "k_vote" in n:
            acc.setdefault(j, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            j += 1
tot = 0.0
for j, v in sorted(acc.items()):
    m = sum(v) / len(v) / 1e3
    tot += m
    print("vote pass %2d  mean %7.1f us" % (j, m))
print("all passes  %7.1f us" % tot)
