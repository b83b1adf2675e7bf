#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv by kernel name: mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

def main(d):
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0][-60:]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name in sorted(acc):
        print(name)
        for c, v in sorted(acc[name].items()):
            print("    %-28s n=%4d mean=%16.1f" % (c, len(v), sum(v) / len(v)))

if __name__ == "__main__":
    main(sys.argv[1])
