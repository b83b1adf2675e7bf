#!/usr/bin/env python3
"""profiles/traffic.json from the two PMC summaries of tools/gpu_profile.sh:
    tools/make_traffic.py <fetch_summary.txt> <write_summary.txt> <name of the committed copies under profiles/>
HBM bytes per launch of the aggregation kernel = (2 * FETCH_SIZE + WRITE_SIZE) KiB, mean over the launches that read
a volume (on this chip FETCH_SIZE reports half of the bytes read, WRITE_SIZE the bytes written exactly: measured on
kernels that move a known 1 GiB in the volume kernels' access shapes, tools/microbench/pmc_calibration.hip, summary in
profiles/r03_pmc_calibration.txt). The file is stamped with a hash of the kernel sources: bench.py reports the figure
only while the sources it runs are the ones the counters were taken on."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def means(path, counter):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)", line)
            if m and m.group(1) == counter:
                out[name] = (int(m.group(2)), float(m.group(3)))
    return out


def main(fetch_txt, write_txt, tag):
    f, w = means(fetch_txt, "FETCH_SIZE"), means(write_txt, "WRITE_SIZE")
    # the volume-reading aggregation launches: every k_agg_pair instantiation except the cost-fused one (COST = true)
    def agg(d):
        n = s = 0
        for k, (cnt, mean) in d.items():
            if "k_agg_pair" in k and not re.search(r"k_agg_pair<\d, (true|false), true,", k):   # third argument: COST
                n += cnt
                s += cnt * mean
        return s / n
    fk, wk = agg(f), agg(w)
    per_kernel = {}
    for k in sorted(set(f) & set(w)):
        if "adc::" in k:
            per_kernel[k] = {"fetch_size_kib": f[k][1], "write_size_kib": w[k][1],
                             "hbm_bytes_per_launch": (2 * f[k][1] + w[k][1]) * 1024}
    out = {
        "note": "HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, KiB units; the "
                "read side doubled as gfx950 tallies 128-B read requests at 64 B). aggregation_pass: mean over the seven "
                "volume-reading aggregation launches of a match (the first launch computes its input).",
        "profiles": tag,
        "source_stamp": bench.kernel_source_stamp(),
        "workload": {"width": 1920, "height": 1080, "disparities": 256},
        "aggregation_pass": {"fetch_size_kib": fk, "write_size_kib": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024},
        "per_kernel": per_kernel,
    }
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out["aggregation_pass"]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
