#!/usr/bin/env python3
"""Prints per-kernel averages of the PMC passes written by tools/pmc_raster.sh (gpurun_out/pmc1, pmc2)."""
import collections
import csv
import sys
pat = sys.argv[1] if len(sys.argv) > 1 else "raster5"
for d in ("pmc1", "pmc2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    try:
        rows = list(csv.DictReader(open("gpurun_out/%s/p_counter_collection.csv" % d)))
    except OSError as e:
        print(d, e)
        continue
    for r in rows:
        k = r["Kernel_Name"]
        if pat not in k:
            continue
        agg[(k[:70], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, c in agg.items():
        print(key)
        for n, v in c.items():
            print("   %-24s %14.0f" % (n, sum(v) / len(v)))
