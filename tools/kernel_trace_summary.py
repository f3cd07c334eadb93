#!/usr/bin/env python3
"""per-kernel launch durations of a rocprofv3 --kernel-trace csv, in order of first appearance: n, avg, min, max (us).  usage: python tools/kernel_trace_summary.py <dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:72]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    a = agg.setdefault(k, [0, 0.0, 1e9, 0.0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
for k, (n, t, lo, hi) in agg.items():
    print("%-72s n=%4d avg=%8.2f min=%8.2f max=%8.2f us" % (k, n, t / n, lo, hi))
