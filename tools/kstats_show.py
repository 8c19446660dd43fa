#!/usr/bin/env python3
"""Prints a rocprofv3 kernel_stats.csv compactly: kernel, calls, average ms, total ms (our kernels only)."""
import csv
import sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("k_"):
        rows.append((n, int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
tot = sum(x[3] for x in rows)
for n, c, a, t in sorted(rows, key=lambda x: -x[3]):
    print("%-22s calls %4d  avg %8.3f ms  total %9.2f ms  %5.1f %%" % (n, c, a, t, 100 * t / tot))
