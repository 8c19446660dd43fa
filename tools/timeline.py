#!/usr/bin/env python3
"""Print the kernel timeline (start/end in ms, queue) of the last step from a rocprofv3 kernel_trace.csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r["Kernel_Name"].split("(")[0].replace("void ", "").startswith("k_")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    print("%-10s q%-3s %9.3f -> %9.3f  (%8.3f ms)" % (r["Kernel_Name"].split("(")[0].replace("void ", "")[:10], r.get("Queue_Id", "?"),
          (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
