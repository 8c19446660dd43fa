#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection.csv per kernel into a small JSON (the raw CSV has
one row per dispatch and counter and is too large to keep)."""
import collections
import csv
import glob
import json
import os
import sys


def main(outdir, dest):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "):
                k = k[5:]
            if not (k.startswith("k_") or k.startswith("k12_")):
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                calls[k] += 1
        os.remove(f)
    for f in glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True):
        os.remove(f)
    out = {k: dict(v, dispatches=calls[k]) for k, v in agg.items()}
    json.dump(out, open(dest, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
