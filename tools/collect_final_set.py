#!/usr/bin/env python3
"""Copies the final measurement set that tools/gpu_final_set.sh left under gpurun_out/<tag>/ to profiles/<tag>_* and names the
counter profile in profiles/CURRENT (bench.py quotes its figures while the library's source hash equals the profile's).
Refuses a set whose files carry different source hashes.   python3 tools/collect_final_set.py <tag>"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    want = open(os.path.join(src, "source_hash.txt")).read().strip()
    prof = json.load(open(os.path.join(src, "profile.json")))
    assert prof["source_hash"] == want, (prof["source_hash"], want)
    name = "%s_profile_%dx%d.json" % (tag, prof["streams"], prof["frames"])
    shutil.copy(os.path.join(src, "profile.json"), os.path.join(dst, name))
    shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, "%s_kernel_stats_%dx%d.csv" % (tag, prof["streams"], prof["frames"])))
    open(os.path.join(dst, "CURRENT"), "w").write(name + "\n")
    copied = [name]
    for f in sorted(glob.glob(os.path.join(src, "*"))):
        b = os.path.basename(f)
        if b.startswith("bench_") and b.endswith(".json") and b != "bench_under_rocprof.json" or b.startswith("parity_config") and b.endswith(".json") \
                or b in ("timeline.txt", "loop_profile.txt", "loop_profile_383.txt", "ulp_census.json", "soak.jsonl"):
            text = open(f).read()
            assert want in text, "%s does not carry the set's source hash %s" % (b, want)
            shutil.copy(f, os.path.join(dst, "%s_%s" % (tag, b)))
            copied.append("%s_%s" % (tag, b))
    print("sources %s: %d files -> profiles/: %s" % (want, len(copied), " ".join(copied)))


if __name__ == "__main__":
    main(sys.argv[1])
