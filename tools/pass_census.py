#!/usr/bin/env python3
"""How the reference's rate / distortion search spends its quantise + count passes, per workload (TEST INFRASTRUCTURE:
the oracle built with -DMP3O_CENSUS counts them).  k_loop reproduces every one of these passes (k_loop.hip), so this is
the kernel's pass budget: bisection probes (bin_search_StepSize, src/loop.c:2119-2140), first and extra passes of
inner_loop (src/loop.c:569-606), iterations of the distortion loop (src/loop.c:415-558).

    python3 tools/pass_census.py [--streams 8] [--frames 96] [--out profiles/r03_pass_census.json]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mp3common  # noqa: E402
import ref_coverage  # noqa: E402  (load_synth)

NAMES = ["granule_channels", "bisection_probes", "bisection_over_budget", "bisection_equal", "outer_iterations", "inner_first_passes",
         "inner_extra_passes", "inner_passes_all_zero", "outer_iterations_without_amplification", "bisection_probes_all_zero", "short_block_granule_channels"]
MIX48 = [64, 96, 128, 192, 256, 320]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--frames", type=int, default=96)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_pass_census.json"))
    args = ap.parse_args()
    so = os.path.join(ROOT, "oracle", "_build", "liboracle_census.so")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-DMP3O_CENSUS", "-shared", "-o", so, os.path.join(ROOT, "oracle", "mp3_oracle.c"), "-lm"], check=True)
    mp3common.ORACLE_SO = so
    orc = mp3common.Oracle()
    cen = (ctypes.c_longlong * 16).in_dll(orc.lib, "mp3o_census")
    synth = ref_coverage.load_synth()
    rows = []
    for name, rate, ch, kbps_of in (("configs[1] 44.1 kHz stereo 128 kbps", 44100, 2, lambda s: 128), ("configs[3] 48 kHz stereo 64-320 kbps", 48000, 2, lambda s: MIX48[s % 6]),
                                    ("configs[4] 32 kHz mono 64 kbps", 32000, 1, lambda s: 64)):
        for i in range(16):
            cen[i] = 0
        for s in range(args.streams):
            orc.encode(synth(args.frames * 1152, ch, rate, s * 37), rate, kbps_of(s), ch)
        c = {n: int(cen[i]) for i, n in enumerate(NAMES)}
        gc = c["granule_channels"]
        passes = c["bisection_probes"] + c["inner_first_passes"] + c["inner_extra_passes"] - c["granule_channels"]  # inner_loop's first pass of iteration 1 repeats the bisection's last probe
        row = {"workload": name, "streams": args.streams, "frames": args.frames, "counts": c,
               "per_granule_channel": {"bisection_probes": round(c["bisection_probes"] / gc, 2), "outer_iterations": round(c["outer_iterations"] / gc, 2),
                                       "inner_first_passes": round(c["inner_first_passes"] / gc, 2), "inner_extra_passes": round(c["inner_extra_passes"] / gc, 2),
                                       "distinct_passes": round(passes / gc, 2), "all_zero_passes": round((c["inner_passes_all_zero"] + c["bisection_probes_all_zero"]) / gc, 2)}}
        rows.append(row)
        print(json.dumps(row["per_granule_channel"]), name)
    json.dump({"what": "quantise + count passes of the reference's search per (granule, channel), counted by the oracle (-DMP3O_CENSUS)", "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
