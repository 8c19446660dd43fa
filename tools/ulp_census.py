#!/usr/bin/env python3
"""Counts, on the device, how often a value that came out of a transcendental lies so close to the rounding or the
comparison it feeds that a libm which is off by one ulp could decide it differently (diagnostic build of the library,
-DMP3MI_ULP_CENSUS: tools/gpu_ulp_census.sh).  Per site: calls, "near" (inside the band that a one-ulp error of every
libm result involved can move the value by) and "wide" (a band 2^20 times wider: the statistics for an estimate where
"near" is too rare to be observed).  One encode call per BASELINE config, full size.

The product computes these values with csrc/dmath.h, correctly rounded; the reference with glibc, which is within one
ulp.  A stream can differ from the reference's only if some "near" event coincides with a glibc result that is not the
correctly rounded one, in the unlucky direction.  So, per frame:
    P(differs from ANY libm that is within one ulp)  <=  near events per frame          (the adversarial bound)
    P(differs from glibc 2.35)                       ~=  sum over sites of near x P(glibc misrounds that function)
with near estimated as wide / 2^20 where no near event was seen.

    python3 tools/ulp_census.py [--configs 1 3 4] [--out gpurun_out/ulp_census.json]
"""
import argparse
import ctypes
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SITES = ["phase: (float) atan2(-im, re)  [k_cw]", "nb: (float)(ecb norm exp(-snr ln10/10)), log + exp  [k_psy]",
         "pe >= 1800 (63 logs)  [k_psy]", "(int)(pe * 3.1 - mean_bits)  [k_loop, ResvMaxBits]", "nint(8 ln sfm): 576 logs, exp, log  [k_prep, quantanf_init]",
         "(int)(log(x) / log 2)  [k_prep, calc_scfsi]", "(float)(cb + c_w e), c_w from two sines and two cosines  [k_part]",
         "the same, sum below the floats' normal range (calls only)  [k_part]"]
# measured against glibc 2.35 with mpmath as the judge (tools/gen_dmath_tables.py samples): the share of
# calls where glibc's result is not the correctly rounded one
GLIBC_MISROUND = [1.1e-3, 2e-4, 3e-4, 3e-4, 3e-4, 3e-4, 1.2e-3, 0.0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, nargs="*", default=[1, 3, 4])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ulp_census.json"))
    args = ap.parse_args()
    mp3 = importlib.import_module("mp3-enc-bsd_amd")
    L = mp3.lib()
    fns = [getattr(L, "mp3mi_debug_ulp_census_" + n, None) for n in ("fft", "psy", "prep", "loop")]
    if any(f is None for f in fns):
        raise SystemExit("this library is not the census build (MP3MI_LIB=... built with -DMP3MI_ULP_CENSUS)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    rows = []
    for cid in args.configs:
        cfg = bench.CONFIGS[cid]
        wl = bench.Workload(mp3, cfg, dev, 0)
        buf = (ctypes.c_ulonglong * 64)()
        for f in fns:
            f(buf)  # clear
        for i in range(64):
            buf[i] = 0
        wl.step()
        for f in fns:
            f(buf)
        frames = cfg["streams"] * cfg["frames"]
        sites = []
        adversarial = expected = 0.0
        for i, name in enumerate(SITES):
            calls, near, wide = int(buf[3 * i]), int(buf[3 * i + 1]), int(buf[3 * i + 2])
            est_near = near if near else wide / 1048576.0
            adversarial += est_near
            expected += est_near * GLIBC_MISROUND[i] * 0.5  # (half of the misroundings go the harmless way)
            sites.append({"site": name, "calls": calls, "near": near, "wide": wide, "near_estimate": round(est_near, 3),
                          "near_per_million_frames": round(est_near / frames * 1e6, 3)})
            print("config %d  %-72s calls %13d  near %6d  wide %9d  (near ~ %.2f)" % (cid, name, calls, near, wide, est_near))
        # site 12 (UC_CW_REACH): of the partitions that hold a near step of the c_w site, how many end in another float when
        # every near step's float is moved one ulp down / up (both at once: the worst case)?
        reach_parts, reach_hits = int(buf[3 * 12]), int(buf[3 * 12 + 1])
        print("config %d  %-72s partitions with a near step %d, of which the sum that leaves k_part changes: %d" % (cid, "c_w: does a one-ulp float survive its partition?", reach_parts, reach_hits))
        nb_reached, nb_changed = int(buf[3 * 13]), int(buf[3 * 13 + 1])
        print("config %d  %-72s thresholds the changed sums reach %d, of which come out as another float: %d" % (cid, "c_w: does it reach a threshold nb?  [k_psy]", nb_reached, nb_changed))
        row = {"config": cid, "workload": cfg["name"], "frames": frames, "sites": sites,
               "cw_near_steps_partitions": reach_parts, "cw_near_steps_partitions_whose_sum_changes": reach_hits,
               "cw_thresholds_reached_by_a_changed_sum": nb_reached, "cw_thresholds_that_change": nb_changed,
               "p_frame_differs_from_any_libm_within_one_ulp": adversarial / frames,
               "p_frame_differs_from_glibc_2_35_estimate": expected / frames}
        print("config %d: %d frames; per frame: <= %.2e against any libm within one ulp, ~ %.1e against glibc 2.35" % (
            cid, frames, row["p_frame_differs_from_any_libm_within_one_ulp"], row["p_frame_differs_from_glibc_2_35_estimate"]))
        rows.append(row)
        wl.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"what": __doc__.split("\n\n")[0], "device": torch.cuda.get_device_name(0), "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
