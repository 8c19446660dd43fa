#!/usr/bin/env python3
"""Layers I and II: counts, on the device, how often a value that came out of a transcendental lies so close to the
float rounding it feeds that a libm which is off by one ulp could decide it differently (diagnostic build of the library,
-DMP3MI_ULP_CENSUS: tools/gpu_ulp_census_l12.sh).  Per site: calls, "near" (inside the band a one-ulp error of every libm
result involved can move the value by) and "wide" (2^20 times wider: the statistics where "near" is too rare to be seen).
One full-width encode per layer at the bench workload (bench.py --layer N).  See tools/ulp_census.py for the Layer III path.

    python3 tools/ulp_census_l12.py [--layers 2 1] [--out gpurun_out/ulp_census_l12.json]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mp3common import L12Run, Mp3mi  # noqa: E402

SITES = {0: "phase: (float) atan2(-im, re)  [k_fft12]", 8: "c[j] = (float)(sqrt(t1^2 + t2^2) / t3), two sines and two cosines  [k12_psy]",
         9: "bc = (float)(tmn tb + nmt (1 - tb)), a logarithm  [k12_psy]", 10: "(float) exp(-bc ln10/10)  [k12_psy]",
         11: "(float)(4.342944819 log x) of the subband ratios  [k12_psy / k12_snr1]"}
# share of calls where glibc 2.35's result is not the correctly rounded one
GLIBC_MISROUND = {0: 1.1e-3, 8: 1.2e-3, 9: 3e-4, 10: 2e-4, 11: 3e-4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, nargs="*", default=[2, 1])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ulp_census_l12.json"))
    a = ap.parse_args()
    mp = Mp3mi()
    fns = [getattr(mp.lib, "mp3mi_debug_ulp_census_" + n, None) for n in ("fft", "l12")]
    if any(f is None for f in fns):
        raise SystemExit("this library is not the census build (MP3MI_LIB=... built with -DMP3MI_ULP_CENSUS)")
    rows = []
    for layer in a.layers:
        kbps, nf = (288, 1149) if layer == 1 else (160, 383)
        run = L12Run(mp, layer, 44100, kbps, "s", n_frames=nf, synth=(4096, 0))
        buf = (ctypes.c_ulonglong * 64)()
        for f in fns:
            f(buf)
        for i in range(64):
            buf[i] = 0
        run.encode()
        for f in fns:
            f(buf)
        run.close()
        frames = 4096 * nf
        sites, adv, exp = [], 0.0, 0.0
        for i, name in SITES.items():
            calls, near, wide = int(buf[3 * i]), int(buf[3 * i + 1]), int(buf[3 * i + 2])
            est = near if near else wide / 1048576.0
            adv += est
            exp += est * GLIBC_MISROUND[i] * 0.5
            sites.append({"site": name, "calls": calls, "near": near, "wide": wide, "near_estimate": round(est, 3),
                          "near_per_million_frames": round(est / frames * 1e6, 3)})
            print("layer %d  %-78s calls %13d  near %7d  wide %10d" % (layer, name, calls, near, wide), flush=True)
        rows.append({"layer": layer, "workload": "4096 x %d frames, 44.1 kHz stereo, %d kbps (bench.py --layer %d)" % (nf, kbps, layer), "frames": frames,
                     "sites": sites, "p_frame_differs_from_any_libm_within_one_ulp": adv / frames,
                     "p_frame_differs_from_glibc_2_35_estimate": exp / frames})
        print("layer %d: %d frames; per frame: <= %.2e against any libm within one ulp, ~ %.1e against glibc 2.35" % (layer, frames, adv / frames, exp / frames))
    json.dump({"what": __doc__.split("\n\n")[0], "rows": rows}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
