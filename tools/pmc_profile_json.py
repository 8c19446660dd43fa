#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE summaries of tools/gpu_profile.sh into the committed profile
JSON that bench.py reads for roofline.traffic.  Usage: pmc_profile_json.py <gpurun_out/tag> <profiles/out.json> <streams> <frames>"""
import json, sys
src, dst, streams, frames = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
f = json.load(open(src + "/pmc_fetch.json")); w = json.load(open(src + "/pmc_write.json"))
out = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (two separate passes, tools/gpu_profile.sh)",
       "workload": "%d streams x %d frames, 44.1 kHz stereo 128 kbps" % (streams, frames), "streams": streams, "frames": frames,
       "units": "FETCH_SIZE/WRITE_SIZE are KiB summed over the dispatches of one step",
       "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section) -> hbm_read = 2*FETCH_SIZE; cross-check on a known byte count: k_mdct reads every subband granule twice (as this granule and as the next one's previous) plus one extra per run of 22 = 2.05*28.9 GB per step + 4.8 GB of block types; WRITE_SIZE is exact (k_filter writes 28.9 GB of subband samples per step).",
       "kernels": {}}
for k in f:
    if k not in w:
        continue
    d = f[k]["dispatches"]
    out["kernels"][k] = {"dispatches": d, "FETCH_SIZE_KiB": f[k]["FETCH_SIZE"], "WRITE_SIZE_KiB": w[k]["WRITE_SIZE"],
                         "hbm_read_GB_per_step": round(2 * f[k]["FETCH_SIZE"] * 1024 / 1e9, 2), "hbm_write_GB_per_step": round(w[k]["WRITE_SIZE"] * 1024 / 1e9, 2),
                         "hbm_bytes_per_launch": int((2 * f[k]["FETCH_SIZE"] + w[k]["WRITE_SIZE"]) * 1024 / d)}
json.dump(out, open(dst, "w"), indent=1)
for k, v in out["kernels"].items():
    print("%-14s read %7.2f GB  write %7.2f GB per step, %d launches" % (k, v["hbm_read_GB_per_step"], v["hbm_write_GB_per_step"], v["dispatches"]))
