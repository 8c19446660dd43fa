#!/bin/bash
# Runs on the GPU box: the profile of `bench.py --layer N` in ONE call (the Layer I / II counterpart of gpu_round_profile.sh) --
#   1. rocprofv3 --kernel-trace --stats            (per-kernel calls / average / total time; the bench line under it)
#   2. rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes, one step: HBM bytes per kernel)
#   3. three SQ counter passes of one step          (instruction mix, wait shares)
# -- merged into gpurun_out/<tag>/profile_layer<N>.json with the library's source hash.  The program goes directly after `--`.
# Copy the result to profiles/<tag>_profile_layer<N>_<S>x<F>.json and name it in profiles/CURRENT_LAYER<N>: bench.py --layer N quotes
# its counters while the library's source hash is the profile's.
# Usage: tools/gpu_l12_profile.sh <tag> <layer>
tag=$1; layer=$2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
raw=/tmp/prof12_$tag
rm -rf $raw
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/kt -o kt -- python3 bench.py --layer $layer --no-cpu-baseline > $out/bench_under_rocprof_layer$layer.json 2> $out/kt.err || exit 1
cp $(find $raw/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats_layer$layer.csv
rm -f $out/kt.err
pass() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $raw/$name -o $name -- python3 bench.py --layer $layer --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/pmc_$name.err || return 1
  python3 tools/pmc_summary.py $raw/$name $out/pmc_$name.json > /dev/null
  rm -f $out/pmc_$name.err
  echo "pass $name done"
}
pass fetch FETCH_SIZE && pass write WRITE_SIZE &&
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH &&
pass b SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 &&
pass c SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA || exit 1
python3 - $out $layer <<'PY'
import csv, json, sys
out, layer = sys.argv[1], int(sys.argv[2])
bench = json.loads(open("%s/bench_under_rocprof_layer%d.json" % (out, layer)).read().strip().splitlines()[-1])
steps = bench["steps"] + bench["warmup"]
k = {}
for r in csv.DictReader(open("%s/kernel_stats_layer%d.csv" % (out, layer))):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("k_") or n.startswith("k12_"):
        k[n] = {"calls_in_trace": int(r["Calls"]), "launches_per_step": int(r["Calls"]) / steps, "avg_ms": float(r["AverageNs"]) / 1e6}
for p in ("fetch", "write", "a", "b", "c"):
    for n, v in json.load(open("%s/pmc_%s.json" % (out, p))).items():
        if n in k:
            k[n].update({a: b for a, b in v.items() if a != "dispatches"})
            k[n]["dispatches_per_step"] = v["dispatches"]
tot_r = tot_w = 0.0
for n, v in k.items():
    if "FETCH_SIZE" in v:
        v["hbm_read_GB_per_step"] = round(2 * v["FETCH_SIZE"] * 1024 / 1e9, 2)  # gfx950: FETCH_SIZE tallies 128-B requests at 64 B
        v["hbm_write_GB_per_step"] = round(v["WRITE_SIZE"] * 1024 / 1e9, 2)
        tot_r += v["hbm_read_GB_per_step"]; tot_w += v["hbm_write_GB_per_step"]
frames = bench["config"]["streams_per_gpu"] * bench["config"]["frames_per_stream"]
res = {"tag": out.split("/")[-1], "layer": layer, "source_hash": bench["roofline"]["source_hash"], "streams": bench["config"]["streams_per_gpu"],
       "frames": bench["config"]["frames_per_stream"], "commands": ["rocprofv3 --kernel-trace --stats -- python3 bench.py --layer N --no-cpu-baseline",
       "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | <8 SQ counters> -- python3 bench.py --layer N --steps 1 --warmup 0 --no-cpu-baseline (one pass each)"],
       "correction": "gfx950: hbm_read = 2 * FETCH_SIZE KiB (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact",
       "hbm_GB_per_step": {"read": round(tot_r, 1), "write": round(tot_w, 1), "total": round(tot_r + tot_w, 1)},
       "algorithmic_GB_per_step": round(bench["roofline"]["algorithmic_bytes_per_frame"] * frames / 1e9, 2),
       "bench_line_under_kernel_trace": bench, "kernels": k}
json.dump(res, open("%s/profile_layer%d.json" % (out, layer), "w"), indent=1)
print("source %s, layer %d, %d x %d: %.1f GB read + %.1f GB written per step (algorithmic %.2f GB); %.2f M frames/s under the kernel trace" % (
    res["source_hash"], layer, res["streams"], res["frames"], tot_r, tot_w, res["algorithmic_GB_per_step"], bench["value"] / 1e6))
for n, v in sorted(k.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches_per_step"]):
    print("%-14s avg %7.3f ms x %4.1f per step  read %6.2f GB  write %6.2f GB  VALU %.3g  SALU %.3g per step" % (
        n[:14], v["avg_ms"], v["launches_per_step"], v.get("hbm_read_GB_per_step", 0), v.get("hbm_write_GB_per_step", 0), v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0)))
PY
