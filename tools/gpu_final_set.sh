#!/bin/bash
# Runs on the GPU box: the round's FINAL MEASUREMENT SET from the sources as they stand, in stages (a gpurun call is limited to
# 20 minutes; every stage writes under gpurun_out/<tag>/ and stamps what it writes with the library's source hash):
#   a  profile   kernel trace + FETCH / WRITE passes + four SQ passes of the default bench command -> profile.json, kernel_stats.csv;
#                the kernel TIMELINE of the last step of two calls issued back to back -> timeline.txt
#   b  bench     the bench lines: the default command (20 steps: value, roofline, cpu_baseline, end_to_end, other_workloads),
#                configs 2 / 3 / 4, Layers II / I, two ranks on one GPU (plain `bench.py --gpus 2`)
#   c  diag      k_loop's phase profile and the one-ulp census (diagnostic builds in /tmp, never the product library)
#   d  parity    every stream of the four workloads against the oracle, every 32nd against the reference binary
#   e  soak      the same on other inputs of the generator (6 runs)
# tools/collect_final_set.py <tag> then copies the set to profiles/<tag>_* and names the profile in profiles/CURRENT.
# Usage: tools/gpu_final_set.sh <tag> <stages, e.g. ab>
tag=$1; stages=$2
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
hash=$(python3 -c "import ctypes; L = ctypes.CDLL('mp3-enc-bsd_amd/libmp3mi.so'); L.mp3mi_source_hash.restype = ctypes.c_char_p; print(L.mp3mi_source_hash().decode())")
echo "$hash" > $out/source_hash.txt
stamp() { python3 - "$1" "$hash" <<'PY'
import json, sys
p, h = sys.argv[1], sys.argv[2]
rows = [json.loads(x) for x in open(p) if x.startswith("{")]
with open(p, "w") as f:
    for d in rows:
        d.setdefault("source_hash", h)
        f.write(json.dumps(d) + "\n")
PY
}
case $stages in *a*)
  bash tools/gpu_round_profile.sh $tag > $out/profile_stage.log 2>&1 || { tail -5 $out/profile_stage.log; exit 1; }
  tail -14 $out/profile_stage.log
  rm -rf /tmp/fs_tl
  timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/fs_tl -o tl -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $out/tl.err
  python3 - $out/timeline.txt "$hash" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/fs_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
rows = sorted([r for r in rows if r[0].startswith("k_") and not r[0].startswith("k_synth")], key=lambda r: r[1])
loops = [i for i, r in enumerate(rows) if r[0] == "k_loop"]
i0 = max(0, loops[-6] - 6)  # from the last k_loop of the call before: the call boundary is part of the picture
t0 = rows[i0][1]
with open(sys.argv[1], "w") as o:
    o.write("# sources %s; rocprofv3 --kernel-trace -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1: the last call (issued back to back\n"
            "# with the one before it) from the last k_loop of the call before; ms from that point; q = HIP stream's queue\n" % sys.argv[2])
    for n, s, e, q in rows[i0:]:
        o.write("%-22s q%-3s %9.3f -> %9.3f  (%7.3f ms)\n" % (n, q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
    lp = [rows[i] for i in loops[-6:]]
    o.write("# k_loop ms: " + " ".join("%.1f" % ((e - s) / 1e6) for _, s, e, _ in lp) + " | between consecutive launches: " +
            " ".join("%.1f" % ((lp[i + 1][1] - lp[i][2]) / 1e6) for i in range(len(lp) - 1)) + "\n")
PY
  tail -2 $out/timeline.txt
;; esac
case $stages in *b*)
  python3 bench.py --steps 20 --warmup 2 > $out/bench_config1.json 2> $out/bench_config1.err; echo "config 1: rc $?"; stamp $out/bench_config1.json
  for c in 2 3 4; do python3 bench.py --config $c --steps 6 --warmup 1 > $out/bench_config$c.json 2> $out/bench_config$c.err; echo "config $c: rc $?"; stamp $out/bench_config$c.json; done
  for l in 2 1; do python3 bench.py --layer $l --steps 6 --warmup 1 > $out/bench_layer$l.json 2> $out/bench_layer$l.err; echo "layer $l: rc $?"; stamp $out/bench_layer$l.json; done
  MP3MI_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 3 --warmup 1 --streams 2048 > $out/bench_two_ranks_one_gpu.json 2> $out/bench_two_ranks.err; echo "two ranks: rc $?"; stamp $out/bench_two_ranks_one_gpu.json
  MP3MI_BENCH_ONE_GPU=1 python3 bench.py --gpus 4 --steps 2 --warmup 1 --streams 1024 > $out/bench_four_ranks_one_gpu.json 2> $out/bench_four_ranks.err; echo "four ranks: rc $?"; stamp $out/bench_four_ranks_one_gpu.json
  python3 - $out <<'PY'
import json, sys, glob, os
for p in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    for line in open(p):
        if line.startswith("{"):
            d = json.loads(line)
            print("%-34s %12s %s  %.2f ms/step  n_gpus %d  exact %s" % (os.path.basename(p), d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d["parity_spot_check"]["bit_exact"]))
PY
;; esac
case $stages in *c*)
  bash tools/gpu_loop_profile.sh $tag > /dev/null 2>&1; sed -i "1i # sources $hash (diagnostic build -DMP3MI_LOOP_PROFILE; 4096 x 48 frames = ONE chunk: a launch without a cost history)" $out/loop_profile.txt; tail -12 $out/loop_profile.txt
  # ... and on the REAL schedule: 383 frames = five chunks, pacing and placement live (the figures are the LAST launch's)
  MP3MI_LIB=/tmp/libmp3mi_prof.so timeout 300 python3 tools/loop_profile.py 383 > $out/loop_profile_383.txt 2>&1
  sed -i "1i # sources $hash (diagnostic build -DMP3MI_LOOP_PROFILE; 4096 x 383 frames = five chunks, the last launch's waves; phase cycles include what runs beside k_loop)" $out/loop_profile_383.txt; tail -8 $out/loop_profile_383.txt
  bash tools/gpu_ulp_census.sh $tag 1 3 4 > /dev/null 2>&1
  python3 - $out/ulp_census.json "$hash" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
if isinstance(d, dict):
    d["source_hash"] = sys.argv[2]
else:
    d = {"source_hash": sys.argv[2], "rows": d}
json.dump(d, open(sys.argv[1], "w"), indent=1)
PY
  tail -6 $out/ulp_census.txt
;; esac
case $stages in *d*)
  for c in 1 2 3 4; do
    python3 tools/full_parity.py --config $c --ref-every 32 --out $out/parity_config$c.json > $out/parity_config$c.log 2>&1; echo "parity config $c: rc $?"
    python3 - $out/parity_config$c.json "$hash" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); d["source_hash"] = sys.argv[2]; json.dump(d, open(sys.argv[1], "w"), indent=1)
print(d["frames_total"], "frames,", d["mismatching_streams"], "mismatching streams,", len(d["reference_binary_mismatches"]), "of", d["compared_with_reference_binary"], "differ from the reference binary")
PY
  done
;; esac
case $stages in *e*)
  bash tools/gpu_soak.sh ${tag}/soak 300000
  python3 - $out/soak.jsonl "$hash" <<'PY'
import json, sys
rows = [json.loads(x) for x in open(sys.argv[1]) if x.startswith("{")]
with open(sys.argv[1], "w") as f:
    for d in rows:
        d["source_hash"] = sys.argv[2]; f.write(json.dumps(d) + "\n")
print(sum(d["frames_total"] for d in rows), "frames,", sum(d["mismatching_streams"] for d in rows), "mismatching streams")
PY
;; esac
