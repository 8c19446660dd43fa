#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of one bench step; prints the dispatches of the last step as a timeline
# (start, end in ms relative to the step's first kernel).  Usage: tools/gpu_timeline.sh <tag> [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
raw=/tmp/tl_$tag
rm -rf $raw
cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --kernel-trace --output-format csv -d $raw -o tl -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > /dev/null 2> $out/tl.err
python3 - <<PY > $out/timeline.txt
import csv, glob
f = glob.glob("$raw/**/*kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows = [r for r in rows if r[0].startswith("k_") and not r[0].startswith("k_synth")]
rows.sort(key=lambda r: r[1])
# the last step = the last 5 k_loop launches
loops = [i for i, r in enumerate(rows) if r[0].startswith("k_loop")]
first = loops[-5]
# start a little before: find the first k_fft of that step
i0 = max(i for i, r in enumerate(rows[:first]) if r[0].startswith("k_fft") and (i == 0 or rows[i - 1][0] not in ("k_fft<2, 12, true>",))) if first else 0
t0 = rows[i0][1]
for n, s, e in rows[i0:]:
    print("%-22s %9.3f -> %9.3f  (%7.3f ms)" % (n, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
tail -c 600 $out/tl.err > $out/tl.err.tail; rm -f $out/tl.err
cat $out/timeline.txt
