#!/bin/bash
# Runs on the GPU box: kernel trace of `bench.py --layer N`, the dispatches of the last step with the idle time before each.
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl12 && cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl12 -o tl -- python3 bench.py --layer ${1:-2} --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/tl12/**/*kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows = [r for r in rows if r[0].startswith("k") and not r[0].startswith("k_synth")]
rows.sort(key=lambda r: r[1])
rows = rows[-14:]
t0 = rows[0][1]; prev = None
for n, s, e in rows:
    print("%-20s %9.3f -> %9.3f (%7.3f ms)  idle before: %6.3f ms" % (n[:20], (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, 0 if prev is None else (s - prev) / 1e6))
    prev = e
PY
