#!/bin/bash
# Runs on the GPU box: kernel trace of the drop-in binary (the reference's own driver over libmp3mi.so) on a 383-frame
# file: per-kernel totals, the share of the process's time the device was busy, and the timeline of one frame in the middle.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python3 - <<'PY'
import os, sys
sys.path.insert(0, "tests")
from mp3common import Mp3mi, SEED
from test_dropin import write_wav
mp = Mp3mi()
write_wav("/tmp/dt_a.wav", mp.synth(441000, 2, 44100, 0, SEED), 2, 44100)
PY
rm -rf /tmp/dt; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dt -o dt -- $GRAFT_REPO_ROOT/oracle/_ref/encode_dropin -s 44.1 -b 128 /tmp/dt_a.wav /tmp/dt_a.mp3 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/dt/**/*kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    print("%-40s calls %6s  total %8.2f ms  avg %8.1f us" % (r["Name"][:40], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
    tot += int(r["TotalDurationNs"])
f = glob.glob("/tmp/dt/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
print("kernels busy %.1f ms of a span of %.1f ms (%d dispatches)" % (tot / 1e6, (rows[-1][1] - rows[0][0]) / 1e6, len(rows)))
# one frame in the middle: from a k_fft start to the next
ffts = [i for i, r in enumerate(rows) if r[2].startswith("k_fft") or "k_fft" in r[2]]
starts = [i for j, i in enumerate(ffts) if j == 0 or ffts[j - 1] != i - 1]
a = starts[len(starts) // 2]; b = starts[len(starts) // 2 + 1]
t0 = rows[a][0]
prev_end = t0
for s, e, n in rows[a:b + 1]:
    print("  %-44s start %8.1f us  dur %7.1f us  gap before %6.1f us" % (n[:44], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
PY
