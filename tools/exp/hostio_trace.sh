cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/hio; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/hio -o hio -- python3 bench.py --steps 2 --warmup 1 --host-io --no-cpu-baseline > gpurun_out/hio_bench.json 2> gpurun_out/hio_bench.err
ls /tmp/hio/*/ | head
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/hio/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print(r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"])
for f in glob.glob("/tmp/hio/**/*memory_copy_stats.csv", recursive=True):
    print(open(f).read()[:1500])
for f in glob.glob("/tmp/hio/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(len(rows), "copies; columns", list(rows[0].keys()))
    for r in rows[-14:]:
        print({k: r[k] for k in r if k in ("Direction", "Start_Timestamp", "End_Timestamp", "Kind", "Bytes", "Src_Agent_Id", "Dst_Agent_Id")})
PY
