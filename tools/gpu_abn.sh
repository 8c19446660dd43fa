#!/bin/bash
# Runs on the GPU box: tools/gpu_ab.sh for ANY number of prebuilt libraries (mp3-enc-bsd_amd/ab_now/lib<name>.so, built
# with tools/ab_build.sh <name>) on the same device, taken in turn for R rounds: A B C A B C ...  Prints per run the
# bench line's ms per step and, from a kernel trace of one more step, the stand-alone time of the step's last k_loop
# launch and the per-kernel averages of the second half of the step.
# Usage: tools/gpu_abn.sh <rounds> <name> [<name> ...] [-- bench args...]
R=$1; shift
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" = "--" ] && shift
cd $GRAFT_REPO_ROOT
out=gpurun_out/abn_$(IFS=_; echo "${names[*]}")
mkdir -p $out
export TMPDIR=/tmp
for r in $(seq 1 $R); do
  for n in "${names[@]}"; do
    export MP3MI_LIB=$GRAFT_REPO_ROOT/mp3-enc-bsd_amd/ab_now/lib$n.so
    python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $out/bench_${n}_$r.json 2> $out/bench_${n}_$r.err || { echo "bench failed for $n"; tail -5 $out/bench_${n}_$r.err; }
    ms=$(python3 -c "import json;d=json.loads(open('$out/bench_${n}_$r.json').read().strip().splitlines()[-1]);print('%.2f ms/step  value %s  exact %s' % (d['ms_per_step'], d['value'], d['parity_spot_check']['bit_exact']))" 2>/dev/null)
    rm -rf /tmp/ab_tl
    timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_tl -o tl -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > /dev/null 2> $out/tl.err
    python3 - "$n" "$r" "$ms" <<'PY' | tee -a $out/summary.txt
import csv, glob, sys
f = glob.glob("/tmp/ab_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows = [r for r in rows if r[0].startswith("k_") and not r[0].startswith("k_synth")]
rows.sort(key=lambda r: r[1])
loops = [r for r in rows if r[0].startswith("k_loop")]
last = loops[-1]
avg = {}
for n, s, e in rows[len(rows) // 2:]:
    avg.setdefault(n, []).append((e - s) / 1e6)
print("%-10s round %s  %s  | last k_loop alone %.3f ms | " % (sys.argv[1], sys.argv[2], sys.argv[3], (last[2] - last[1]) / 1e6) +
      "  ".join("%s %.2f" % (k.replace("k_", ""), sum(v) / len(v)) for k, v in sorted(avg.items()) if k not in ("k_gate", "k_rank", "k_hist_save")))
PY
  done
done
