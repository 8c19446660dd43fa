#!/bin/bash
# Runs on the GPU box: A/B/C... of ENVIRONMENT settings of one library on the same device (boxes differ by several per
# cent: numbers from two gpurun calls do not compare).  Every configuration is "name:VAR=value,VAR=value" ("base:" = the
# defaults); they are run in turn, R rounds, the bench line's ms per step printed for each, and -- from a kernel trace of
# one more step of the first round -- the time between consecutive k_loop launches (end of one to start of the next).
# Usage: tools/gpu_abenv.sh <tag> <rounds> <config> [<config> ...] [-- bench args...]
tag=$1; R=$2; shift 2
cfgs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do cfgs+=("$1"); shift; done
[ "$1" == "--" ] && shift
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for r in $(seq 1 $R); do
  for c in "${cfgs[@]}"; do
    n=${c%%:*}; e=${c#*:}
    envs=(); IFS=',' read -ra kv <<< "$e"; for x in "${kv[@]}"; do [ -n "$x" ] && envs+=("$x"); done
    env "${envs[@]}" python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 "$@" > $out/bench_${n}_$r.json 2> $out/bench_${n}_$r.err || { echo "bench failed for $n"; tail -5 $out/bench_${n}_$r.err; }
    ms=$(python3 -c "import json;d=json.loads(open('$out/bench_${n}_$r.json').read().strip().splitlines()[-1]);print('%.2f ms/step  value %s  exact %s' % (d['ms_per_step'], d['value'], d['parity_spot_check']['bit_exact']))" 2>/dev/null)
    gaps=""
    if [ $r -eq 1 ]; then
      rm -rf /tmp/ab_tl
      # (rocprofv3 needs the program itself after --: the environment is exported for this one command by a subshell)
      ( for x in "${envs[@]}"; do export "$x"; done; timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_tl -o tl -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > /dev/null 2> $out/tl_$n.err )
      gaps=$(python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ab_tl/**/*kernel_trace.csv", recursive=True)
if f:
    rows = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f[0]))]
    loops = sorted([r for r in rows if r[0].startswith("k_loop")], key=lambda r: r[1])[-10:]
    print("k_loop ms " + " ".join("%.1f" % ((e - s) / 1e6) for _, s, e in loops) + " | gaps ms " + " ".join("%.1f" % ((loops[i + 1][1] - loops[i][2]) / 1e6) for i in range(len(loops) - 1)))
PY
)
      python3 - "$out/timeline_$n.txt" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/ab_tl/**/*kernel_trace.csv", recursive=True)
if f:
    rows = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in csv.DictReader(open(f[0]))]
    rows = sorted([r for r in rows if r[0].startswith("k_") and not r[0].startswith("k_synth")], key=lambda r: r[1])
    loops = [i for i, r in enumerate(rows) if r[0] == "k_loop"]
    i0 = max(0, loops[-6] - 6) if len(loops) >= 6 else 0
    t0 = rows[i0][1]
    with open(sys.argv[1], "w") as o:
        for n, s, e, q in rows[i0:]:
            o.write("%-22s q%-3s %9.3f -> %9.3f  (%7.3f ms)\n" % (n, q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
PY
    fi
    echo "$n round $r  $ms  $gaps" | tee -a $out/summary.txt
  done
done
