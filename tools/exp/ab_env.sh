#!/bin/bash
# Runs on the GPU box: A/B of ONE library under two settings of an environment variable (an option of mp3mi_batch_options_from_env),
# taken in turn for R rounds.  Usage: tools/exp/ab_env.sh <rounds> <VAR> <value A> <value B> [bench args...]
R=$1; V=$2; A=$3; B=$4; shift 4
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for r in $(seq 1 $R); do
  for x in $A $B; do
    export $V=$x
    ms=$(python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('%.2f ms/step exact %s' % (d['ms_per_step'], d['parity_spot_check']['bit_exact']))")
    rm -rf /tmp/ab_tl
    timeout 240 rocprofv3 --kernel-trace --output-format csv -d /tmp/ab_tl -o tl -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 "$@" > /dev/null 2>&1
    python3 - "$V=$x" "$r" "$ms" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/ab_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows = [r for r in rows if r[0].startswith("k_") and not r[0].startswith("k_synth")]
rows.sort(key=lambda r: r[1])
loops = [r for r in rows if r[0].startswith("k_loop")]
gaps = [(loops[i + 1][1] - loops[i][2]) / 1e6 for i in range(len(loops) - 1)]
avg = {}
for n, s, e in rows[len(rows) // 2:]:
    avg.setdefault(n, []).append((e - s) / 1e6)
print("%-22s round %s  %s | gaps between k_loop launches %s | " % (sys.argv[1], sys.argv[2], sys.argv[3], " ".join("%.1f" % g for g in gaps[-4:])) +
      "  ".join("%s %.2f" % (k.replace("k_", ""), sum(v) / len(v)) for k, v in sorted(avg.items()) if k not in ("k_gate", "k_rank", "k_hist_save", "k_cw_fix", "k_cw_fix_reset", "k_prep")))
PY
  done
done
