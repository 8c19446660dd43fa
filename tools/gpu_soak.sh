#!/bin/bash
# Runs on the GPU box: tools/full_parity.py on inputs other than the bench's (other stream indices of the generator)
# for the four workloads; one JSON line per run in gpurun_out/<tag>.jsonl.  Usage: tools/gpu_soak.sh <tag> [first stream index]
tag=$1; s0=${2:-200000}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag.jsonl
: > $out
for run in "1 0" "1 4096" "3 8192" "4 12288" "2 28672" "1 45056"; do
  set -- $run
  python3 tools/full_parity.py --config $1 --stream0 $((s0 + $2)) --ref-every 64 --out gpurun_out/${tag}_c$1_$2.json 2>/dev/null | tail -1 >> $out
  echo "config $1 stream0 $((s0 + $2)): $(tail -1 $out | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["frames_total"], "frames,", d["mismatching_streams"], "mismatching,", len(d["reference_binary_mismatches"]), "vs reference of", d["compared_with_reference_binary"])')"
done
