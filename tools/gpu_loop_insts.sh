#!/bin/bash
# Runs on the GPU box: SQ instruction counters of k_loop for one step of the bench command, for each of the libraries named
# (mp3-enc-bsd_amd/ab_now/lib<name>.so, tools/ab_build.sh): vector / scalar / LDS / branch instructions per (granule, channel).
# Usage: tools/gpu_loop_insts.sh <tag> <name> [<name> ...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for n in "$@"; do
  export MP3MI_LIB=$GRAFT_REPO_ROOT/mp3-enc-bsd_amd/ab_now/lib$n.so
  raw=/tmp/li_$n; rm -rf $raw
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $raw -o a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $out/$n.err
  python3 tools/pmc_summary.py $raw $out/insts_$n.json > /dev/null
  python3 - $out/insts_$n.json $n <<'PY' | tee -a $out/summary.txt
import json, sys
m = json.load(open(sys.argv[1]))
for k, v in m.items():
    if not k.startswith("k_loop"): continue
    n = 4096 * 383 * 4  # (granule, channel) records of a 4096 x 383 stereo step
    print("%-8s %s: per (granule, channel) VALU %.0f  SALU %.0f  LDS %.0f  branch %.0f  SMEM %.0f  VMEM %.0f" % (sys.argv[2], k, v["SQ_INSTS_VALU"] / n, v["SQ_INSTS_SALU"] / n, v["SQ_INSTS_LDS"] / n, v["SQ_INSTS_BRANCH"] / n, v["SQ_INSTS_SMEM"] / n, (v["SQ_INSTS_VMEM_RD"] + v["SQ_INSTS_VMEM_WR"]) / n))
PY
done
