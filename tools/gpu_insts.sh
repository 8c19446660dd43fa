#!/bin/bash
# Runs on the GPU box: instruction counters (SQ_INSTS_*) per kernel for one step of the default bench command.
# Usage: tools/gpu_insts.sh <tag> [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
raw=/tmp/insts_$tag
rm -rf $raw
cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $raw/a -o a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /dev/null 2> $out/insts.err
python3 tools/pmc_summary.py $raw/a $out/insts.json > /dev/null
tail -c 1000 $out/insts.err > $out/insts.err.tail; rm -f $out/insts.err
python3 - <<PY
import json
d=json.load(open("$out/insts.json"))
for k,v in d.items():
    w=v.get("SQ_WAVES",0) or 1
    print("%-14s waves %10d  VALU %8.0f  SALU %8.0f  LDS %7.0f per wave"%(k,w,v.get("SQ_INSTS_VALU",0)/w,v.get("SQ_INSTS_SALU",0)/w,v.get("SQ_INSTS_LDS",0)/w))
PY
