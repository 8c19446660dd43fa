#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of the default bench command only (no PMC passes).
# Usage: tools/gpu_kstats.sh <tag> [bench args...]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
raw=/tmp/prof_$tag
rm -rf $raw
cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $raw/kt -o kt -- python3 bench.py --no-cpu-baseline "$@" > $out/bench_under_rocprof.json 2> $out/kt.err
cp $(find $raw/kt -name '*kernel_stats.csv' | head -1) $out/kernel_stats.csv 2>/dev/null
tail -c 1500 $out/kt.err > $out/kt.err.tail; rm -f $out/kt.err
cut -d, -f1-4 $out/kernel_stats.csv
